"""The serving graph's VGG16 trunks on the hand-written MFMA convolution (csrc/conv3x3_mfma.hip).

`MfmaTrunks(net)` runs the 3x3 convolutions + 2x2 pools of lib/networks/MV3D_test.py:34-78 (conv1_1 .. conv5_3 of every view,
rpn_conv/3x3) through mv3d_conv3x3_f16 / mv3d_maxpool2x2_f16 (or their _bf16 twins when the graph's amp_dtype is bfloat16):
16-bit operands, f32 accumulation, activations kept in HBM as framed NHWC f16 / bf16 (one-pixel zero frame = the SAME padding, written once when a buffer is made).  Forward only, lower precision
than the reference's fp32 graph (BASELINE configs[4] "fp16 VGG16"): never the parity contract; the maps the hot-path layers
read (conv5_3*) are handed over as f32 NHWC, as those layers require.

conv1_1 (9 / 3 input channels) goes through the kernel's input-layer variant: channels zero-padded to 16 in the framed buffer,
a K step = 4 taps x 16 channels (3 steps instead of 9).
Packed weights are cached per layer and re-packed when the fp32 parameter changes (`tensor._version`).
"""
import torch

from . import ops


class MfmaTrunks:
    def __init__(self, net, vgg, dtype=torch.float16):
        self.net = net
        self.vgg = vgg                       # [(stem, c_out, pool_after)]
        self.dtype = dtype                   # float16 (default), bfloat16 (f16's range is the usual worry with raw-pixel inputs) or
                                             # float32 = the reference's precision on the f32 MFMA
        self._w = {}                         # name -> (version, packed f16 weights, f32 bias)
        self._buf = {}                       # (tag, B, H, W, C) -> framed f16 buffer (frame stays zero: only interiors are written)
        self.fuse_pools = True               # trunks(): conv1_2 / conv2_2 with their pools in one launch (16-bit trunks)

    def _packed(self, name, input_layer=False):
        w, b = self.net.params[name]
        ver = (w._version, b._version)
        hit = self._w.get(name)
        if hit is None or hit[0] != ver:
            if input_layer and self.dtype == torch.float32:      # (f32: the input layer's channels padded to one 32-channel K step)
                packed = ops.pack_conv3x3_weights(w, 32, dtype=self.dtype)
            else:
                packed = (ops.pack_conv3x3_weights_input_layer if input_layer else ops.pack_conv3x3_weights)(w, dtype=self.dtype)
            hit = (ver, packed, b.detach().float().contiguous())
            self._w[name] = hit
        return hit[1], hit[2]

    def clear(self):
        """drop the cached framed buffers and packed weights (e.g. after serving a different batch size)"""
        self._w.clear()
        self._buf.clear()

    def _framed(self, tag, B, H, W, C, dev):
        key = (tag, B, H, W, C)
        buf = self._buf.get(key)
        if buf is None:
            buf = self._buf[key] = ops.framed_buffer(B, H, W, C, dev, self.dtype)
        return buf

    def trunk(self, x_nhwc, suffix, last_framed):
        """one trunk (see trunks()): x_nhwc (B, H, W, 9 | 3) f32 -> its conv5_3, framed in the trunk's type if `last_framed` (rpn_conv/3x3
        reads it) else the bare f32 NHWC map"""
        return self.trunks([x_nhwc], [suffix], [last_framed])[0]

    def trunks(self, xs_nhwc, suffixes, last_framed):
        """Several trunks walked together: every depth is ONE launch for all views (mv3d_conv3x3_views_*, mv3d_maxpool2x2_views_*:
        the views' tiles share the grid, so a view's partly filled last round of workgroups is filled by the others' -- what the
        side stream of round 3 did, from one stream, graph-capturable).  xs_nhwc[v] (B, H, W, 9 | 3) f32; last_framed[v]: conv5_3 of
        view v as a framed map of the trunk's type (rpn_conv/3x3 reads it) instead of the bare f32 NHWC map.  Returns the list."""
        nv, n = len(xs_nhwc), len(self.vgg)
        dev = xs_nhwc[0].device
        L = self.net.layers
        cpad = 32 if self.dtype == torch.float32 else 16
        cur, hw = [], []
        for x, sfx in zip(xs_nhwc, suffixes):
            B, H, W, _ = x.shape
            cur.append(ops.frame_nhwc_f16(x.contiguous(), self._framed("in" + sfx, B, H, W, cpad, dev)))
            hw.append((H, W))
        outs = [None] * nv
        for i, (stem, cout, pool) in enumerate(self.vgg):
            wb = [self._packed(stem + sfx, input_layer=(i == 0)) for sfx in suffixes]
            if i == n - 1:
                for framed in (True, False):                      # (the two output forms are two launches)
                    vs = [v for v in range(nv) if bool(last_framed[v]) == framed]
                    if not vs:
                        continue
                    if framed:
                        ys = [self._framed(stem + suffixes[v], cur[v].shape[0], hw[v][0], hw[v][1], cout, dev) for v in vs]
                    else:
                        ys = [torch.empty((cur[v].shape[0], hw[v][0], hw[v][1], cout), dtype=torch.float32, device=dev) for v in vs]
                    ops.conv3x3_views([(cur[v], wb[v][0], wb[v][1], None, y) for v, y in zip(vs, ys)], out_framed=framed, out_f32=not framed)
                    for v, y in zip(vs, ys):
                        outs[v] = y
                        L[stem + suffixes[v]] = y[:, 1:-1, 1:-1] if framed else y
                return outs
            if pool and self.fuse_pools and i > 0 and self.dtype != torch.float32 and self._pool_waste(hw, cout) < 0.08 and \
                    all(c.numel() * c.element_size() < 2 ** 31 - 1 for c in cur):
                # the pool in the convolution's epilogue (mv3d_conv3x3_pool_views_*): the full-size map is never written (and is
                # not in net.layers: only the pooled one exists).  Skipped where the two-row tiles would hang too far over the
                # maps' right edge (conv3_3 of the 152-wide BEV map: 21 % of the tiles' columns)
                hw = [(h // 2, w // 2) for h, w in hw]
                ps = [self._framed(stem + suffixes[v] + "/pool", cur[v].shape[0], hw[v][0], hw[v][1], cout, dev) for v in range(nv)]
                ops.conv3x3_pool_views([(cur[v], wb[v][0], wb[v][1], ps[v]) for v in range(nv)])
                cur = ps
                continue
            ys = [self._framed(stem + suffixes[v], cur[v].shape[0], hw[v][0], hw[v][1], cout, dev) for v in range(nv)]
            ops.conv3x3_views([(cur[v], wb[v][0], wb[v][1], None, ys[v]) for v in range(nv)])
            for v in range(nv):
                L[stem + suffixes[v]] = ys[v][:, 1:-1, 1:-1]
            if pool:
                hw = [(h // 2, w // 2) for h, w in hw]
                ps = [self._framed(stem + suffixes[v] + "/pool", ys[v].shape[0], hw[v][0], hw[v][1], cout, dev) for v in range(nv)]
                ops.maxpool2x2_views([(ys[v], ps[v]) for v in range(nv)])
                cur = ps
            else:
                cur = ys
        return outs

    @staticmethod
    def _pool_waste(hw, cout):
        """share of the pooled-epilogue tiles' columns that hang over the maps' right edge (tile = 2 rows x 64 | 128 columns)"""
        tw = 128 if cout % 128 else 64
        used = sum((h // 2) * 2 * (w // 2) for h, w in hw)
        tiled = sum((h // 2) * ((2 * (w // 2) + tw - 1) // tw) * tw for h, w in hw)
        return 1.0 - used / max(tiled, 1)

    def rpn_conv(self, conv5_3_framed):
        """rpn_conv/3x3 (MV3D_test.py:82-84) on the framed BEV conv5_3 -> (B, H, W, 512) NHWC of the trunk's type"""
        wp, bias = self._packed("rpn_conv/3x3")
        return ops.conv3x3_f16(conv5_3_framed, wp, bias, out_framed=False, out_f32=False)


def serving_layers(vgg, inputs=(("", 608, 608, 9), ("_2", 375, 1242, 3), ("_3", 64, 512, 3))):
    """[(name, H, W, c_in, c_out)] of every 3x3 convolution of the serving graph on KITTI-shaped inputs (+ rpn_conv/3x3)"""
    rows = []
    for suffix, H, W, c in inputs:
        for stem, cout, pool in vgg:
            rows.append((stem + suffix, H, W, c, cout))
            c = cout
            if pool:
                H, W = H // 2, W // 2
        if suffix == "":
            rows.append(("rpn_conv/3x3", H, W, c, 512))
    return rows


def bench_conv_layers(vgg, batch=16, reps=3, dtype=torch.float16):
    """Roofline entry of the convolution kernel for bench.py, AS THE SERVING STEP LAUNCHES IT: per VGG depth one grouped launch for the
    BEV / image / front-view trunks of the 3-view serving graph (conv1_2 / conv2_2 of the 16-bit trunks with their 2x2 pools in the
    epilogue), conv5_3 in its two output forms, rpn_conv/3x3 on the BEV map -- 15 launches for the 40 convolutions at `batch` frames,
    each timed with HIP events on the launch stream over `reps` launches after one warm-up.  achieved = ALGORITHMIC flops (2 * B*H*W *
    c_out * 9 * c_in with the true c_in, i.e. conv1_1's zero padding is not counted; the pools' comparisons are not counted either)
    / time; peak = the dense MFMA peak of MI355X_MICROARCH.md for the operand type (f16 / bf16: 2.5 PFLOP/s; f32 on
    v_mfma_f32_32x32x2_f32: 157.3 TFLOP/s)."""
    dev = torch.device("cuda")
    f32 = dtype == torch.float32
    peak = 157.3 if f32 else 2500.0
    inputs = (("", 608, 608, 9), ("_2", 375, 1242, 3), ("_3", 64, 512, 3))
    hw = [(h, w) for _, h, w, _ in inputs]
    cins = [c for _, _, _, c in inputs]
    tot_fl, tot_ms, per, nconv = 0.0, 0.0, {}, 0

    def timed(fn):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps

    def operands(H, W, cin, cout, first):
        cpad = (32 if f32 else 16) if first else cin
        x = ops.framed_buffer(batch, H, W, cpad, dev, dtype)
        x[:, 1:-1, 1:-1, :cin] = torch.randn((batch, H, W, cin), device=dev, dtype=dtype)
        w = torch.randn((cout, cin, 3, 3), device=dev) * (2.0 / (9 * cin)) ** 0.5
        wp = ops.pack_conv3x3_weights_input_layer(w, dtype=dtype) if (first and not f32) else ops.pack_conv3x3_weights(w, cpad, dtype=dtype)
        return x, wp, torch.zeros(cout, device=dev)

    for i, (stem, cout, pool) in enumerate(vgg):
        ops_v = [operands(hw[v][0], hw[v][1], cins[v], cout, i == 0) for v in range(3)]
        fl = sum(2.0 * batch * hw[v][0] * hw[v][1] * cout * 9 * cins[v] for v in range(3))
        fused = pool and i > 0 and not f32 and MfmaTrunks._pool_waste(hw, cout) < 0.08
        if fused:
            outs = [ops.framed_buffer(batch, h // 2, w // 2, cout, dev, dtype) for h, w in hw]
            ms = timed(lambda: ops.conv3x3_pool_views([(ops_v[v][0], ops_v[v][1], ops_v[v][2], outs[v]) for v in range(3)]))
        else:
            outs = [ops.framed_buffer(batch, h, w, cout, dev, dtype) for h, w in hw]
            ms = timed(lambda: ops.conv3x3_views([(ops_v[v][0], ops_v[v][1], ops_v[v][2], None, outs[v]) for v in range(3)]))
        per[stem + (" (+ pool)" if fused else "")] = {"ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1)}
        tot_fl += fl
        tot_ms += ms
        nconv += 3
        cins = [cout] * 3
        if pool:
            hw = [(h // 2, w // 2) for h, w in hw]
        del ops_v, outs
    x, wp, b = operands(hw[0][0], hw[0][1], 512, 512, False)                     # rpn_conv/3x3 on the BEV conv5_3
    out = torch.empty((batch, hw[0][0], hw[0][1], 512), dtype=dtype, device=dev)
    ms = timed(lambda: ops.conv3x3_f16(x, wp, b, out=out, out_framed=False, out_f32=f32))
    fl = 2.0 * batch * hw[0][0] * hw[0][1] * 512 * 9 * 512
    per["rpn_conv/3x3"] = {"ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1)}
    tot_fl += fl
    tot_ms += ms
    nconv += 1
    ach = tot_fl / tot_ms / 1e9
    best = max(per.items(), key=lambda kv: kv[1]["tflops"])
    # what a plain library GEMM reaches on THIS box right now (torch.matmul = hipBLASLt, 8192 x 4608 x 8192, randn operands): the
    # matrix cores do not hold the clock the datasheet peak assumes, so `frac` against 2.5 PFLOP/s understates how close a kernel is
    # to what the part sustains (profiles/r04_gemm_ceiling.txt: 1.2 - 1.36 PFLOP/s); reported next to `peak`, never instead of it
    lib = None
    if not f32:
        a_, b_ = torch.randn(8192, 4608, device=dev, dtype=dtype), torch.randn(4608, 8192, device=dev, dtype=dtype)
        lib = round(2.0 * 8192 * 4608 * 8192 / timed(lambda: torch.matmul(a_, b_)) / 1e9, 1)
        del a_, b_
    mfma = "v_mfma_f32_32x32x2_f32 (exact f32)" if f32 else "v_mfma_f32_32x32x16_%s" % ("f16" if dtype == torch.float16 else "bf16")
    return {"kernel": "conv3x3_f16_kernel<%s> (%s; the %d 3x3 convolutions of the 3-view serving graph as the step launches them: %d "
                      "grouped launches, batch %d)" % (str(dtype).split(".")[-1], mfma, nconv, len(per), batch),
            "bound": "mfma", "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "alg_flop_per_step": tot_fl, "ms_per_step": round(tot_ms, 3), "launches_timed": len(per) * reps,
            "best_layer": {"name": best[0], **best[1]}, "per_depth": per, "traffic": None,
            "library_gemm_tflops_same_box": lib}
