"""The TRAINING graph's VGG16 trunks with forward AND backward convolutions on the hand-written MFMA kernel.

Opt-in (`MV3D.mfma_trunk = True` on a TRAIN graph), in one of two precisions (`dtype`):
  bfloat16  mixed precision: bf16 activations / gradients / weight copies, f32 accumulation in the matrix cores, fp32 master weights
            and optimiser -- a lower precision than the reference's fp32 training (lib/fast_rcnn/train_mv.py:138-219), reported next
            to it, never instead of it.  bf16 rather than f16: the gradients of a 13-layer trunk span more than f16's 5-bit exponent
            and would need loss scaling.  (The kernel names below are this variant's.)
  float32   the reference's precision: f32 framed maps, the same three convolutions on the exact-f32 instantiations
            (mv3d_conv3x3_f32, mv3d_conv3x3_wgrad_f32, mv3d_maxpool2x2_bwd_f32: v_mfma_f32_32x32x2_f32, f32 products and sums).

One `torch.autograd.Function` per trunk (conv1_1 .. conv5_3 of lib/networks/MV3D_train.py:44-81):

  forward    x (B, H, W, 9 | 3) f32 -> framed bf16 -> 13 x mv3d_conv3x3_bf16 (+ bias, ReLU) with mv3d_maxpool2x2_bf16 after
             conv1_2 / conv2_2 / conv3_3 -> conv5_3 as f32 NHWC (what RoiPool and rpn_conv/3x3 read).  Every layer's framed
             input and output stay alive for the backward pass.
  backward   per layer, last to first:
             dY   = gradient w.r.t. the layer's pre-activation, framed bf16 with a ZERO frame (= dX of the layer above times the
                    ReLU mask; through a pool: mv3d_maxpool2x2_bwd_bf16, mask included)
             dW   = sum over pixels of dY[p][co] * X[p + tap][ci]      (weight gradient: mv3d_conv3x3_wgrad_bf16)
             db   = sum over pixels of dY                              (from the same launch: dY times an all-ones operand)
             dX   = mv3d_conv3x3_bf16(dY, W flipped by 180 degrees with its channel axes swapped): the data gradient of a
                    3x3 / stride 1 / SAME convolution IS such a convolution, and the zero frame of dY is its padding.
"""
import torch

from . import ops

BF = torch.bfloat16


def _wgrad_torch(x_framed, dy_framed, c_in):
    """weight / bias gradient through torch (MIOpen): ((O, c_in, 3, 3) f32, (O,) f32) -- the yardstick of tests / tools, not the
    product path"""
    x = x_framed[:, 1:-1, 1:-1, :c_in].permute(0, 3, 1, 2)
    dy = dy_framed[:, 1:-1, 1:-1].permute(0, 3, 1, 2)
    return torch.nn.grad.conv2d_weight(x, (dy.shape[1], c_in, 3, 3), dy, padding=1).float(), dy_framed.sum((0, 1, 2), dtype=torch.float32)


def wgrad_mfma(x_framed, dy_framed, c_in):
    """weight AND bias gradient on the MFMA kernel (csrc/conv3x3_wgrad.hip): ((O, c_in, 3, 3) f32, (O,) f32).  (The input layer's
    9 / 3 channels sit in a 64-channel framed buffer during training, so it goes through the same kernel; its padding channels
    are cut off by it.)"""
    return ops.conv3x3_wgrad_bf16(x_framed, dy_framed, c_in, want_bias=True)


class BufferPool:
    """Framed buffers of a training trunk, kept across steps (their zero frames are written once).  ONE forward / backward pair
    per trunk (tag prefix) may be in flight: the next forward overwrites the activations the backward pass reads -- every forward
    takes a new generation number and a backward pass that finds a newer one raises instead of computing wrong gradients.
    Per tag at most `max_shapes` buffer shapes stay alive (least recently used first out: KITTI images come in four sizes)."""

    def __init__(self, max_shapes=4):
        self._buf = {}                     # tag -> {shape key: buffer}, insertion order = recency
        self._gen = {}
        self.max_shapes = int(max_shapes)

    def begin(self, trunk_tag):
        g = self._gen[trunk_tag] = self._gen.get(trunk_tag, 0) + 1
        return g

    def check(self, trunk_tag, generation):
        if self._gen.get(trunk_tag) != generation:
            raise RuntimeError("BufferPool: trunk %r ran forward again before this backward pass (its saved activations are "
                               "overwritten); use one pool per forward / backward pair in flight, or pool=None" % trunk_tag)

    def get(self, tag, B, H, W, C, dev, dtype=BF):
        key = (B, H, W, C, dtype, str(dev))
        per = self._buf.setdefault(tag, {})
        buf = per.pop(key, None)
        if buf is None:
            buf = ops.framed_buffer(B, H, W, C, dev, dtype)
            while len(per) >= self.max_shapes:
                per.pop(next(iter(per)))
        per[key] = buf                     # (re-inserted: most recently used last)
        return buf


class _NoPool:
    @staticmethod
    def get(tag, B, H, W, C, dev, dtype=BF):
        return ops.framed_buffer(B, H, W, C, dev, dtype)

    @staticmethod
    def begin(trunk_tag):
        return 0

    @staticmethod
    def check(trunk_tag, generation):
        pass


def _pack_pair(w, c_in_pad, want_dgrad, dtype):
    """(forward packing, data-gradient packing | None) of an fp32 OIHW filter in the trunk's type (one launch)"""
    return ops.pack_conv3x3_train_many([(w, c_in_pad, want_dgrad)], dtype=dtype)[0]


class TrunksFunction(torch.autograd.Function):
    """apply(layers, wgrad, pool, nv, x_0 .. x_{nv-1}, (w, b) x 13 of view 0, ... of view nv - 1) -> nv x conv5_3 (B, H', W', 512) f32;
    layers = [(name, c_out, pool_after)], wgrad = callable(x_framed, dy_framed, c_in) -> ((O, c_in, 3, 3) f32, (O,) f32) or None = the
    library's grouped kernel, pool = (BufferPool | None, [tag per view], dtype).

    The nv trunks (BEV / image / front view of lib/networks/MV3D_train.py:44-81) walk the SAME layer list, so every depth is ONE
    launch for all of them -- forward convolution, pool, data gradient (+ ReLU gate), pool gradient, weight gradient + reduce -- and
    all filters of the step are packed by one launch.  At a training batch of 2 a single trunk's launches leave the chip partly idle
    (364 tiles of conv4_x on 512 workgroup slots); the grouped launches fill it from ONE stream, which is also what data parallelism
    and graph capture want (the side streams of round 3 are gone)."""

    @staticmethod
    def forward(ctx, layers, wgrad, pool_tag, nv, *args):
        xs, wb = args[:nv], args[nv:]
        n = len(layers)
        dev = xs[0].device
        bufs, tags, dt = (pool_tag[0] or _NoPool), pool_tag[1], pool_tag[2]
        # (the input layer's channels zero-padded to 64: the same kernels as every other layer, forward and both gradients)
        cpad0 = 64
        ctx.gens = [bufs.begin(t) for t in tags]
        W = lambda v, i: wb[(v * n + i) * 2]
        Bs = lambda v, i: wb[(v * n + i) * 2 + 1]
        # every filter of the step, both packings, in one launch
        packed = ops.pack_conv3x3_train_many([(W(v, i), cpad0 if i == 0 else None, i > 0) for v in range(nv) for i in range(n)], dtype=dt)
        wp = lambda v, i: packed[v * n + i][0]
        shapes = [tuple(x.shape) for x in xs]
        cur = [ops.frame_nhwc_f16(x.contiguous(), bufs.get(tags[v] + "/in", x.shape[0], x.shape[1], x.shape[2], cpad0, dev, dt))
               for v, x in enumerate(xs)]
        hw = [(s[1], s[2]) for s in shapes]
        saved = [[] for _ in range(nv)]                # per view and layer: (framed input, framed output | None for the last, H, W)
        outs = None
        for i, (_, cout, pool) in enumerate(layers):
            biases = [Bs(v, i).detach().float().contiguous() for v in range(nv)]
            if i == n - 1:
                outs = [torch.empty((shapes[v][0], hw[v][0], hw[v][1], cout), dtype=torch.float32, device=dev) for v in range(nv)]
                ops.conv3x3_views([(cur[v], wp(v, i), biases[v], None, outs[v]) for v in range(nv)], out_framed=False, out_f32=True)
                for v in range(nv):
                    saved[v].append((cur[v], None, hw[v][0], hw[v][1]))
                break
            ys = [bufs.get("%s/y%d" % (tags[v], i), shapes[v][0], hw[v][0], hw[v][1], cout, dev, dt) for v in range(nv)]
            ops.conv3x3_views([(cur[v], wp(v, i), biases[v], None, ys[v]) for v in range(nv)])
            for v in range(nv):
                saved[v].append((cur[v], ys[v], hw[v][0], hw[v][1]))
            if pool:
                hw = [(h // 2, w // 2) for h, w in hw]
                ps = [bufs.get("%s/p%d" % (tags[v], i), shapes[v][0], hw[v][0], hw[v][1], cout, dev, dt) for v in range(nv)]
                ops.maxpool2x2_views([(ys[v], ps[v]) for v in range(nv)])
                cur = ps
            else:
                cur = ys
        ctx.layers, ctx.wgrad, ctx.saved, ctx.bufs, ctx.tags, ctx.dt, ctx.nv = layers, wgrad, saved, bufs, tags, dt, nv
        ctx.c0 = [s[3] for s in shapes]
        ctx.packed_dgrad = [[packed[v * n + i][1] for i in range(n)] for v in range(nv)]
        ctx.save_for_backward(*outs)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        outs = ctx.saved_tensors
        layers, saved, nv, bufs, tags, dt = ctx.layers, ctx.saved, ctx.nv, ctx.bufs, ctx.tags, ctx.dt
        n = len(layers)
        dev = outs[0].device
        for t, gen in zip(tags, ctx.gens):
            bufs.check(t, gen)
        grads = [None] * (2 * n * nv)
        # gradient w.r.t. conv5_3's pre-activation, framed (a view nobody used downstream has a zero gradient)
        dys = []
        for v in range(nv):
            _, _, H, W = saved[v][n - 1]
            g = gs[v] if gs[v] is not None else torch.zeros_like(outs[v])
            dys.append(ops.frame_nhwc_f16((g * (outs[v] > 0)).contiguous(),
                                          bufs.get(tags[v] + "/g%d" % (n - 1), g.shape[0], H, W, layers[n - 1][1], dev, dt)))
        zero_bias = torch.zeros(max(c for _, c, _ in layers), dtype=torch.float32, device=dev)      # (the data-gradient convolutions add no bias)
        for i in range(n - 1, -1, -1):
            c_ins = [ctx.c0[v] if i == 0 else layers[i - 1][1] for v in range(nv)]
            if ctx.wgrad is None:                      # one launch + one reduce for the views (the input layers: 9 / 3 real channels each)
                res = ops.conv3x3_wgrad_views([(saved[v][i][0], dys[v]) for v in range(nv)], c_ins, want_bias=True)
            else:                                      # (a yardstick wgrad of the tests)
                fn = ctx.wgrad or wgrad_mfma
                res = [fn(saved[v][i][0], dys[v], c_ins[v]) for v in range(nv)]
            for v in range(nv):
                grads[(v * n + i) * 2], grads[(v * n + i) * 2 + 1] = res[v]
            if i == 0:
                break
            c_in = c_ins[0]
            # two gradient buffers per resolution alternate (dy of layer i is read while dx = dy of layer i - 1 is written)
            obufs = [bufs.get("%s/dx%d" % (tags[v], i), dys[v].shape[0], saved[v][i][2], saved[v][i][3], c_in, dev, dt) for v in range(nv)]
            y_prev = [saved[v][i - 1][1] for v in range(nv)]
            wd = [ctx.packed_dgrad[v][i] for v in range(nv)]
            if layers[i - 1][2]:                       # a pool sits between layer i - 1 and layer i: route through it (mask fused)
                ops.conv3x3_views([(dys[v], wd[v], zero_bias, None, obufs[v]) for v in range(nv)], relu=False)
                gbufs = [bufs.get("%s/g%d" % (tags[v], i - 1), dys[v].shape[0], saved[v][i - 1][2], saved[v][i - 1][3], c_in, dev, dt)
                         for v in range(nv)]
                dys = ops.maxpool2x2_bwd_views([(y_prev[v], obufs[v], gbufs[v]) for v in range(nv)])
            else:                                      # data gradient and the ReLU mask of layer i - 1's output in one launch
                dys = ops.conv3x3_views([(dys[v], wd[v], zero_bias, y_prev[v], obufs[v]) for v in range(nv)], relu=False)
        return (None, None, None, None) + (None,) * nv + tuple(grads)


class ConvReluFunction(torch.autograd.Function):
    """One 3x3 + ReLU layer between f32 NHWC maps (rpn_conv/3x3 on conv5_3, MV3D_train.py:84-86) on the same three kernels:
    apply(wgrad, x_nhwc_f32, w, b) -> (B, H, W, c_out) f32."""

    @staticmethod
    def forward(ctx, wgrad_dt, x_nhwc, w, b):
        wgrad, dt = wgrad_dt
        B, H, W, cin = x_nhwc.shape
        x = ops.frame_nhwc_f16(x_nhwc.contiguous(), ops.framed_buffer(B, H, W, cin, x_nhwc.device, dt))
        wp, wd = _pack_pair(w, None, True, dt)
        out = ops.conv3x3_f16(x, wp, b.detach().float().contiguous(), out_framed=False, out_f32=True)
        ctx.wgrad, ctx.x, ctx.wd, ctx.dt = wgrad, x, wd, dt
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        B, H, W, cout = g.shape
        cin = ctx.x.shape[3]
        dev = g.device
        dy = ops.frame_nhwc_f16((g * (out > 0)).contiguous(), ops.framed_buffer(B, H, W, cout, dev, ctx.dt))
        gw, gb = ctx.wgrad(ctx.x, dy, cin)
        dx = ops.conv3x3_f16(dy, ctx.wd, torch.zeros(cin, dtype=torch.float32, device=dev), relu=False, out_framed=False, out_f32=True)
        return None, dx, gw, gb


def conv_relu(x_nhwc, w, b, wgrad=None, dtype=BF):
    return ConvReluFunction.apply((wgrad or wgrad_mfma, dtype), x_nhwc, w, b)     # (wgrad_mfma: csrc/conv3x3_wgrad.hip, bf16 or f32 by dtype)


def trunks(layers, xs_nhwc, params, suffixes, wgrad=None, pool=None, dtype=BF):
    """conv1_1<suffix> .. conv5_3<suffix> of the TRAIN graph's trunks, ALL of them walked together (one launch per depth and kind):
    xs_nhwc = [x (B, H, W, c) f32 per view], suffixes = ["", "_2", "_3"]; params = {name: [w, b]} (fp32, OIHW); pool: a BufferPool
    that keeps the framed buffers across steps (one forward / backward pair in flight), None = fresh buffers every call.  dtype =
    bfloat16: mixed precision; float32: the reference's precision on the exact-f32 MFMA kernels -- forward, data gradient and
    weight gradient alike.  wgrad: None = the library's kernel (grouped); a callable = per view (tests' yardsticks).
    Returns the list of conv5_3 maps (B, H', W', 512) f32."""
    wb = []
    for suffix in suffixes:
        for stem, _, _ in layers:
            wb += list(params[stem + suffix])
    tags = ["trunk" + sfx for sfx in suffixes]
    return list(TrunksFunction.apply(layers, wgrad, (pool, tags, dtype), len(xs_nhwc), *xs_nhwc, *wb))


def trunk(layers, x_nhwc, params, suffix, wgrad=None, pool=None, dtype=BF):
    """one trunk (see trunks())"""
    return trunks(layers, [x_nhwc], params, [suffix], wgrad=wgrad, pool=pool, dtype=dtype)[0]


def bench_wgrad_layers(vgg, batch=2, reps=3, dtype=BF):
    """Roofline entry of the weight-gradient kernel for bench.py, AS THE TRAINING STEP LAUNCHES IT: per VGG depth one grouped launch
    (+ one reduce launch) for the BEV / image / front-view trunks of the 3-view TRAIN graph at the training batch (conv1_2 .. conv5_3;
    the 64-channel-padded input layers are left out of the flop count), each timed with HIP events on the launch stream over `reps`
    launches after one warm-up.  achieved = sum over the views of 2 * B*H*W * c_out * 9 * c_in / time; peak = the dense MFMA peak of
    MI355X_MICROARCH.md for the operand type (bf16: 2.5 PFLOP/s; f32: 157.3 TFLOP/s)."""
    dev = torch.device("cuda")
    inputs = (("", 608, 608), ("_2", 375, 1242), ("_3", 64, 512))
    hw = [(h, w) for _, h, w in inputs]
    tot_fl, tot_ms, n = 0.0, 0.0, 0
    best = (None, 0.0, 0.0)
    cin = None
    for stem, cout, pool in vgg:
        if cin is not None:
            views = []
            fl = 0.0
            for H, W in hw:
                x = ops.framed_buffer(batch, H, W, cin, dev, dtype)
                x[:, 1:-1, 1:-1] = torch.randn((batch, H, W, cin), device=dev, dtype=dtype)
                dy = ops.framed_buffer(batch, H, W, cout, dev, dtype)
                dy[:, 1:-1, 1:-1] = torch.randn((batch, H, W, cout), device=dev, dtype=dtype)
                views.append((x, dy))
                fl += 2.0 * batch * H * W * cout * 9 * cin
            ops.conv3x3_wgrad_views(views, want_bias=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ops.conv3x3_wgrad_views(views, want_bias=True)
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1) / reps
            tot_fl += fl
            tot_ms += ms
            n += 1
            if fl / ms / 1e9 > best[1]:
                best = (stem, fl / ms / 1e9, ms)
            del views
        cin = cout
        if pool:
            hw = [(h // 2, w // 2) for h, w in hw]
    ach = tot_fl / tot_ms / 1e9
    f32 = dtype == torch.float32
    peak = 157.3 if f32 else 2500.0
    how = "conv3x3_wgrad_f32_kernel + reduce (v_mfma_f32_32x32x2_f32, exact f32" if f32 else \
        "conv3x3_wgrad_kernel + reduce (v_mfma_f32_32x32x16_bf16 fed by ds_read_b64_tr_b16"
    return {"kernel": "%s; the weight gradients of the 3-view training trunks as the step launches them: %d grouped launches "
                      "(BEV + image + front view per VGG depth), batch %d)" % (how, n, batch),
            "bound": "mfma", "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "alg_flop_per_step": tot_fl, "ms_per_step": round(tot_ms, 3), "launches_timed": n * reps,
            "best_layer": {"name": best[0], "tflops": round(best[1], 1), "ms": round(best[2], 4)}, "traffic": None}
