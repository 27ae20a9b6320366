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
    """(forward packing, data-gradient packing | None) of an fp32 OIHW filter in the trunk's type"""
    if dtype == BF:
        return ops.pack_conv3x3_train_bf16(w, c_in_pad, want_dgrad=want_dgrad)
    fwd = ops.pack_conv3x3_weights(w, c_in_pad, dtype=dtype)
    return fwd, (ops.pack_conv3x3_weights(w.detach().flip(2, 3).transpose(0, 1), dtype=dtype) if want_dgrad else None)


class TrunkFunction(torch.autograd.Function):
    """apply(layers, wgrad, pool, x_nhwc_f32, w_0, b_0, ..., w_12, b_12) -> conv5_3 (B, H', W', 512) f32;
    layers = [(name, c_out, pool_after)], wgrad = callable(x_framed, dy_framed, c_in) -> ((O, c_in, 3, 3) f32, (O,) f32),
    pool = (BufferPool | None, tag, dtype)"""

    @staticmethod
    def forward(ctx, layers, wgrad, pool_tag, x_nhwc, *wb):
        B, H, W, c0 = x_nhwc.shape
        dev = x_nhwc.device
        bufs, tag, dt = (pool_tag[0] or _NoPool), pool_tag[1], pool_tag[2]
        # (the input layer's channels zero-padded to 64: the same kernels as every other layer, forward and both gradients)
        cpad0 = 64
        ctx.gen = bufs.begin(tag)
        x = ops.frame_nhwc_f16(x_nhwc.contiguous(), bufs.get(tag + "/in", B, H, W, cpad0, dev, dt))
        saved = []                                     # per layer: (framed input, framed output | None for the last, H, W)
        packed_dgrad = []
        n = len(layers)
        out = None
        for i, (_, cout, pool) in enumerate(layers):
            w, b = wb[2 * i], wb[2 * i + 1]
            wp, wd = _pack_pair(w, cpad0 if i == 0 else None, i > 0, dt)              # (bf16: both packings in one launch)
            packed_dgrad.append(wd)
            bias = b.detach().float().contiguous()
            if i == n - 1:
                out = ops.conv3x3_f16(x, wp, bias, out_framed=False, out_f32=True)
                saved.append((x, None, H, W))
                break
            y = ops.conv3x3_f16(x, wp, bias, out=bufs.get("%s/y%d" % (tag, i), B, H, W, cout, dev, dt))
            saved.append((x, y, H, W))
            if pool:
                H, W = H // 2, W // 2
                x = ops.maxpool2x2_f16(y, out=bufs.get("%s/p%d" % (tag, i), B, H, W, cout, dev, dt))
            else:
                x = y
        ctx.layers, ctx.wgrad, ctx.saved, ctx.c0, ctx.bufs, ctx.tag, ctx.dt = layers, wgrad, saved, c0, bufs, tag, dt
        ctx.packed_dgrad = packed_dgrad
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        layers, saved = ctx.layers, ctx.saved
        n = len(layers)
        B = g.shape[0]
        dev = g.device
        grads = [None] * (2 * n)
        # gradient w.r.t. conv5_3's pre-activation, framed
        bufs, tag, dt = ctx.bufs, ctx.tag, ctx.dt
        bufs.check(tag, ctx.gen)
        x_last, _, H, W = saved[n - 1]
        dy = ops.frame_nhwc_f16((g * (out > 0)).contiguous(), bufs.get(tag + "/g%d" % (n - 1), B, H, W, layers[n - 1][1], dev, dt))
        zero_bias = torch.zeros(max(c for _, c, _ in layers), dtype=torch.float32, device=dev)      # (the data-gradient convolutions add no bias)
        for i in range(n - 1, -1, -1):
            x_in, _, H, W = saved[i]
            c_in = ctx.c0 if i == 0 else layers[i - 1][1]
            grads[2 * i], grads[2 * i + 1] = ctx.wgrad(x_in, dy, c_in)
            if i == 0:
                break
            # two gradient buffers per resolution alternate (dy of layer i is read while dx = dy of layer i - 1 is written)
            _, y_prev, Hp, Wp_ = saved[i - 1]
            out_buf = bufs.get("%s/dx%d" % (tag, i), B, H, W, c_in, dev, dt)
            if layers[i - 1][2]:                       # a pool sits between layer i - 1 and layer i: route through it (mask fused)
                dx = ops.conv3x3_f16(dy, ctx.packed_dgrad[i], zero_bias, relu=False, out=out_buf)
                dy = ops.maxpool2x2_bwd_bf16(y_prev, dx, bufs.get("%s/g%d" % (tag, i - 1), B, Hp, Wp_, c_in, dev, dt))
            elif dt == BF:                             # data gradient and the ReLU mask of layer i - 1's output in one launch
                dy = ops.conv3x3_gated_bf16(dy, ctx.packed_dgrad[i], zero_bias, y_prev, out_buf)
            else:
                dy = ops.conv3x3_f16(dy, ctx.packed_dgrad[i], zero_bias, relu=False, out=out_buf).mul_(y_prev > 0)
        return (None, None, None, None) + tuple(grads)


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


def _default_wgrad(dtype):
    return wgrad_mfma                         # (bf16: transposing LDS reads; f32: one ds_read_b32 per operand -- csrc/conv3x3_wgrad.hip)


def conv_relu(x_nhwc, w, b, wgrad=None, dtype=BF):
    return ConvReluFunction.apply((wgrad or _default_wgrad(dtype), dtype), x_nhwc, w, b)


def trunk(layers, x_nhwc, params, suffix, wgrad=None, pool=None, dtype=BF):
    """conv1_1<suffix> .. conv5_3<suffix> of a TRAIN graph: params = {name: [w, b]} (fp32, OIHW); pool: a BufferPool that keeps
    the framed buffers across steps (one forward / backward pair in flight), None = fresh buffers every call.  dtype = bfloat16:
    mixed precision; float32: the reference's precision on the exact-f32 MFMA kernels (v_mfma_f32_32x32x2_f32) -- forward, data
    gradient and weight gradient alike."""
    wb = []
    for stem, _, _ in layers:
        wb += list(params[stem + suffix])
    return TrunkFunction.apply(layers, wgrad or _default_wgrad(dtype), (pool, "trunk" + suffix, dtype), x_nhwc, *wb)


def bench_wgrad_layers(vgg, batch=2, reps=3, dtype=BF):
    """Roofline entry of the weight-gradient kernel for bench.py: every trunk layer of the 3-view TRAIN graph it serves (conv1_2 ..
    conv5_3 of the three trunks; the 64-channel-padded input layers are left out of the flop count) at the training batch, each
    timed with HIP events on the launch stream over `reps` launches (kernel + its split-K reduce) after one warm-up.
    achieved = 2 * B*H*W * c_out * 9 * c_in / time; peak = the dense MFMA peak of MI355X_MICROARCH.md for the operand type (bf16:
    2.5 PFLOP/s; f32: 157.3 TFLOP/s)."""
    from .trunk import serving_layers
    dev = torch.device("cuda")
    tot_fl, tot_ms, n = 0.0, 0.0, 0
    best = (None, 0.0, 0.0)
    for name, H, W, cin, cout in serving_layers(vgg):
        if cin < 64 or name.startswith("rpn_conv"):
            continue
        x = ops.framed_buffer(batch, H, W, cin, dev, dtype)
        x[:, 1:-1, 1:-1] = torch.randn((batch, H, W, cin), device=dev, dtype=dtype)
        dy = ops.framed_buffer(batch, H, W, cout, dev, dtype)
        dy[:, 1:-1, 1:-1] = torch.randn((batch, H, W, cout), device=dev, dtype=dtype)
        ops.conv3x3_wgrad_bf16(x, dy)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.conv3x3_wgrad_bf16(x, dy)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = 2.0 * batch * H * W * cout * 9 * cin
        tot_fl += fl
        tot_ms += ms
        n += 1
        if fl / ms / 1e9 > best[1]:
            best = (name, fl / ms / 1e9, ms)
    ach = tot_fl / tot_ms / 1e9
    f32 = dtype == torch.float32
    peak = 157.3 if f32 else 2500.0
    how = "conv3x3_wgrad_f32_kernel + reduce (v_mfma_f32_32x32x2_f32, exact f32" if f32 else \
        "conv3x3_wgrad_kernel + reduce (v_mfma_f32_32x32x16_bf16 fed by ds_read_b64_tr_b16"
    return {"kernel": "%s; the %d weight gradients of the 3-view training trunks, batch %d)" % (how, n, batch),
            "bound": "mfma", "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "alg_flop_per_step": tot_fl, "ms_per_step": round(tot_ms, 3), "launches_timed": n * reps,
            "best_layer": {"name": best[0], "tflops": round(best[1], 1), "ms": round(best[2], 4)}, "traffic": None}
