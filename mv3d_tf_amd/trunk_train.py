"""The TRAINING graph's VGG16 trunks with forward AND backward convolutions on the hand-written MFMA kernel.

Mixed precision, opt-in (`MV3D.mfma_trunk = True` on a TRAIN graph): bfloat16 activations / gradients / weight copies, f32
accumulation in the matrix cores, fp32 master weights and optimiser -- a lower precision than the reference's fp32 training
(lib/fast_rcnn/train_mv.py:138-219), reported next to it, never instead of it.  bfloat16 rather than float16: the gradients of a
13-layer trunk span more than f16's 5-bit exponent and would need loss scaling.

One `torch.autograd.Function` per trunk (conv1_1 .. conv5_3 of lib/networks/MV3D_train.py:44-81):

  forward    x (B, H, W, 9 | 3) f32 -> framed bf16 -> 13 x mv3d_conv3x3_bf16 (+ bias, ReLU) with mv3d_maxpool2x2_bf16 after
             conv1_2 / conv2_2 / conv3_3 -> conv5_3 as f32 NHWC (what RoiPool and rpn_conv/3x3 read).  Every layer's framed
             input and output stay alive for the backward pass.
  backward   per layer, last to first:
             dY   = gradient w.r.t. the layer's pre-activation, framed bf16 with a ZERO frame (= dX of the layer above times the
                    ReLU mask; through a pool: mv3d_maxpool2x2_bwd_bf16, mask included)
             dW   = sum over pixels of dY[p][co] * X[p + tap][ci]      (weight gradient)
             db   = sum over pixels of dY
             dX   = mv3d_conv3x3_bf16(dY, W flipped by 180 degrees with its channel axes swapped): the data gradient of a
                    3x3 / stride 1 / SAME convolution IS such a convolution, and the zero frame of dY is its padding.
"""
import torch

from . import ops

BF = torch.bfloat16


def _dgrad_weights(w_oihw):
    """(O, I, 3, 3) -> packed bf16 weights of the data-gradient convolution: (I, 9 * O), W'[i][ky][kx][o] = W[o][i][2-ky][2-kx]"""
    return ops.pack_conv3x3_weights(w_oihw.detach().flip(2, 3).transpose(0, 1), dtype=BF)


def _wgrad_torch(x_framed, dy_framed, c_in):
    """weight gradient through torch (MIOpen): (O, c_in, 3, 3) f32"""
    x = x_framed[:, 1:-1, 1:-1, :c_in].permute(0, 3, 1, 2)
    dy = dy_framed[:, 1:-1, 1:-1].permute(0, 3, 1, 2)
    return torch.nn.grad.conv2d_weight(x, (dy.shape[1], c_in, 3, 3), dy, padding=1).float()


def wgrad_mfma(x_framed, dy_framed, c_in):
    """weight gradient on the MFMA kernel (csrc/conv3x3_wgrad.hip); the input layer (9 / 3 channels, 0.1 % of the trunk's
    flops, a K dimension the 64-channel tiles do not cover) stays with torch"""
    if x_framed.shape[3] % 64:
        return _wgrad_torch(x_framed, dy_framed, c_in)
    dw = ops.conv3x3_wgrad_bf16(x_framed, dy_framed)                 # (O, 9, I) f32
    return dw.reshape(dw.shape[0], 3, 3, dw.shape[2]).permute(0, 3, 1, 2)


class TrunkFunction(torch.autograd.Function):
    """apply(layers, wgrad, x_nhwc_f32, w_0, b_0, ..., w_12, b_12) -> conv5_3 (B, H', W', 512) f32;
    layers = [(name, c_out, pool_after)], wgrad = callable(x_framed, dy_framed, c_in) -> (O, c_in, 3, 3) f32"""

    @staticmethod
    def forward(ctx, layers, wgrad, x_nhwc, *wb):
        B, H, W, c0 = x_nhwc.shape
        dev = x_nhwc.device
        x = ops.frame_nhwc_f16(x_nhwc.contiguous(), ops.framed_buffer(B, H, W, 16, dev, BF))
        saved = []                                     # per layer: (framed input, framed output | None for the last, H, W)
        n = len(layers)
        out = None
        for i, (_, cout, pool) in enumerate(layers):
            w, b = wb[2 * i], wb[2 * i + 1]
            wp = ops.pack_conv3x3_weights_input_layer(w, BF) if i == 0 else ops.pack_conv3x3_weights(w, dtype=BF)
            bias = b.detach().float().contiguous()
            if i == n - 1:
                out = ops.conv3x3_f16(x, wp, bias, out_framed=False, out_f32=True)
                saved.append((x, None, H, W))
                break
            y = ops.conv3x3_f16(x, wp, bias)
            saved.append((x, y, H, W))
            if pool:
                H, W = H // 2, W // 2
                x = ops.maxpool2x2_f16(y)
            else:
                x = y
        ctx.layers, ctx.wgrad, ctx.saved, ctx.c0 = layers, wgrad, saved, c0
        ctx.weights = [wb[2 * i] for i in range(n)]
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
        x_last, _, H, W = saved[n - 1]
        dy = ops.frame_nhwc_f16((g * (out > 0)).contiguous(), ops.framed_buffer(B, H, W, layers[n - 1][1], dev, BF))
        for i in range(n - 1, -1, -1):
            x_in, _, H, W = saved[i]
            c_in = ctx.c0 if i == 0 else layers[i - 1][1]
            grads[2 * i] = ctx.wgrad(x_in, dy, c_in)
            grads[2 * i + 1] = dy[:, 1:-1, 1:-1].float().sum((0, 1, 2))
            if i == 0:
                break
            zero_bias = torch.zeros(c_in, dtype=torch.float32, device=dev)
            dx = ops.conv3x3_f16(dy, _dgrad_weights(ctx.weights[i]), zero_bias, relu=False)      # framed bf16, c_in channels
            _, y_prev, Hp, Wp_ = saved[i - 1]
            if layers[i - 1][2]:                       # a pool sits between layer i - 1 and layer i
                dy = ops.maxpool2x2_bwd_bf16(y_prev, dx, ops.framed_buffer(B, Hp, Wp_, c_in, dev, BF))
            else:
                dy = dx.mul_(y_prev > 0)               # ReLU mask (the frame of both is zero)
        return (None, None, None) + tuple(grads)


def trunk(layers, x_nhwc, params, suffix, wgrad=wgrad_mfma):
    """conv1_1<suffix> .. conv5_3<suffix> of a TRAIN graph: params = {name: [w, b]} (fp32, OIHW)"""
    wb = []
    for stem, _, _ in layers:
        wb += list(params[stem + suffix])
    return TrunkFunction.apply(layers, wgrad, x_nhwc, *wb)
