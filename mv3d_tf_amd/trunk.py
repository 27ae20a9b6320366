"""The serving graph's VGG16 trunks on the hand-written MFMA convolution (csrc/conv3x3_mfma.hip).

`MfmaTrunks(net)` runs the 3x3 convolutions + 2x2 pools of lib/networks/MV3D_test.py:34-78 (conv1_1 .. conv5_3 of every view,
rpn_conv/3x3) through mv3d_conv3x3_f16 / mv3d_maxpool2x2_f16: f16 operands, f32 accumulation, activations kept in HBM as
framed NHWC f16 (one-pixel zero frame = the SAME padding, written once when a buffer is made).  Forward only, lower precision
than the reference's fp32 graph (BASELINE configs[4] "fp16 VGG16"): never the parity contract; the maps the hot-path layers
read (conv5_3*) are handed over as f32 NHWC, as those layers require.

conv1_1 (9 / 3 input channels) goes through the kernel's input-layer variant: channels zero-padded to 16 in the framed buffer,
a K step = 4 taps x 16 channels (3 steps instead of 9).
Packed weights are cached per layer and re-packed when the fp32 parameter changes (`tensor._version`).
"""
import torch

from . import ops


class MfmaTrunks:
    def __init__(self, net, vgg):
        self.net = net
        self.vgg = vgg                       # [(stem, c_out, pool_after)]
        self._w = {}                         # name -> (version, packed f16 weights, f32 bias)
        self._buf = {}                       # (tag, B, H, W, C) -> framed f16 buffer (frame stays zero: only interiors are written)

    def _packed(self, name, input_layer=False):
        w, b = self.net.params[name]
        ver = (w._version, b._version)
        hit = self._w.get(name)
        if hit is None or hit[0] != ver:
            pack = ops.pack_conv3x3_weights_input_layer if input_layer else ops.pack_conv3x3_weights
            hit = (ver, pack(w), b.detach().float().contiguous())
            self._w[name] = hit
        return hit[1], hit[2]

    def _framed(self, tag, B, H, W, C, dev):
        key = (tag, B, H, W, C)
        buf = self._buf.get(key)
        if buf is None:
            buf = self._buf[key] = ops.framed_buffer(B, H, W, C, dev)
        return buf

    def trunk(self, x_nhwc, suffix, last_framed):
        """x_nhwc (B, H, W, 9 | 3) f32 -> the trunk's conv5_3: framed f16 if `last_framed` (rpn_conv/3x3 reads it) else the bare
        f32 NHWC map.  Intermediate maps are left in net.layers as f16 NHWC views of their framed buffers."""
        B, H, W, c = x_nhwc.shape
        dev = x_nhwc.device
        L = self.net.layers
        x = ops.frame_nhwc_f16(x_nhwc.contiguous(), self._framed("in" + suffix, B, H, W, 16, dev))
        n = len(self.vgg)
        for i, (stem, cout, pool) in enumerate(self.vgg):
            name = stem + suffix
            wp, bias = self._packed(name, input_layer=(i == 0))
            if i == n - 1 and not last_framed:
                y = ops.conv3x3_f16(x, wp, bias, out_framed=False, out_f32=True)
                L[name] = y
                return y
            y = ops.conv3x3_f16(x, wp, bias, out=self._framed(name, B, H, W, cout, dev))
            L[name] = y[:, 1:-1, 1:-1]
            if pool:
                H, W = H // 2, W // 2
                y = ops.maxpool2x2_f16(y, out=self._framed(name + "/pool", B, H, W, cout, dev))
            x = y
        return x

    def rpn_conv(self, conv5_3_framed):
        """rpn_conv/3x3 (MV3D_test.py:82-84) on the framed BEV conv5_3 -> (B, H, W, 512) f16 NHWC"""
        wp, bias = self._packed("rpn_conv/3x3")
        return ops.conv3x3_f16(conv5_3_framed, wp, bias, out_framed=False, out_f32=False)
