"""Times mv3d_conv3x3_f16 on the trunk's layer shapes against torch's (MIOpen) f16 convolution: TFLOP/s per layer.
    python tools/conv_probe.py [batch] [--no-torch]"""
import sys
import time

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mv3d_tf_amd import _lib, build, ops  # noqa: E402

if "--lib" in sys.argv:                      # an experiment build of the library (tools only)
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
else:
    build.build()
ONLY = sys.argv[sys.argv.index("--only") + 1].split(",") if "--only" in sys.argv else None
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
TORCH = "--no-torch" not in sys.argv
F32 = "--f32" in sys.argv                    # the exact-f32 instantiation (f32 framed maps)
DT = torch.float32 if F32 else torch.float16
SHAPES = [("bev conv1_1", 608, 608, 9, 64), ("rgb conv1_1", 375, 1242, 3, 64), ("bev conv1_2", 608, 608, 64, 64), ("bev conv2_2", 304, 304, 128, 128), ("bev conv3_2", 152, 152, 256, 256),
          ("bev conv4_1", 76, 76, 256, 512), ("bev conv4_2", 76, 76, 512, 512),
          ("rgb conv1_2", 375, 1242, 64, 64), ("rgb conv2_2", 187, 621, 128, 128), ("rgb conv3_2", 93, 310, 256, 256),
          ("rgb conv4_2", 46, 155, 512, 512),
          # fixed cost per workgroup: the same tiles with 2x / 4x the K steps (time(2K) - time(K) = K steps of pure loop)
          ("fix c1_2 k18", 608, 608, 128, 64), ("fix c1_2 k36", 608, 608, 256, 64), ("fix c2_2 k36", 304, 304, 256, 128),
          ("fix c4_2 k36", 76, 76, 256, 512),
          # address-stride probes: the same layer with pixel strides that are not a power of two (1152 / 896 B instead of 1024)
          ("stride c4 cin576", 76, 76, 576, 512), ("stride c4 cin448", 76, 76, 448, 512), ("stride c3 cin320", 152, 152, 320, 256),
          ("stride c3 cin256", 152, 152, 256, 256)]



def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, H, W, cin, cout in SHAPES:
    if ONLY and not any(o in name for o in ONLY):
        continue
    x = torch.randn((B, H, W, cin), device="cuda")
    w = torch.randn((cout, cin, 3, 3), device="cuda") * 0.02
    b = torch.zeros(cout, device="cuda")
    cpad = (32 if F32 else 16) if cin < 16 else cin
    xf = ops.frame_nhwc_f16(x, ops.framed_buffer(B, H, W, cpad, "cuda", DT))
    wp = ops.pack_conv3x3_weights_input_layer(w) if (cin < 16 and not F32) else ops.pack_conv3x3_weights(w, cpad, dtype=DT)
    out = ops.framed_buffer(B, H, W, cout, "cuda", DT)
    ms = timed(lambda: ops.conv3x3_f16(xf, wp, b, out=out))
    fl = 2.0 * B * H * W * cout * 9 * cin
    if not TORCH:
        print("%-12s B=%d  mfma %.3f ms %.0f TF/s" % (name, B, ms, fl / ms / 1e9), flush=True)
        continue
    xh = x.half().permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
    wh = w.half().contiguous(memory_format=torch.channels_last)
    bh = b.half()
    ms_t = timed(lambda: torch.relu_(torch.nn.functional.conv2d(xh, wh, bh, padding=1)))
    xn, wn = x.half().permute(0, 3, 1, 2).contiguous(), w.half()
    ms_n = timed(lambda: torch.relu_(torch.nn.functional.conv2d(xn, wn, bh, padding=1)))
    fl = 2.0 * B * H * W * cout * 9 * cin
    print("%-12s B=%d  mfma %.3f ms %.0f TF/s | torch nhwc %.3f ms %.0f TF/s | torch nchw %.3f ms %.0f TF/s" % (
        name, B, ms, fl / ms / 1e9, ms_t, fl / ms_t / 1e9, ms_n, fl / ms_n / 1e9), flush=True)
    del x, xf, out, xh, xn
