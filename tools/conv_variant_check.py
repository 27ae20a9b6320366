"""One 3x3 convolution variant of an experiment build (MV3D_CONV_TILE / MV3D_CONV_BIG_MIN / MV3D_CONV_PP in the environment): a hash of the
output (variants that keep the k order are bit-identical) and the time per layer.
    python tools/conv_variant_check.py --lib build_variants/libmv3d_tuning.so [batch] [--bf16]"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mv3d_tf_amd import _lib, build, ops  # noqa: E402

if "--lib" in sys.argv:
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
else:
    build.build()
args = [a for a in sys.argv[1:] if a.isdigit()]
B = int(args[0]) if args else 16
DT = torch.bfloat16 if "--bf16" in sys.argv else torch.float16
SHAPES = [("bev conv3_2", 152, 152, 256, 256), ("bev conv4_1", 76, 76, 256, 512), ("bev conv4_2", 76, 76, 512, 512),
          ("rgb conv3_2", 93, 310, 256, 256), ("rgb conv4_2", 46, 155, 512, 512), ("fv conv3_2", 16, 128, 256, 256)]


def timed(fn, n=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


torch.manual_seed(7)
for name, H, W, cin, cout in SHAPES:
    x = torch.randn((B, H, W, cin), device="cuda")
    w = torch.randn((cout, cin, 3, 3), device="cuda") * 0.02
    b = torch.randn(cout, device="cuda") * 0.1
    xf = ops.frame_nhwc_f16(x, ops.framed_buffer(B, H, W, cin, "cuda", DT))
    wp = ops.pack_conv3x3_weights(w, cin, dtype=DT)
    out = ops.framed_buffer(B, H, W, cout, "cuda", DT)
    ms = timed(lambda: ops.conv3x3_f16(xf, wp, b, out=out))
    fl = 2.0 * B * H * W * cout * 9 * cin
    h = hashlib.sha1(out.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:12]
    print("%-12s B=%d  %.3f ms %.0f TF/s  sha1 %s" % (name, B, ms, fl / ms / 1e9, h), flush=True)
