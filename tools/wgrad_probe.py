"""Times mv3d_conv3x3_wgrad_bf16 on the training trunk's layer shapes (kernel + split-K reduce):  python tools/wgrad_probe.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mv3d_tf_amd import _lib, build, ops  # noqa: E402

if "--lib" in sys.argv:
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
else:
    build.build()
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2
BF = torch.float32 if "--f32" in sys.argv else torch.bfloat16
SHAPES = [("bev conv1_2", 608, 608, 64, 64), ("bev conv2_2", 304, 304, 128, 128), ("bev conv3_2", 152, 152, 256, 256),
          ("bev conv4_1", 76, 76, 256, 512), ("bev conv4_2", 76, 76, 512, 512), ("rgb conv1_2", 375, 1242, 64, 64),
          ("rgb conv3_2", 93, 310, 256, 256), ("rgb conv4_2", 46, 155, 512, 512)]
for name, H, W, cin, cout in SHAPES:
    x = ops.framed_buffer(B, H, W, cin, "cuda", BF)
    x[:, 1:-1, 1:-1] = torch.randn((B, H, W, cin), device="cuda", dtype=BF)
    dy = ops.framed_buffer(B, H, W, cout, "cuda", BF)
    dy[:, 1:-1, 1:-1] = torch.randn((B, H, W, cout), device="cuda", dtype=BF)
    ops.conv3x3_wgrad_bf16(x, dy)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv3x3_wgrad_bf16(x, dy)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 2.0 * B * H * W * cout * 9 * cin
    print("%-12s B=%d  wgrad %.3f ms %.0f TF/s" % (name, B, ms, fl / ms / 1e9), flush=True)
