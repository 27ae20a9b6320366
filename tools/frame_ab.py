import sys, os
sys.path.insert(0, "/root/repo")
import torch
from mv3d_tf_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from mv3d_tf_amd import ops
for B, H, W, C in ((16, 608, 608, 9), (16, 375, 1242, 3), (16, 64, 512, 3)):
    x = torch.randn((B, H, W, C), device="cuda")
    out = ops.framed_buffer(B, H, W, 16, "cuda", torch.float16)
    for _ in range(3): ops.frame_nhwc_f16(x, out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): ops.frame_nhwc_f16(x, out)
    b.record(); torch.cuda.synchronize()
    print(sys.argv[1].split("/")[-1], (B, H, W, C), "%.1f us" % (a.elapsed_time(b) / 20 * 1e3))
