"""Grouped (three views, one launch) against per-view launches of the training trunk's kernels, per VGG depth:
    python tools/group_probe.py [bf16|f32] [batch]      (forward convolution and weight gradient + reduce)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mv3d_tf_amd import _lib, build, ops  # noqa: E402

if "--lib" in sys.argv:                      # an experiment build of the library (tools only)
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
else:
    build.build()
DT = torch.float32 if len(sys.argv) > 1 and sys.argv[1] == "f32" else torch.bfloat16
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
VIEWS = [("bev", 608, 608), ("rgb", 375, 1242), ("fv", 64, 512)]
DEPTHS = [("conv1_2", 64, 64, 1), ("conv2_2", 128, 128, 2), ("conv3_2", 256, 256, 4), ("conv4_1", 256, 512, 8), ("conv4_2", 512, 512, 8)]
dev = "cuda"


def timed(fn, reps=8):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, cin, cout, div in DEPTHS:
    xs, dys, ws, outs, fl = [], [], [], [], 0.0
    for _, H, W in VIEWS:
        h, w = H // div, W // div
        x = ops.framed_buffer(B, h, w, cin, dev, DT)
        x[:, 1:-1, 1:-1] = torch.randn((B, h, w, cin), device=dev, dtype=DT)
        dy = ops.framed_buffer(B, h, w, cout, dev, DT)
        dy[:, 1:-1, 1:-1] = torch.randn((B, h, w, cout), device=dev, dtype=DT)
        wp = ops.pack_conv3x3_train_many([(torch.randn((cout, cin, 3, 3), device=dev) * 0.05, None, False)], dtype=DT)[0][0]
        xs.append(x); dys.append(dy); ws.append(wp); outs.append(ops.framed_buffer(B, h, w, cout, dev, DT))
        fl += 2.0 * B * h * w * cout * 9 * cin
    bias = torch.zeros(cout, device=dev)
    t_sep = timed(lambda: [ops.conv3x3_f16(xs[v], ws[v], bias, out=outs[v]) for v in range(3)])
    t_grp = timed(lambda: ops.conv3x3_views([(xs[v], ws[v], bias, None, outs[v]) for v in range(3)]))
    w_sep = timed(lambda: [ops.conv3x3_wgrad_bf16(xs[v], dys[v], want_bias=True) for v in range(3)])
    w_grp = timed(lambda: ops.conv3x3_wgrad_views([(xs[v], dys[v]) for v in range(3)], want_bias=True))
    print("%-8s %s B=%d  conv: 3 launches %.3f ms (%.0f TF/s), grouped %.3f ms (%.0f TF/s) | wgrad+reduce: 3 x 2 launches %.3f ms (%.0f), grouped %.3f ms (%.0f)"
          % (name, str(DT).split(".")[-1], B, t_sep, fl / t_sep / 1e9, t_grp, fl / t_grp / 1e9, w_sep, fl / w_sep / 1e9, w_grp, fl / w_grp / 1e9), flush=True)
