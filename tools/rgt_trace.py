#!/usr/bin/env python3
"""Per-workgroup phase stamps of roi_grad_tile_kernel on the bench's training batch (library built with -DMV3D_TUNING):
start, after the ROI filter, geometry cycles, stream cycles, records listed, end.  Prints phase statistics by list length
and the launch's timeline (workgroups started / finished per microsecond)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mv3d_tf_amd import build, hot_path, synth
from mv3d_tf_amd._lib import RoiGradView, check, lib
from mv3d_tf_amd.fast_rcnn.config import apply_end2end_yml

build.build()
apply_end2end_yml()
np.random.seed(3)
dev = torch.device("cuda")
NB = 4
batches = []
for k in range(NB):
    frames = [synth.rpn_head(100000 + 2 * k + b, 76, 76, "peaky", return_gt=True) for b in range(2)]
    batches.append(hot_path.TrainPathBatch(frames, hot_path.synth_maps(2, k, dev), top_diff_seed=k).setup())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
views = tuple(os.environ.get("ONLY", "bev+rgb+fv").split("+"))


def call(bt):
    arr = (RoiGradView * len(views))()
    for k, v in enumerate(views):
        B, H, W, Cc = bt.maps[v].shape
        arr[k] = RoiGradView(bt.bottom_diff[v].data_ptr(), bt.rois[v].data_ptr(), bt.top_diff[v].data_ptr(),
                             bt.tops[v][1].data_ptr(), 0.125, B, bt.num_rois, H, W, Cc)
    return lambda: check(lib().mv3d_roi_pool_backward_views(len(views), arr, 7, 7, None, 0, st), "bwd")


fns = [call(b) for b in batches]
for f in fns:
    f()
torch.cuda.synchronize()
NBLK = 1 << 16
tr = torch.zeros(NBLK * 8, dtype=torch.int64, device=dev)
os.environ["MV3D_RT_TRACE"] = str(tr.data_ptr())
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
fns[1](); fns[2]()
torch.cuda.synchronize()
tr.zero_()
a.record(); fns[3](); b.record()
torch.cuda.synchronize()
us = a.elapsed_time(b) * 1e3
t = tr.cpu().numpy().reshape(-1, 8)
t = t[t[:, 0] != 0]
blk = np.nonzero(tr.cpu().numpy().reshape(-1, 8)[:, 0])[0]
for x in range(8):                      # the cycle counters of the 8 XCDs have their own origins: align each to its first stamp
    m = (blk & 7) == x
    if m.any():
        base = t[m][:, 0].min()
        t[m, 0] -= base; t[m, 1] -= base; t[m, 2] -= base; t[m, 7] -= base
t0 = 0
span = t[:, 7].max() - t0
print("workgroups %d, event time %.1f us, stamp span %d ticks -> %.1f ticks/us" % (len(t), us, span, span / us))
tpu = span / us
dur = (t[:, 7] - t[:, 0]) / tpu
g0 = (t[:, 1] - t[:, 0]) / tpu
geo, stre, wo, nl, nbt = t[:, 3] / tpu, t[:, 4] / tpu, (t[:, 7] - t[:, 2]) / tpu, t[:, 5], t[:, 6]
for lo, hi in ((0, 0), (1, 16), (17, 64), (65, 128), (129, 256), (257, 100000)):
    m = (nl >= lo) & (nl <= hi)
    if m.sum() == 0:
        continue
    print("records %4d..%-6d: %5d wg  total %6.2f us (max %6.2f)  roi filter %5.2f  geometry %5.2f  stream %6.2f (max %6.2f)  write-out %5.2f  batches %.1f"
          % (lo, hi, m.sum(), dur[m].mean(), dur[m].max(), g0[m].mean(), geo[m].mean(), stre[m].mean(), stre[m].max(), wo[m].mean(), nbt[m].mean()))
start = (t[:, 0] - t0) / tpu
end = (t[:, 7] - t0) / tpu
edges = np.arange(0, us + 5, 5)
print("timeline (5 us bins): started   ", np.histogram(start, edges)[0].tolist())
print("                      finished  ", np.histogram(end, edges)[0].tolist())
print("                      resident  ", [int(((start <= e) & (end > e)).sum()) for e in edges[:-1]])
late = np.argsort(end)[-8:]
for i in late:
    print("  late wg: start %.1f end %.1f records %d geo %.2f stream %.2f" % (start[i], end[i], nl[i], geo[i], stre[i]))
