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
dur = t[:, 7] - t[:, 0]
tpu = dur.max() / us            # the longest workgroup spans (nearly) the whole launch
print("workgroups %d, event time %.1f us, longest workgroup %d ticks -> ~%.0f ticks/us" % (len(t), us, dur.max(), tpu))
nl, ntl, view = t[:, 5], t[:, 6], t[:, 1]
d = dur / tpu
for k in sorted(set(view.tolist())):
    m = view == k
    print("view %d: %d workgroups, duration mean %.1f us (min %.1f max %.1f), tiles %.1f, ring entries %.0f (max %d)"
          % (k, m.sum(), d[m].mean(), d[m].min(), d[m].max(), ntl[m].mean(), nl[m].mean(), nl[m].max()))
o = np.argsort(d)
for i in list(o[:3]) + list(o[-6:]):
    print("   wg (view %d): %.1f us, tiles %d, entries %d" % (view[i], d[i], ntl[i], nl[i]))
print("correlation duration ~ entries: %.2f" % np.corrcoef(d, nl)[0, 1])
