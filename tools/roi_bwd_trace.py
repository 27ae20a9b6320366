#!/usr/bin/env python3
"""Per-wave cycle trace of roi_bwd_gather_kernel on one bench batch (MV3D_BWD_TRACE diagnostics hook)."""
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
views = tuple(os.environ.get("ONLY", "bev+rgb+fv").split("+"))
frames = [synth.rpn_head(100000 + b, 76, 76, "peaky", return_gt=True) for b in range(2)]
bt = hot_path.TrainPathBatch(frames, hot_path.synth_maps(2, 0, dev)).setup()
groups = int(os.environ.get("MV3D_BWG_GROUPS", "256"))
nw = groups * 8 * 4
trace = torch.zeros((nw, 8), dtype=torch.int64, device=dev)
arr = (RoiGradView * len(views))()
for k, v in enumerate(views):
    m = bt.maps[v]
    B, H, W, Cc = m.shape
    arr[k] = RoiGradView(bt.bottom_diff[v].data_ptr(), bt.rois[v].data_ptr(), bt.top_diff[v].data_ptr(), bt.tops[v][1].data_ptr(),
                         0.125, B, bt.num_rois, H, W, Cc)
ws = torch.zeros(lib().mv3d_roi_pool_backward_workspace_bytes(len(views), arr, 7, 7), dtype=torch.uint8, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
call = lambda: check(lib().mv3d_roi_pool_backward_views(len(views), arr, 7, 7, C.c_void_p(ws.data_ptr()), ws.numel(), st), "bwd")
for _ in range(3):
    call()
torch.cuda.synchronize()
os.environ["MV3D_BWD_TRACE"] = str(trace.data_ptr())
call()
torch.cuda.synchronize()
t = trace.cpu().numpy()
act = t[t[:, 5] > 0]
t0 = t[:, 0][t[:, 0] > 0].min()
print("waves launched %d, with items %d; clock = s_memtime ticks" % ((t[:, 0] > 0).sum(), len(act)))
q = lambda a: "min %d p50 %d p90 %d p99 %d max %d" % tuple(np.percentile(a, [0, 50, 90, 99, 100]).astype(np.int64))
print("start - t0        :", q(act[:, 0] - t0))
print("items loaded      :", q(act[:, 1] - act[:, 0]))
print("offsets loaded    :", q(act[:, 2] - act[:, 1]))
print("first item done   :", q(act[:, 3] - act[:, 2]))
print("end - start       :", q(act[:, 4] - act[:, 0]))
print("end - t0          :", q(act[:, 4] - t0))
print("items per wave    :", q(act[:, 5]), " candidates per wave:", q(act[:, 6]))
slow = act[np.argsort(act[:, 4] - act[:, 0])[-5:]]
for r in slow:
    print("  slow wave: dur %d items %d cands %d (start+%d)" % (r[4] - r[0], r[5], r[6], r[0] - t0))
