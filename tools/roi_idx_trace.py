#!/usr/bin/env python3
"""Per-workgroup cycle stamps of the pair's LISTS launch (roi_pair_index_kernel<true>) on one bench batch (experiment build,
MV3D_IDX_TRACE): stamps 0 start, 1 fill issued, 2 prefix words arrived, 3 filter + counts done, 4 end; word 7 = entries of the list."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mv3d_tf_amd import _lib, hot_path, synth
from mv3d_tf_amd._lib import RoiGradView, check, lib
from mv3d_tf_amd.fast_rcnn.config import apply_end2end_yml

_lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
apply_end2end_yml()
np.random.seed(3)
dev = torch.device("cuda")
frames = [synth.rpn_head(100000 + b, 76, 76, "peaky", return_gt=True) for b in range(2)]
bt = hot_path.TrainPathBatch(frames, hot_path.synth_maps(2, 0, dev)).setup()
views = hot_path.VIEWS
arr = (RoiGradView * 3)()
nseg = []
for k, v in enumerate(views):
    m = bt.maps[v]
    B, H, W, Cc = m.shape
    arr[k] = RoiGradView(bt.bottom_diff[v].data_ptr(), bt.rois[v].data_ptr(), bt.top_diff[v].data_ptr(), bt.tops[v][1].data_ptr(), 0.125, B,
                         bt.num_rois, H, W, Cc)
    nseg.append((v, B * H * ((W + 15) // 16), B * H * W))
ws = torch.zeros(lib().mv3d_roi_pool_pair_workspace_bytes(3, arr, 7, 7), dtype=torch.uint8, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
call = lambda: check(lib().mv3d_roi_pool_backward_views_pair(3, arr, 7, 7, C.c_void_p(ws.data_ptr()), ws.numel(), st), "bwd")
for _ in range(3):
    call()
torch.cuda.synchronize()
total = sum(n for _, n, _ in nseg)
trace = torch.zeros((total, 8), dtype=torch.int64, device=dev)
os.environ["MV3D_IDX_TRACE"] = str(trace.data_ptr())
call()
torch.cuda.synchronize()
t = trace.cpu().numpy()
t0 = t[:, 0].min()
q = lambda a: "min %6d p50 %6d p90 %6d max %6d" % tuple(np.percentile(a, [0, 50, 90, 100]).astype(np.int64))
print("clock = s_memtime ticks (100 MHz => 10 ns each?) ; kernel span %d ticks" % (t[:, 4].max() - t0))
# the index orders the views densest first: fv, then bev / rgb by rows per pixel
order = sorted(nseg, key=lambda x: -bt.num_rois / x[2])
o = 0
for v, n, _ in order:
    r = t[o:o + n]
    o += n
    print("== %s: %d segments, list entries %s" % (v, n, q(r[:, 7])))
    print("   start - t0       :", q(r[:, 0] - t0))
    print("   fill issued      :", q(r[:, 1] - r[:, 0]))
    print("   prefix arrived   :", q(r[:, 2] - r[:, 1]))
    print("   filter + counts  :", q(r[:, 3] - r[:, 2]))
    print("   items + lists    :", q(r[:, 4] - r[:, 3]))
    print("   end - t0         :", q(r[:, 4] - t0))
