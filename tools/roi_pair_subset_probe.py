#!/usr/bin/env python3
"""RoiPoolGrad of the pair on SUBSETS of the bench's three views (bev, rgb, fv), with a workspace (index + gather) and without one
(the one-launch tile kernel): which view costs what in which structure.  Kernel-only, cold batches, HIP events around the call."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mv3d_tf_amd import _lib, build, hot_path, synth
from mv3d_tf_amd._lib import RoiGradView, RoiView, check, lib
from mv3d_tf_amd.fast_rcnn.config import apply_end2end_yml

if "--lib" in sys.argv:
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
else:
    build.build()
apply_end2end_yml()
NB, ROUNDS = int(os.environ.get("NB", "10")), int(os.environ.get("ROUNDS", "5"))
np.random.seed(3)
dev = torch.device("cuda")
L = lib()
batches = []
for k in range(NB):
    frames = [synth.rpn_head(100000 + 2 * k + b, 76, 76, "peaky", return_gt=True) for b in range(2)]
    batches.append(hot_path.TrainPathBatch(frames, hot_path.synth_maps(2, k, dev), top_diff_seed=k).setup())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
VIEWS = hot_path.VIEWS


def views_of(bt, names):
    fwd, bwd = (RoiView * len(names))(), (RoiGradView * len(names))()
    for k, v in enumerate(names):
        m = bt.maps[v]
        B, H, W, Cc = m.shape
        fwd[k] = RoiView(m.data_ptr(), bt.rois[v].data_ptr(), bt.tops[v][0].data_ptr(), bt.tops[v][1].data_ptr(), 0.125, B, bt.num_rois, H, W, Cc)
        bwd[k] = RoiGradView(bt.bottom_diff[v].data_ptr(), bt.rois[v].data_ptr(), bt.top_diff[v].data_ptr(), bt.tops[v][1].data_ptr(), 0.125, B,
                             bt.num_rois, H, W, Cc)
    return fwd, bwd


all_fwd = [views_of(b, VIEWS) for b in batches]
for f, _ in all_fwd:                                   # the pair's codes for every batch
    check(L.mv3d_roi_pool_forward_views_pair(3, f, 7, 7, 1, st), "fwd")
torch.cuda.synchronize()
for names in (("bev", "rgb", "fv"), ("bev", "rgb"), ("fv",), ("bev",), ("rgb",), ("rgb", "bev")):
    for no_ws in (False, True):
        cs = [views_of(b, names)[1] for b in batches]
        ws = torch.zeros(L.mv3d_roi_pool_pair_workspace_bytes(len(names), cs[0], 7, 7), dtype=torch.uint8, device=dev)
        wp, wn = (None, 0) if no_ws else (C.c_void_p(ws.data_ptr()), ws.numel())
        call = lambda a: check(L.mv3d_roi_pool_backward_views_pair(len(names), a, 7, 7, wp, wn, st), "bwd")
        for a in cs:
            call(a)
        torch.cuda.synchronize()
        tot, n = 0.0, 0
        for _ in range(ROUNDS):
            evs = []
            for a in cs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); call(a); e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            for e0, e1 in evs:
                tot += e0.elapsed_time(e1); n += 1
        print("%-14s %-22s %6.1f us" % ("+".join(names), "tiles (no workspace)" if no_ws else "index + gather", tot / n * 1e3), flush=True)
