#!/usr/bin/env python3
"""Kernel-only timing of the RoiPool launches on the bench's training batches (BASELINE configs[2] path: batch 2,
128 sampled ROIs / frame), per view and for all views, cycling over several resident batches so that consecutive
launches touch different maps / records.  `R=0` rows = the kernels' floor (zero fill + ROI filtering only)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mv3d_tf_amd import build, hot_path, synth
from mv3d_tf_amd._lib import RoiGradView, RoiView, check, lib
from mv3d_tf_amd.fast_rcnn.config import apply_end2end_yml

build.build()
apply_end2end_yml()
NB = int(os.environ.get("NB", "6"))
FR = int(os.environ.get("FRAMES", "2"))          # frames per batch
np.random.seed(3)
dev = torch.device("cuda")
batches = []
for k in range(NB):
    frames = [synth.rpn_head(100000 + 2 * k + b, 76, 76, "peaky", return_gt=True) for b in range(FR)]
    batches.append(hot_path.TrainPathBatch(frames, hot_path.synth_maps(FR, k, dev), top_diff_seed=k).setup())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
if os.environ.get("SHARE_MAPS"):                     # every batch reads the maps of batch 0 (outputs stay distinct)
    for b in batches[1:]:
        b.maps = batches[0].maps
if os.environ.get("SHARE_OUT"):                      # every batch writes the outputs of batch 0 (inputs stay distinct)
    for b in batches[1:]:
        if b.num_rois == batches[0].num_rois:
            b.tops, b.bottom_diff = batches[0].tops, batches[0].bottom_diff


def ev(fns, rounds=int(os.environ.get("ROUNDS", "8"))):
    for f in fns:
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(rounds):
        for f in fns:
            f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (rounds * len(fns)) * 1e3


def bwd_call(bt, views, R=None):
    arr = (RoiGradView * len(views))()
    for k, v in enumerate(views):
        m = bt.maps[v]
        B, H, W, Cc = m.shape
        arr[k] = RoiGradView(bt.bottom_diff[v].data_ptr(), bt.rois[v].data_ptr(), bt.top_diff[v].data_ptr(),
                             bt.tops[v][1].data_ptr(), 0.125, B, bt.num_rois if R is None else R, H, W, Cc)
    if os.environ.get("NOWS"):
        return lambda: check(lib().mv3d_roi_pool_backward_views(len(views), arr, 7, 7, None, 0, st), "bwd")
    ws = torch.zeros(lib().mv3d_roi_pool_backward_workspace_bytes(len(views), arr, 7, 7), dtype=torch.uint8, device=dev)
    return lambda: check(lib().mv3d_roi_pool_backward_views(len(views), arr, 7, 7, C.c_void_p(ws.data_ptr()), ws.numel(), st), "bwd")


def fwd_call(bt, views):
    arr = (RoiView * len(views))()
    for k, v in enumerate(views):
        m = bt.maps[v]
        B, H, W, Cc = m.shape
        arr[k] = RoiView(m.data_ptr(), bt.rois[v].data_ptr(), bt.tops[v][0].data_ptr(), bt.tops[v][1].data_ptr(), 0.125, B,
                         bt.num_rois, H, W, Cc)
    return lambda: check(lib().mv3d_roi_pool_forward_views(len(views), arr, 7, 7, st), "fwd")


R = batches[0].num_rois
CASES = (("bev",), ("rgb",), ("fv",), ("bev", "rgb", "fv"))
if os.environ.get("ONLY"):
    CASES = (tuple(os.environ["ONLY"].split("+")),)
for views in CASES:
    maps_b = sum(batches[0].maps[v].numel() * 4 for v in views)
    rec_b = sum(R * 49 * batches[0].maps[v].shape[3] * 8 for v in views)
    tf = ev([fwd_call(b, views) for b in batches])
    tb = ev([bwd_call(b, views) for b in batches])
    t0 = ev([bwd_call(b, views, 0) for b in batches])
    alg = maps_b + rec_b
    print("%-12s R=%d: fwd %6.1f us (%5.0f GB/s)  bwd %6.1f us (%5.0f GB/s)  bwd with R=0 %6.1f us (zero fill %5.0f GB/s)"
          % ("+".join(views), R, tf, alg / tf / 1e3, tb, alg / tb / 1e3, t0, maps_b / t0 / 1e3))
