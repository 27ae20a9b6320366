#!/usr/bin/env python3
"""A / B of the RoiPool pair on the bench's training batches (BASELINE configs[2] path: batch 2, 128 sampled ROIs / frame, three
views), kernel-only, cycling over NB resident batches (distinct maps / records, so consecutive launches are cold):

    plain    mv3d_roi_pool_forward_views_cold + mv3d_roi_pool_backward_views (workspace: index x 2 launches + gather)
    pair     mv3d_roi_pool_forward_views_pair (cold; 16-bit argmax codes) + mv3d_roi_pool_backward_views_pair (index + fill, gather)

forward and backward are timed alternating (fwd b0, bwd b0, fwd b1, ...) -- the order a step runs them in -- with HIP events
around every call, and back to back per kind."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mv3d_tf_amd import _lib, build, hot_path, synth
from mv3d_tf_amd._lib import RoiGradView, RoiView, check, lib
from mv3d_tf_amd.fast_rcnn.config import apply_end2end_yml

if "--lib" in sys.argv:                      # an experiment build of the library (tools only)
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
else:
    build.build()
apply_end2end_yml()
NB = int(os.environ.get("NB", "12"))
ROUNDS = int(os.environ.get("ROUNDS", "6"))
np.random.seed(3)
dev = torch.device("cuda")
L = lib()
batches = []
for k in range(NB):
    frames = [synth.rpn_head(100000 + 2 * k + b, 76, 76, "peaky", return_gt=True) for b in range(2)]
    batches.append(hot_path.TrainPathBatch(frames, hot_path.synth_maps(2, k, dev), top_diff_seed=k).setup())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
VIEWS = hot_path.VIEWS


def calls(bt, indexed):
    fwd, bwd = (RoiView * 3)(), (RoiGradView * 3)()
    for k, v in enumerate(VIEWS):
        m = bt.maps[v]
        B, H, W, Cc = m.shape
        fwd[k] = RoiView(m.data_ptr(), bt.rois[v].data_ptr(), bt.tops[v][0].data_ptr(), bt.tops[v][1].data_ptr(), 0.125, B, bt.num_rois, H, W, Cc)
        bwd[k] = RoiGradView(bt.bottom_diff[v].data_ptr(), bt.rois[v].data_ptr(), bt.top_diff[v].data_ptr(), bt.tops[v][1].data_ptr(), 0.125, B,
                             bt.num_rois, H, W, Cc)
    ws = torch.zeros(L.mv3d_roi_pool_pair_workspace_bytes(3, bwd, 7, 7), dtype=torch.uint8, device=dev)
    wp, wn = C.c_void_p(ws.data_ptr()), ws.numel()
    if os.environ.get("PAIR_NO_WS"):            # the one-launch tile RoiPoolGrad (what the path runs)
        wp, wn = None, 0
    keep = (fwd, bwd, ws)
    if indexed:
        return (lambda: check(L.mv3d_roi_pool_forward_views_pair(3, fwd, 7, 7, 1, st), "fwd"),
                lambda: check(L.mv3d_roi_pool_backward_views_pair(3, bwd, 7, 7, wp, wn, st), "bwd"), keep)
    return (lambda: check(L.mv3d_roi_pool_forward_views_cold(3, fwd, 7, 7, st), "fwd"),
            lambda: check(L.mv3d_roi_pool_backward_views(3, bwd, 7, 7, wp, wn, st), "bwd"), keep)


def run(indexed):
    cs = [calls(b, indexed) for b in batches]
    for f, b, _ in cs:
        f(); b()
    torch.cuda.synchronize()
    tf = tb = 0.0
    n = 0
    for _ in range(ROUNDS):
        evs = []
        for f, b, _ in cs:
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record(); f(); e[1].record(); b(); e[2].record()
            evs.append(e)
        torch.cuda.synchronize()
        for e in evs:
            tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2]); n += 1
    # back to back per kind
    def b2b(k):
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(ROUNDS):
            for c in cs:
                c[k]()
        z.record()
        torch.cuda.synchronize()
        return a.elapsed_time(z) / (ROUNDS * len(cs)) * 1e3
    return tf / n * 1e3, tb / n * 1e3, b2b(0), b2b(1)


want = None
ORDER = (("pair", True),) * 2 if os.environ.get("PAIR_ONLY") else (("plain", False), ("pair", True), ("plain", False), ("pair", True))
for name, indexed in ORDER:
    f, b, f2, b2 = run(indexed)
    print("%-8s alternating: fwd %6.1f us  bwd %6.1f us  sum %6.1f   back-to-back: fwd %6.1f  bwd %6.1f" % (name, f, b, f + b, f2, b2), flush=True)
    snap = [batches[0].bottom_diff[v].clone() for v in VIEWS] + [batches[0].tops[v][0].clone() for v in VIEWS]
    if want is None:
        want = snap
    elif not os.environ.get("MV3D_IDX_DBG"):
        assert all(torch.equal(a, b_) for a, b_ in zip(want, snap)), "plain and pair outputs differ"
print("outputs of both pairs bit-identical on batch 0")
