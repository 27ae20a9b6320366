#!/usr/bin/env python3
"""The work list the pair's forward launch writes for its backward (csrc/roi_grad_plan.h), read back from the unused quarter of view 0's argmax
buffer on the bench's training batches: units, sub-tile units, and the distribution of the planner's per-tile entry estimates."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mv3d_tf_amd import _lib, build, hot_path, synth
from mv3d_tf_amd._lib import RoiView, check, lib
from mv3d_tf_amd.fast_rcnn.config import apply_end2end_yml

if "--lib" in sys.argv:
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
else:
    build.build()
apply_end2end_yml()
np.random.seed(3)
dev = torch.device("cuda")
L = lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for k in range(int(os.environ.get("NB", "4"))):
    frames = [synth.rpn_head(100000 + 2 * k + b, 76, 76, "peaky", return_gt=True) for b in range(2)]
    bt = hot_path.TrainPathBatch(frames, hot_path.synth_maps(2, k, dev), top_diff_seed=k).setup()
    fwd = (RoiView * 3)()
    for i, v in enumerate(hot_path.VIEWS):
        m = bt.maps[v]
        B, H, W, Cc = m.shape
        fwd[i] = RoiView(m.data_ptr(), bt.rois[v].data_ptr(), bt.tops[v][0].data_ptr(), bt.tops[v][1].data_ptr(), 0.125, B, bt.num_rois, H, W, Cc)
    check(L.mv3d_roi_pool_forward_views_pair(3, fwd, 7, 7, 1, st), "fwd")
    torch.cuda.synchronize()
    a = bt.tops[hot_path.VIEWS[0]][1]
    n0 = a.numel()                                   # int32 words = bytes of one quarter
    raw = a.view(torch.uint8).reshape(-1)[3 * n0:].cpu().numpy()
    n_work = int(raw[:4].view(np.int32)[0])
    units = raw[256:256 + 16 * n_work].view(np.int32).reshape(-1, 4)
    heat = units[:, 3]
    shape = (units[:, 0] >> 16) & 0xff                                  # ths | tws << 4
    view_of = units[:, 0] & 15
    # a view's whole tiles have its largest shape; sub-tile units (four per cut tile, first in the list) a smaller one
    full = {v: max(shape[view_of == v], key=lambda x: (x & 15) + (x >> 4)) for v in set(view_of.tolist())}
    is_sub = np.array([shape[i] != full[view_of[i]] for i in range(n_work)])
    sub = int(is_sub.sum())
    empty = int(((units[:, 0] >> 25) & 1).sum())
    print(f"batch {k}: units {n_work}, sub-tile units {sub} ({sub // 4} tiles cut), flagged empty {empty}; estimates along the list: "
          f"{heat[::max(n_work // 12, 1)].tolist()}")
    h = np.concatenate([heat[is_sub][::4], heat[~is_sub]])
    hv_view = np.concatenate([view_of[is_sub][::4], view_of[~is_sub]])
    for view in range(3):
        hv = h[hv_view == view]
        print(f"   launch view {view}: tiles {hv.size}, estimate = 0: {(hv == 0).sum()}, >= 100: {(hv >= 100).sum()}, >= 160: {(hv >= 160).sum()}, >= 250: {(hv >= 250).sum()}, "
              f">= 350: {(hv >= 350).sum()}, >= 450: {(hv >= 450).sum()}, max {hv.max()}")
