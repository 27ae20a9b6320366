#!/usr/bin/env python3
"""Per-wave wall-clock stamps (100 MHz) of the pair's tile RoiPoolGrad (roi_pair_tiles_kernel) on one bench batch (experiment build,
MV3D_RGT_TRACE): 0 start, 1 first 64 ROI rows filtered, 3 expansion + full drains done, 4 last drain done, 5 end; word 6 = ring
entries of the wave's tile."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mv3d_tf_amd import _lib, hot_path, synth
from mv3d_tf_amd._lib import RoiGradView, RoiView, check, lib
from mv3d_tf_amd.fast_rcnn.config import apply_end2end_yml

_lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
apply_end2end_yml()
np.random.seed(3)
dev = torch.device("cuda")
NB = 6
bts = []
for k in range(NB):
    frames = [synth.rpn_head(100000 + 2 * k + b, 76, 76, "peaky", return_gt=True) for b in range(2)]
    bts.append(hot_path.TrainPathBatch(frames, hot_path.synth_maps(2, k, dev), top_diff_seed=k).setup())
views = hot_path.VIEWS
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def mk(bt):
    fwd, arr = (RoiView * 3)(), (RoiGradView * 3)()
    for k, v in enumerate(views):
        m = bt.maps[v]
        B, H, W, Cc = m.shape
        fwd[k] = RoiView(m.data_ptr(), bt.rois[v].data_ptr(), bt.tops[v][0].data_ptr(), bt.tops[v][1].data_ptr(), 0.125, B, bt.num_rois, H, W, Cc)
        arr[k] = RoiGradView(bt.bottom_diff[v].data_ptr(), bt.rois[v].data_ptr(), bt.top_diff[v].data_ptr(), bt.tops[v][1].data_ptr(), 0.125, B,
                             bt.num_rois, H, W, Cc)
    return fwd, arr


cs = [mk(b) for b in bts]
ws = torch.zeros(lib().mv3d_roi_pool_pair_workspace_bytes(3, cs[0][1], 7, 7), dtype=torch.uint8, device=dev)
for fwd, arr in cs:
    check(lib().mv3d_roi_pool_forward_views_pair(3, fwd, 7, 7, 1, st), "fwd")
bwd = lambda arr: check(lib().mv3d_roi_pool_backward_views_pair(3, arr, 7, 7, C.c_void_p(ws.data_ptr()) if os.environ.get("PAIR_WS") else None, ws.numel() if os.environ.get("PAIR_WS") else 0, st), "bwd")
for _ in range(2):
    for fwd, arr in cs:
        bwd(arr)
torch.cuda.synchronize()
NW = 1 << 17
trace = torch.zeros((NW, 8), dtype=torch.int64, device=dev)
os.environ["MV3D_RGT_TRACE"] = str(trace.data_ptr())
bwd(cs[0][1])                                    # cold: five other batches went through since
torch.cuda.synchronize()
t = trace.cpu().numpy()
t = t[t[:, 0] != 0]
t0 = t[:, 0].min()
q = lambda a: "min %6d p50 %6d p90 %6d max %6d  (x10 ns)" % tuple(np.percentile(a, [0, 50, 90, 100]).astype(np.int64))
print("waves %d, kernel span %d ticks of 10 ns" % (len(t), t[:, 5].max() - t0))
live = t[:, 5] != 0
t = t[live]
print("   start - t0       :", q(t[:, 0] - t0))
print("   ROI rows filtered:", q(t[:, 1] - t[:, 0]))
print("   expand + drains  :", q(t[:, 3] - t[:, 1]))
print("   last drain       :", q(t[:, 4] - t[:, 3]))
print("   write-out        :", q(t[:, 5] - t[:, 4]))
print("   whole wave       :", q(t[:, 5] - t[:, 0]))
print("   end - t0         :", q(t[:, 5] - t0))
print("   ring entries     :", q(t[:, 6]), " total", t[:, 6].sum())
ne = t[t[:, 6] > 0]
if int(os.environ.get("MV3D_RGT_DBG", "0")) & 16:     # the per-pixel streams kernel: 1 = pass loop done; words 2 listing, 7 adding, 3 ranges, 4 pixel loop (with flushes), 6 entries
    whole = t[:, 5] - t[:, 0]
    loop = t[:, 1] - t[:, 0]
    print("   whole            :", q(whole), " total", whole.sum())
    print("   pass loop        :", q(loop), " total", loop.sum())
    print("   write-out        :", q(t[:, 5] - t[:, 1]), " total", (t[:, 5] - t[:, 1]).sum())
    print("   ranges           :", q(t[:, 3]), " total", t[:, 3].sum())
    print("   pixel loop       :", q(t[:, 4]), " total", t[:, 4].sum())
    print("     listing        :", q(t[:, 2]), " total", t[:, 2].sum())
    print("     adding         :", q(t[:, 7]), " total", t[:, 7].sum())
    print("   filter (rest)    :", q(loop - t[:, 3] - t[:, 4]), " total", (loop - t[:, 3] - t[:, 4]).sum())
    print("   entries          :", q(t[:, 6]), " total", t[:, 6].sum())
    hot = np.argsort(-whole)[:10]
    print("the 10 longest waves: entries, start, whole, pass loop, ranges, pixel loop, listing, adding")
    for i in hot:
        print("   entries %5d  start %5d  whole %5d  loop %5d ranges %5d  pixels %5d  listing %5d  adding %5d" % (t[i, 6], t[i, 0] - t0, whole[i], loop[i], t[i, 3], t[i, 4], t[i, 2], t[i, 7]))
    span = int(t[:, 5].max() - t0)
    alive = np.zeros(span + 1, dtype=np.int64)
    for a_, b_ in zip(t[:, 0] - t0, t[:, 5] - t0):
        alive[a_:b_ + 1] += 1
    step = max(span // 20, 1)
    print("waves alive over time (every %d ticks):" % step, alive[::step].tolist())
    sys.exit(0)
n_own, n_other = t[:, 2] & 0xffffffff, t[:, 2] >> 32
t_room, t_turn = t[:, 7] & 0xffffffff, t[:, 7] >> 32
print("   groups added on the own stream %d, on other waves' streams %d" % (n_own.sum(), n_other.sum()))
hot = np.argsort(-t[:, 6])[:12]
print("the 12 fullest streams: entries, start, expansion phase (of which waiting for room in the ring), help phase; groups this wave added own / others, ticks waiting for turns")
for i in hot:
    print("   %4d  start %5d  expansion %5d (room %5d)  help %5d  groups own %3d other %3d  turns %5d" % (t[i, 6], t[i, 0] - t0, t[i, 3] - t[i, 1], t_room[i], t[i, 4] - t[i, 3], n_own[i], n_other[i], t_turn[i]))
print("waves with entries: %d" % len(ne))
print("   expand + drains  :", q(ne[:, 3] - ne[:, 1]))
print("   last drain       :", q(ne[:, 4] - ne[:, 3]))
print("   whole wave       :", q(ne[:, 5] - ne[:, 0]))
# occupancy over time: waves alive per microsecond
span = int(t[:, 5].max() - t0)
alive = np.zeros(span + 1, dtype=np.int64)
for a, b in zip(t[:, 0] - t0, t[:, 5] - t0):
    alive[a:b + 1] += 1
step = max(span // 20, 1)
print("waves alive over time (every %d ticks):" % step, alive[::step].tolist())
