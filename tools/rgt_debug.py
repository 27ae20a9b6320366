#!/usr/bin/env python3
"""The one-launch RoiPoolGrad (pair without a workspace) against the oracle on the malformed / overhanging / huge ROI case of
tests/test_roi_pair.py, with a mismatch report: differing pixels, and which single ROIs reproduce a difference."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mv3d_tf_amd import build, ops
from oracle import oracle

build.build()
C = int(os.environ.get("C", "256"))
rs = np.random.RandomState(C)
B, H, W = 2, 70, 130
m = rs.uniform(-1, 1, (B, H, W, C)).astype(np.float32)
m += (np.arange(W, dtype=np.float32)[None, None, :, None] + np.arange(H, dtype=np.float32)[None, :, None, None]) * np.float32(4)
rois = [[0, 0, 0, 448, 100], [0, 100, 50, 40, 80], [1, 100, 50, 140, 20], [1, 40, 80, 30, 10],
        [0, 8, 8, 8 + 113 * 8, 8 + 56 * 8], [1, 16, 0, 16 + 120 * 8, 456], [0, 0, 16, 500, 16 + 56 * 8], [1, 24, 24, 24 + 56 * 8, 24 + 56 * 8],
        [0, -20000, -20000, 20000, 20000], [1, -100, -40000, 300, 40000]]
for _ in range(60):
    x1, y1 = rs.randint(-40, W * 8), rs.randint(-40, H * 8)
    rois.append([rs.randint(0, B), x1, y1, x1 + rs.choice([56 * 8, 113 * 8, 120 * 8, -30, 200]), y1 + rs.choice([56 * 8, -20, 90])])
rois = np.asarray(rois, np.float32)
G_TEST = rs.uniform(-1, 1, (len(rois), 7, 7, C)).astype(np.float32)          # the test's top_diff (same generator position)
dev = lambda a: torch.as_tensor(a).cuda()


def run(sel):
    r = rois[sel]
    d, rr = dev(m), dev(r)
    res = ops.roi_pool_forward_views_pair([(d, rr, 0.125)], 7, 7)
    o_top, o_am = oracle.roi_pool(m, r, 7, 7, 0.125)
    g = G_TEST[sel] if os.environ.get("G_TEST", "1") == "1" else np.random.RandomState(1).uniform(-1, 1, o_top.shape).astype(np.float32)
    want = oracle.roi_pool_grad(m, r, o_am, g, 7, 7, 0.125)
    got = ops.roi_pool_backward_views_pair([(dev(g), rr, res[0][1], m.shape, 0.125)], 7, 7, workspace=False)[0].cpu().numpy()
    return got, want


got, want = run(np.arange(len(rois)))
bad = np.argwhere(got != want)
print("all ROIs: %d of %d elements differ" % (len(bad), want.size))
if len(bad):
    px = np.unique(bad[:, :3], axis=0)
    print("   pixels:", len(px), px[:12].tolist(), "channels:", np.unique(bad[:, 3])[:16].tolist())
    for b in bad[:4]:
        print("   at %s got %r want %r" % (b.tolist(), got[tuple(b)], want[tuple(b)]))
for i in range(len(rois)):
    got, want = run(np.array([i]))
    n = int((got != want).sum())
    if n:
        bad = np.argwhere(got != want)
        print("ROI %d %s alone: %d differ, pixels %s" % (i, rois[i].tolist(), n, np.unique(bad[:, :3], axis=0)[:8].tolist()))
        b = bad[0]
        print("   at %s got %r want %r" % (b.tolist(), got[tuple(b)], want[tuple(b)]))
