#!/usr/bin/env python3
"""Small RoiPoolGrad cases against the oracle with a mismatch report (which channels / pixels differ)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mv3d_tf_amd import build, ops, synth
from oracle import oracle

build.build()
for (B, H, W, C, R, seed) in ((1, 9, 11, 64, 6, 1), (2, 13, 17, 64, 40, 2), (2, 20, 31, 128, 90, 3), (2, 24, 24, 512, 128, 4), (1, 9, 11, 64, 300, 5), (2, 12, 12, 64, 700, 6), (1, 40, 40, 256, 256, 7)):
    rng = np.random.RandomState(seed)
    data = synth.feature_map(seed, H, W, C, B)
    x1 = rng.uniform(-8, W * 8 - 8, R); y1 = rng.uniform(-8, H * 8 - 8, R)
    rois = np.stack([rng.randint(0, B, R).astype(np.float64), x1, y1, x1 + rng.uniform(0, 60, R), y1 + rng.uniform(0, 60, R)], 1).astype(np.float32)
    top, am = oracle.roi_pool(data, rois, 7, 7, 0.125)
    grad = rng.uniform(-1, 1, top.shape).astype(np.float32)
    want = oracle.roi_pool_grad(data, rois, am, grad, 7, 7, 0.125)
    got = ops.roi_pool_backward_views([(torch.as_tensor(grad).cuda(), torch.as_tensor(rois).cuda(), torch.as_tensor(am).cuda(), tuple(data.shape), 0.125)], 7, 7)[0].cpu().numpy()   # (the workspace path)
    bad = np.argwhere(got != want)
    print("case B%d H%d W%d C%d R%d: %d of %d elements differ" % (B, H, W, C, R, len(bad), want.size))
    if len(bad):
        ch = bad[:, 3]
        print("   channels even/odd: %d / %d ; lower half (c%%64<32) %d ; first: %s" % ((ch % 2 == 0).sum(), (ch % 2 == 1).sum(), ((ch % 64) < 32).sum(), bad[:6].tolist()))
        for b in bad[:6]:
            print("   got %r want %r" % (got[tuple(b)], want[tuple(b)]))
        nz = (want != 0)
        print("   want nonzero %d, got nonzero %d, got==0 where want!=0: %d" % (nz.sum(), (got != 0).sum(), ((got == 0) & nz).sum()))

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_roipool_pin as trp
for name in ("roipool_bev_C512", "roipool_rgb_C512"):
    g, data, rois, grad = trp.load_case(name)
    top, am = oracle.roi_pool(data, rois, 7, 7, 0.125)
    want = oracle.roi_pool_grad(data, rois, am, grad, 7, 7, 0.125)
    got = ops.roi_pool_backward_views([(torch.as_tensor(grad).cuda(), torch.as_tensor(rois).cuda(), torch.as_tensor(am).cuda(), tuple(data.shape), 0.125)], 7, 7)[0].cpu().numpy()   # (the workspace path)
    bad = np.argwhere(got != want)
    print(name, data.shape, rois.shape, "differ:", len(bad), "nan in want", np.isnan(want).sum(), "nan in got", np.isnan(got).sum())
    if len(bad):
        px = np.unique(bad[:, :3], axis=0)
        print("   pixels with differences:", len(px), px[:10].tolist())
        for b in bad[:6]:
            print("   at %s got %r want %r" % (b.tolist(), got[tuple(b)], want[tuple(b)]))
        b = bad[0]
        n, h, w, c = b
        cand = []
        for r, roi in enumerate(rois):
            for ph in range(7):
                for pw in range(7):
                    if am[r, ph, pw, c] == (h * data.shape[2] + w) * data.shape[3] + c and int(roi[0]) == n:
                        cand.append((r, ph, pw, float(grad[r, ph, pw, c])))
        print("   contributions to the first bad element:", cand)
