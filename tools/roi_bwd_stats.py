#!/usr/bin/env python3
"""Host-side statistics of the RoiPoolGrad candidate lists on the bench's training batches (numpy, from the ROIs alone):
per view -- pixels with candidates, (pixel, record) pairs, the distribution of list lengths, and what merging G adjacent
pixels of a row into one item would do (items, union list entries, longest union)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mv3d_tf_amd import build, hot_path, synth
from mv3d_tf_amd.fast_rcnn.config import apply_end2end_yml

build.build()
apply_end2end_yml()
np.random.seed(3)
dev = torch.device("cuda")
frames = [synth.rpn_head(100000 + b, 76, 76, "peaky", return_gt=True) for b in range(2)]
bt = hot_path.TrainPathBatch(frames, hot_path.synth_maps(2, 0, dev)).setup()
f32 = np.float32


def rnd(x):                       # C round(): half away from zero, on the f32 product
    return np.where(x >= 0, np.floor(x + f32(0.5)), np.ceil(x - f32(0.5))).astype(np.int64)


for v in hot_path.VIEWS:
    rois = bt.rois[v].cpu().numpy()
    B, H, W, C = bt.maps[v].shape
    lists = {}                    # (n, h, w) -> count ; per group later
    pairs_by_pix = {}
    per_roi_cols = []
    for r, roi in enumerate(rois):
        n = int(roi[0])
        s = (roi[1:] * f32(0.125)).astype(f32)
        rsw, rsh, rew, reh = [int(x) for x in rnd(s)]
        rh, rw = max(reh - rsh + 1, 1), max(rew - rsw + 1, 1)
        bh, bw = f32(rh) / f32(7), f32(rw) / f32(7)
        for h in range(max(rsh, 0), min(reh, H - 1) + 1):
            phs = int(np.floor(f32(h - rsh) / bh)); phe = int(np.ceil(f32(h - rsh + 1) / bh))
            phs, phe = min(max(phs, 0), 7), min(max(phe, 0), 7)
            if phe <= phs:
                continue
            for w in range(max(rsw, 0), min(rew, W - 1) + 1):
                x0 = int(np.floor(f32(w - rsw) / bw)); x1 = int(np.ceil(f32(w - rsw + 1) / bw))
                x0, x1 = min(max(x0, 0), 7), min(max(x1, 0), 7)
                if x1 <= x0:
                    continue
                pairs_by_pix.setdefault((n, h, w), []).append((r, phs, phe, x0, x1))
    cnt = np.array([sum((e[2] - e[1]) * (e[4] - e[3]) for e in L) for L in pairs_by_pix.values()])
    print("%-4s R=%d map %dx%dx%d: pixels with candidates %d of %d, (pixel, record) pairs %d = %.2f per record; list length "
          "mean %.1f median %d p90 %d max %d; round trips (1 + ceil(len/32)) total %d"
          % (v, len(rois), B * H, W, C, len(cnt), B * H * W, cnt.sum(), cnt.sum() / (len(rois) * 49.0), cnt.mean(), np.median(cnt),
             np.percentile(cnt, 90), cnt.max(), int((1 + np.ceil(cnt / 32.0)).sum())))
    for G in (2, 4, 8):
        groups = {}
        for (n, h, w), L in pairs_by_pix.items():
            groups.setdefault((n, h, w // G), {}).setdefault("px", []).append(L)
        un = []
        for key, g in groups.items():
            ent = {}
            for L in g["px"]:
                for (r, phs, phe, x0, x1) in L:
                    a = ent.setdefault(r, [phs, phe, x0, x1])
                    a[2] = min(a[2], x0); a[3] = max(a[3], x1)
            un.append(sum((e[1] - e[0]) * (e[3] - e[2]) for e in ent.values()))
        un = np.array(un)
        print("     groups of %d pixels: items %d, union entries %d (%.2f of the pairs), longest %d, round trips %d"
              % (G, len(un), un.sum(), un.sum() / cnt.sum(), un.max(), int((1 + np.ceil(un / 32.0)).sum())))
