#!/usr/bin/env python3
"""CPU-side statistics for the tiled RoiPoolGrad (no GPU): the bench's training-batch ROIs are produced with the oracle,
then for a tile shape (th x tw pixels) per view: records per tile (a record = one (roi, ph, pw) whose candidate rectangle
meets the tile), visits per record, the busiest tiles.  Usage: roi_tile_stats.py [nbatches]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mv3d_tf_amd import synth
from oracle import oracle

f32 = np.float32
TRAIN_CFG = dict(oracle.TRAIN)
TRAIN_CFG.update(BG_THRESH_LO=0.0, BG_THRESH_HI=0.5, FG_THRESH=0.7)
SHAPES = {"bev": (76, 76), "rgb": (46, 155), "fv": (8, 64)}


def batch_rois(k, B=2):
    out = {v: [] for v in SHAPES}
    for b in range(B):
        prob, pred, info, calib, (gt_bv, gt_3d, gt_cnr) = synth.rpn_head(100000 + 2 * k + b, 76, 76, "peaky", return_gt=True)
        bv, img, b3 = oracle.proposal_layer_3d(prob, pred, info, calib, "TRAIN", [8, ])
        r_bv, r_img, r_lab, r_tg, r_3d = oracle.proposal_target_layer_3d(bv, b3, gt_bv, gt_3d, gt_cnr, calib, 2, train=TRAIN_CFG)
        r_bv[:, 0] = b; r_img[:, 0] = b; r_3d[:, 0] = b
        out["bev"].append(r_bv); out["rgb"].append(r_img); out["fv"].append(oracle.rois_3d_to_fv(r_3d))
    return {v: np.concatenate(x) for v, x in out.items()}


def rnd(x):
    return np.where(x >= 0, np.floor(x + f32(0.5)), np.ceil(x - f32(0.5))).astype(np.int64)


def ranges(lo, hi, n, P=7):
    """candidate range of pixel coordinates per pooled index: list of (first, last) or None -- the reference's per-pixel
    test (roi_pooling_op.cc:423-426) transposed"""
    size = max(hi - lo + 1, 1)
    b = f32(size) / f32(P)
    res = [None] * P
    for x in range(max(lo, 0), min(hi, n - 1) + 1):
        s = int(np.floor(f32(x - lo) / b)); e = int(np.ceil(f32(x - lo + 1) / b))
        s, e = min(max(s, 0), P), min(max(e, 0), P)
        for p in range(s, e):
            res[p] = (x, x) if res[p] is None else (res[p][0], x)
    return res


def stats(rois, H, W, th, tw):
    tiles = {}
    nrec = 0
    for r, roi in enumerate(rois):
        n = int(roi[0])
        rsw, rsh, rew, reh = [int(x) for x in rnd((roi[1:] * f32(0.125)).astype(f32))]
        hr, wr = ranges(rsh, reh, H), ranges(rsw, rew, W)
        for ph in range(7):
            for pw in range(7):
                if hr[ph] is None or wr[pw] is None:
                    continue
                nrec += 1
                for ty in range(hr[ph][0] // th, hr[ph][1] // th + 1):
                    for tx in range(wr[pw][0] // tw, wr[pw][1] // tw + 1):
                        tiles[(n, ty, tx)] = tiles.get((n, ty, tx), 0) + 1
    cnt = np.array(sorted(tiles.values())) if tiles else np.zeros(1)
    ntiles = 2 * ((H + th - 1) // th) * ((W + tw - 1) // tw)
    return nrec, cnt, ntiles


if __name__ == "__main__":
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    np.random.seed(3)
    for k in range(nb):
        rois = batch_rois(k)
        for v, (H, W) in SHAPES.items():
            for th, tw in ((1, 1), (1, 2), (2, 2), (2, 4), (4, 4), (4, 8), (8, 8), (8, 16)):
                if th > H:
                    continue
                nrec, cnt, ntiles = stats(rois[v], H, W, th, tw)
                print("batch %d %-3s tile %dx%-2d: tiles %5d (with records %5d), live records %5d, visits %6d = %.2f / record; "
                      "per tile median %3d p90 %3d max %4d top5 %s"
                      % (k, v, th, tw, ntiles, len(cnt), nrec, cnt.sum(), cnt.sum() / max(nrec, 1), np.median(cnt),
                         np.percentile(cnt, 90), cnt.max(), cnt[-5:][::-1].tolist()))
