#!/usr/bin/env python3
"""Generate tests/golden/roipool_*.npz (SURVEY.md Appendix D, rows `roipool_{bev,rgb}_{R128,R300}` and
`roipool_edge`).

No runnable reference exists for RoiPool (TensorFlow op), so a vector is committed only where TWO INDEPENDENT
restatements of the reference agree bit for bit:
    oracle/mv3d_oracle.c          <- roi_pooling_op.cc:127-181, :373-443        (CPU op)
    tests/roipool_restatement.py  <- roi_pooling_op_gpu.cu.cc:27-84, :121-189   (CUDA kernels)
The script aborts on the first disagreement.  Inputs are regenerated from seeds (mv3d_tf_amd.synth is
bit-reproducible); ROIs are the first R rows of the oracle's proposal_layer_3d blobs for seeded frames, i.e. real
BEV / image boxes with the batch-index column set per frame.

    python tests/golden/make_roipool_golden.py          # ~2 min on 1 core
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]

import oracle  # noqa: E402
import roipool_restatement as rs  # noqa: E402
from mv3d_tf_amd import synth  # noqa: E402

TRAIN = dict(RPN_PRE_NMS_TOP_N=12000, RPN_POST_NMS_TOP_N=2000, RPN_NMS_THRESH=0.7, RPN_MIN_SIZE=5)
VIEWS = {"bev": (76, 76, 0), "rgb": (46, 155, 1)}        # H, W, which blob (0 = blob_bv, 1 = blob_img)


def frame_rois(view, R, B, seed0):
    """first R/B rows of each frame's proposals, frame index in column 0"""
    per = R // B
    rows = []
    for b in range(B):
        prob, pred, info, calib = synth.rpn_head(seed0 + b, 76, 76, "peaky")
        blobs = oracle.proposal_layer_3d(prob, pred, info, calib, "TRAIN", [8, ], cfg={"TRAIN": TRAIN})
        r = blobs[VIEWS[view][2]][:per].copy()
        assert r.shape[0] == per, "frame has fewer proposals than requested"
        r[:, 0] = b
        rows.append(r)
    return np.concatenate(rows).astype(np.float32)


def edge_rois(H, W, B):
    """SURVEY Appendix D `roipool_edge`: outside the map, 1x1, negative coordinates, malformed, larger than the map,
    .5 rounding (half away from zero, both signs), batch index > 0, a run of ROIs stacked on one pixel."""
    S = 8.0
    r = [[0, 0, 0, 0, 0], [0, W * S + 50, H * S + 50, W * S + 90, H * S + 90], [0, 30, 30, 10, 10],
         [0, -100, -100, W * S + 100, H * S + 100], [0, 11.5, 3.5, 51.5, 43.5], [0, -12, -20, 36, 28],
         [0, -4, -4, -4, -4], [0, -3.9, -4.1, 4.1, 3.9], [B - 1, 0, 0, W * S - 1, H * S - 1], [B - 1, 20, 4, 21, 60],
         [B - 1, 4, 20, 60, 21], [0, (W - 1) * S, (H - 1) * S, (W - 1) * S + 7, (H - 1) * S + 7],
         [0, 8, 8, 63, 63], [0, 8, 8, 64, 64], [0, 8, 8, 71, 71], [0, 8, 8, 72, 72]]
    r += [[0, 16, 16, 16 + 8 * k, 16 + 8 * ((k * 3) % 7)] for k in range(12)]            # all contain pixel (2, 2)
    return np.array(r, np.float32)


def tie_map(data):
    data[0, 0, 0, :] = 1.5
    data[0, 0, 1 % data.shape[2], :] = 1.5            # equal maxima: the first one (scan order h, w) must win
    data[0, 2, 2, 0] = np.nan                          # NaN never wins (strict >)
    return data


def run_case(name, data, rois, grad_seed, store_arrays, meta):
    B, H, W, C = data.shape
    o_top, o_am = oracle.roi_pool(data, rois, 7, 7, 0.125)
    r_top, r_am = rs.forward(data, rois, 7, 7, 0.125)
    assert np.array_equal(o_top, r_top, equal_nan=True) and np.array_equal(o_am, r_am), name + ": forward restatements disagree"
    grad = np.random.RandomState(grad_seed).uniform(-1, 1, o_top.shape).astype(np.float32)
    o_bd = oracle.roi_pool_grad(data, rois, o_am, grad, 7, 7, 0.125)
    r_bd = rs.backward(grad, r_am, rois, B, H, W, 7, 7, 0.125)
    assert np.array_equal(o_bd, r_bd), name + ": backward restatements disagree"
    out = dict(meta, rois=rois, grad_seed=grad_seed, shape=np.array(data.shape), data_sha=synth.sha256(data),
               numpy_version=np.__version__, top_sha=synth.sha256(o_top), argmax_sha=synth.sha256(o_am),
               bottom_diff_sha=synth.sha256(o_bd),
               nonempty_bins=int((o_am[..., 0] >= 0).sum()), grad_elems=int((o_bd != 0).sum()))
    if store_arrays:
        out.update(top=o_top, argmax=o_am, bottom_diff=o_bd)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("%-22s data %s rois %d  non-empty bins %d  nonzero grads %d" % (name, data.shape, len(rois), out["nonempty_bins"],
                                                                       out["grad_elems"]))


def main():
    oracle.build()
    for view, (H, W, _) in VIEWS.items():
        for R in (128, 300):
            B = 2
            rois = frame_rois(view, R, B, 500)
            seed = 40 + (0 if view == "bev" else 1)
            run_case("roipool_%s_R%d" % (view, R), synth.feature_map(seed, H, W, 8, B), rois, 900 + R, True,
                     dict(map_seed=seed, channels=8, batch=B, ties=0))
    # one full-width (C = 512) case per view: hashes only (the arrays are tens of MB)
    for view, (H, W, _) in VIEWS.items():
        rois = frame_rois(view, 128, 1, 600)
        seed = 50 + (0 if view == "bev" else 1)
        run_case("roipool_%s_C512" % view, synth.feature_map(seed, H, W, 512, 1), rois, 777, False,
                 dict(map_seed=seed, channels=512, batch=1, ties=0))
    H, W, B = 12, 9, 3
    run_case("roipool_edge", tie_map(synth.feature_map(60, H, W, 8, B)), edge_rois(H, W, B), 555, True,
             dict(map_seed=60, channels=8, batch=B, ties=1))


if __name__ == "__main__":
    main()
