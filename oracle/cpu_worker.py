#!/usr/bin/env python3
"""One host core's share of bench.py's `cpu_baseline`: the C oracle on whole frames, in its OWN process (own numpy global
RNG, no GIL shared with anybody), from a common start time to a common stop time.

TEST INFRASTRUCTURE ONLY (like everything under oracle/): executed by bench.py's cpu_baseline leg, never by the product.

    python oracle/cpu_worker.py <inputs.npz> <train|test> <start_unix_time> <stop_unix_time> <seed>

prints one line: "<frames done> <seconds between the common start and this worker's last frame>".
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle  # noqa: E402

VIEWS = ("bev", "rgb", "fv")


def make_frame(z, workload):
    prob, pred, info, calib = z["prob"], z["pred"], z["info"], z["calib"]
    gt_bv, gt_3d, gt_cnr = z["gt_bv"], z["gt_3d"], z["gt_cnr"]
    maps = {v: z["map_" + v] for v in VIEWS}
    key = "TRAIN" if workload == "train" else "TEST"
    cfg = {key: {k[len("cfg_"):]: (float(z[k]) if z[k].dtype.kind == "f" else int(z[k])) for k in z.files if k.startswith("cfg_")}}
    score = np.zeros((1, prob.shape[1], prob.shape[2], 8), np.float32)
    yml_train = dict(oracle.TRAIN, BG_THRESH_LO=0.0, BG_THRESH_HI=0.5, FG_THRESH=0.7)      # faster_rcnn_end2end.yml:10-12

    def frame():
        bv, img, b3 = oracle.proposal_layer_3d(prob, pred, info, calib, key, [8, ], cfg=cfg)
        if workload == "train":
            oracle.anchor_target_layer(score, gt_bv, gt_3d, info, [8, ])
            r_bv, r_img, _, _, r_3d = oracle.proposal_target_layer_3d(bv, b3, gt_bv, gt_3d, gt_cnr, calib, 2, train=yml_train)
            rois = {"bev": r_bv, "rgb": r_img, "fv": oracle.rois_3d_to_fv(r_3d)}
        else:
            rois = {"bev": bv, "rgb": img, "fv": oracle.rois_3d_to_fv(b3)}
        for v in VIEWS:
            top, am = oracle.roi_pool(maps[v], rois[v], 7, 7, 0.125)
            if workload == "train":
                oracle.roi_pool_grad(maps[v], rois[v], am, top, 7, 7, 0.125)
    return frame


def main():
    path, workload, start, stop, seed = sys.argv[1], sys.argv[2], float(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5])
    z = np.load(path)
    np.random.seed(seed)                          # a private stream per worker (its own process' global RNG)
    frame = make_frame(z, workload)
    frame()                                       # warm-up: library loaded, pages touched
    while time.time() < start:
        time.sleep(0.001)
    n, last = 0, start
    while time.time() < stop:
        frame()
        n += 1
        last = time.time()
    print(n, last - start)


if __name__ == "__main__":
    main()
