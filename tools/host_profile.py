#!/usr/bin/env python3
"""cProfile of the numpy-in / numpy-out training callables (runs on the GPU box): where the host time goes."""
import cProfile, pstats, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mv3d_tf_amd import synth
from mv3d_tf_amd.rpn_msr.anchor_target_layer_tf import anchor_target_layer
from mv3d_tf_amd.rpn_msr.proposal_layer_tf import proposal_layer_3d
from mv3d_tf_amd.rpn_msr.proposal_target_layer_tf import proposal_target_layer_3d

prob, pred, info, calib = synth.rpn_head(1000, 76, 76, "peaky")
gtbv, gt3d, gtc = synth.gt_cars(np.random.RandomState(1), 8)
bv, img, b3 = proposal_layer_3d(prob, pred, info, calib, "TRAIN", [8, ])
score = np.zeros((1, 76, 76, 8), np.float32)
for name, fn in (("anchor_target_layer", lambda: anchor_target_layer(score, gtbv, gt3d, info, [8, ])),
                 ("proposal_target_layer_3d", lambda: proposal_target_layer_3d(bv, b3, gtbv, gt3d, gtc, calib, 2)),
                 ("proposal_layer_3d", lambda: proposal_layer_3d(prob, pred, info, calib, "TRAIN", [8, ]))):
    for _ in range(5):
        fn()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        fn()
    pr.disable()
    print("=====", name)
    pstats.Stats(pr).sort_stats("tottime").print_stats(12)
