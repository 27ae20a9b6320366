#!/usr/bin/env python3
"""Timing of the training-side kernels (runs on the GPU box): RoiPoolGrad on both views, the
anchor_target_layer and proposal_target_layer_3d callables (incl. their host syncs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mv3d_tf_amd import ops, synth
from mv3d_tf_amd.fast_rcnn.config import cfg
from mv3d_tf_amd.rpn_msr.anchor_target_layer_tf import anchor_target_layer
from mv3d_tf_amd.rpn_msr.proposal_layer_tf import proposal_layer_3d
from mv3d_tf_amd.rpn_msr.proposal_target_layer_tf import proposal_target_layer_3d


def ev(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


prob, pred, info, calib = synth.rpn_head(1000, 76, 76, "peaky")
r = np.random.RandomState(1)
gtbv, gt3d, gtc = synth.gt_cars(r, 8)
bv, img, b3 = proposal_layer_3d(prob, pred, info, calib, "TRAIN", [8, ])
np.random.seed(3)
out = proposal_target_layer_3d(bv, b3, gtbv, gt3d, gtc, calib, 2)
print("proposals", bv.shape, "sampled rois", out[0].shape)
for name, (H, W), rois in (("BEV", (76, 76), out[0]), ("RGB", (46, 155), out[1])):
    data = torch.as_tensor(synth.feature_map(7, H, W, 512, 1)).cuda()
    rt = torch.as_tensor(rois).cuda()
    top, am = ops.roi_pool_forward(data, rt, 7, 7, 0.125)
    g = torch.rand_like(top)
    us_f = ev(lambda: ops.roi_pool_forward(data, rt, 7, 7, 0.125))
    us_b = ev(lambda: ops.roi_pool_backward(g, rt, am, data.shape, 7, 7, 0.125))
    R = rois.shape[0]
    alg_b = R * 49 * 512 * 8 + H * W * 512 * 4
    print(f"RoiPool {name} R={R}: fwd {us_f:.1f} us, bwd {us_b:.1f} us ({alg_b/us_b/1e3:.0f} GB/s of {alg_b/1e6:.1f} MB algorithmic)")
score = np.zeros((1, 76, 76, 8), np.float32)
t0 = time.perf_counter()
for _ in range(20):
    anchor_target_layer(score, gtbv, gt3d, info, [8, ])
torch.cuda.synchronize()
print(f"anchor_target_layer (numpy in/out, 1-2 host syncs): {(time.perf_counter()-t0)/20*1e6:.0f} us/call")
t0 = time.perf_counter()
for _ in range(20):
    proposal_target_layer_3d(bv, b3, gtbv, gt3d, gtc, calib, 2)
torch.cuda.synchronize()
print(f"proposal_target_layer_3d (numpy in/out, 1 host sync): {(time.perf_counter()-t0)/20*1e6:.0f} us/call")
t0 = time.perf_counter()
for _ in range(20):
    proposal_layer_3d(prob, pred, info, calib, "TRAIN", [8, ])
torch.cuda.synchronize()
print(f"proposal_layer_3d TRAIN (numpy in/out): {(time.perf_counter()-t0)/20*1e6:.0f} us/call")
