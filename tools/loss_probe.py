#!/usr/bin/env python3
"""Kernel time of the fused training losses (runs on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mv3d_tf_amd import ops
def ev(fn, it=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3
N = 23104
z = torch.randn(N, 2, device="cuda"); lab = torch.full((N,), -1.0, device="cuda"); lab[:128] = 0; lab[:32] = 1
p = torch.randn(N, 6, device="cuda"); t = torch.randn(N, 6, device="cuda")
print("mv3d_rpn_loss  (23104 anchors, values + gradients, 2 launches): %.1f us" % ev(lambda: ops.rpn_loss(z, lab, p, t)))
cs = torch.randn(128, 2, device="cuda"); lb = torch.randint(0, 2, (128,), device="cuda", dtype=torch.int32)
bp = torch.randn(128, 48, device="cuda"); bt = torch.randn(128, 48, device="cuda")
print("mv3d_rcnn_loss (128 rois,      values + gradients, 2 launches): %.1f us" % ev(lambda: ops.rcnn_loss(cs, lb, bp, bt)))
