#!/usr/bin/env python3
"""Kernel-only timing of RoiPool forward/backward through the C-ABI with preallocated buffers."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mv3d_tf_amd import synth
from mv3d_tf_amd._lib import lib, check
P = lambda t: C.c_void_p(t.data_ptr())
def ev(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3
for (H, W, name) in ((76, 76, "BEV"), (46, 155, "RGB")):
    data = torch.as_tensor(synth.feature_map(7, H, W, 512, 1)).cuda()
    for R in (0, 30, 128, 300, 2000):
        rng = np.random.RandomState(R)
        x1 = rng.uniform(0, W * 8 - 40, max(R, 1)); y1 = rng.uniform(0, H * 8 - 20, max(R, 1))
        rois = np.stack([np.zeros(max(R, 1)), x1, y1, x1 + rng.uniform(10, 120, max(R, 1)), y1 + rng.uniform(10, 60, max(R, 1))], 1).astype(np.float32)
        rt = torch.as_tensor(rois).cuda()
        top = torch.empty((max(R, 1), 7, 7, 512), device="cuda"); am = torch.empty((max(R, 1), 7, 7, 512), dtype=torch.int32, device="cuda")
        out = torch.empty_like(data)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        f = lambda: check(lib().mv3d_roi_pool_forward(P(data), C.c_float(0.125), 1, R, H, W, 512, 7, 7, P(rt), P(top), P(am), st), "f")
        b = lambda: check(lib().mv3d_roi_pool_backward(P(top), C.c_float(0.125), 1, R, H, W, 512, 7, 7, P(rt), P(out), P(am), st), "b")
        f(); tf = ev(f) if R else 0.0; tb = ev(b)
        alg = R * 49 * 512 * 8 + H * W * 512 * 4
        print(f"{name} R={R:5d}: fwd {tf:7.1f} us  bwd {tb:7.1f} us  bwd alg {alg/1e6:7.1f} MB -> {alg/tb/1e3:7.0f} GB/s")
