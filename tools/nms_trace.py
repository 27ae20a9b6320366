#!/usr/bin/env python3
"""Per-block trace of the NMS greedy chain (runs on the GPU box).  Uses the bench's frame
generator to build the sorted dets of one frame via the oracle-free path: proposal ->
order is internal, so here we simply trace standalone NMS inputs (synth.nms_dets) and the
peaky proposal frame's top-K boxes reconstructed with the product's own decode."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mv3d_tf_amd import ops, synth
from mv3d_tf_amd._lib import lib, check


def trace(dets_sorted, thresh, max_keep, tag):
    d = torch.as_tensor(dets_sorted).cuda()
    n = d.shape[0]
    nb = (n + 63) // 64
    keep = torch.empty((n,), dtype=torch.int32, device="cuda")
    cnt = torch.zeros((2,), dtype=torch.int32, device="cuda")
    tr = torch.zeros((nb * 4,), dtype=torch.int64, device="cuda")
    ws = torch.empty((lib().mv3d_nms_workspace_bytes(n),), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        rc = lib().mv3d_nms_device_trace(C.c_void_p(d.data_ptr()), n, float(thresh), int(max_keep), C.c_void_p(keep.data_ptr()),
                                         C.c_void_p(cnt.data_ptr()), C.c_void_p(cnt[1:].data_ptr()), C.c_void_p(ws.data_ptr()),
                                         ws.numel(), None, C.c_void_p(tr.data_ptr()))
        check(rc, "trace")
        torch.cuda.synchronize()
    t = tr.cpu().numpy().reshape(nb, 4)
    used = t[:, 2] != 0
    t = t[used]
    iters = t[:, 3] >> 32
    kept = t[:, 3] & 0xffffffff
    step = np.diff(t[:, 0])
    print(f"[{tag}] n={n} blocks run={len(t)}/{nb} kept={int(cnt[0])} | cycles/step median={np.median(step):.0f} "
          f"mean={step.mean():.0f} max={step.max()} | wait median={np.median(t[:,1]-t[:,0]):.0f} mean={(t[:,1]-t[:,0]).mean():.0f} "
          f"| fixpoint cyc median={np.median(t[:,2]-t[:,1]):.0f} iters mean={iters.mean():.1f} max={iters.max()} | kept/block mean={kept.mean():.1f}")
    print("   total chain cycles:", int(t[-1, 2] - t[0, 0]))


if __name__ == "__main__":
    for n, var, cap in ((6000, "clustered", 300), (6000, "rand", 300), (6000, "clustered", 0), (12000, "clustered", 2000), (12000, "rand", 2000)):
        dets = synth.nms_dets(5, n, var)
        dets = dets[np.argsort(-dets[:, 4], kind="stable")]
        trace(dets, 0.7, cap, f"{var} cap={cap}")
