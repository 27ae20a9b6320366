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
    tr = torch.zeros((nb * 4 + 8,), dtype=torch.int64, device="cuda")
    ws = torch.empty((lib().mv3d_nms_workspace_bytes(n),), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        rc = lib().mv3d_nms_device_trace(C.c_void_p(d.data_ptr()), n, float(thresh), int(max_keep), C.c_void_p(keep.data_ptr()),
                                         C.c_void_p(cnt.data_ptr()), C.c_void_p(cnt[1:].data_ptr()), C.c_void_p(ws.data_ptr()),
                                         ws.numel(), None, C.c_void_p(tr.data_ptr()))
        check(rc, "trace")
        torch.cuda.synchronize()
    ph = tr.cpu().numpy()[nb * 4:]
    t = tr.cpu().numpy()[:nb * 4].reshape(nb, 4)
    used = t[:, 2] != 0
    t = t[used]
    iters = t[:, 3] >> 32
    kept = t[:, 3] & 0xffffffff
    step = np.diff(t[:, 0])
    print(f"[{tag}] n={n} blocks run={len(t)}/{nb} kept={int(cnt[0])} | cycles/step median={np.median(step):.0f} "
          f"mean={step.mean():.0f} max={step.max()} | wait median={np.median(t[:,1]-t[:,0]):.0f} mean={(t[:,1]-t[:,0]).mean():.0f} "
          f"| fixpoint cyc median={np.median(t[:,2]-t[:,1]):.0f} iters mean={iters.mean():.1f} max={iters.max()} | kept/block mean={kept.mean():.1f}")
    print("   total chain cycles:", int(t[-1, 2] - t[0, 0]))
    if ph[0]:
        print("   round-1 kernel phases (cycles): start->first tiles %d, ->chain done %d, ->end (emit) %d; total %d"
              % (ph[1] - ph[0], ph[2] - ph[1], ph[3] - ph[2], ph[3] - ph[0]))
    if "-v" in sys.argv:
        for i in range(len(t)):
            print("    b=%3d begin=%8d wait=%6d fix=%6d iters=%2d kept=%2d" % (i, t[i, 0] - t[0, 0], t[i, 1] - t[i, 0], t[i, 2] - t[i, 1], iters[i], kept[i]))


def bench_frame_dets(variant="peaky", pre=6000):
    """the bench frame's pre-NMS boxes in processing order, from the product itself: proposal_3d with
    an NMS threshold nothing reaches returns every sorted valid box"""
    prob, pred, info, calib = synth.rpn_head(1000, 76, 76, variant)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
    prm = ops.proposal_params(dict(RPN_PRE_NMS_TOP_N=pre, RPN_POST_NMS_TOP_N=pre, RPN_NMS_THRESH=2.0, RPN_MIN_SIZE=5))
    bv, img, b3, num, st = ops.proposal_3d(t(prob), t(pred), t(info), t(calib[None]), prm)
    k = int(num[0])
    d = np.zeros((k, 5), np.float32)
    d[:, :4] = bv[0, :k, 1:5].cpu().numpy()
    d[:, 4] = np.linspace(1, 0, k, dtype=np.float32)
    return d


if __name__ == "__main__":
    trace(bench_frame_dets("peaky"), 0.7, 300, "bench frame peaky TEST")
    trace(bench_frame_dets("rand"), 0.7, 300, "bench frame rand TEST")
    trace(bench_frame_dets("peaky", 12000), 0.7, 2000, "bench frame peaky TRAIN")
    if "--all" not in sys.argv:
        sys.exit(0)
    for n, var, cap in ((6000, "clustered", 300), (6000, "rand", 300), (6000, "clustered", 0), (12000, "clustered", 2000), (12000, "rand", 2000)):
        dets = synth.nms_dets(5, n, var)
        dets = dets[np.argsort(-dets[:, 4], kind="stable")]
        trace(dets, 0.7, cap, f"{var} cap={cap}")
