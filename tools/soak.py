#!/usr/bin/env python3
"""Soak test (runs on the GPU box): the bench step repeated many times on several streams at once; every result
must be bit-identical to the first one (catches rare races in the LDS-flag protocols of the NMS kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=3000)
ap.add_argument("--streams", type=int, default=3)
ap.add_argument("--cfg", default="TEST")
ap.add_argument("--variant", default="peaky")
ap.add_argument("--batch", type=int, default=1)
a = ap.parse_args()
args = argparse.Namespace(batch=a.batch, cfg=a.cfg, variant=a.variant, only="", graph=False, no_graph=False)
from mv3d_tf_amd import build
build.build()
frs = [bench.Frames(args, 0, torch.cuda.Stream()) for _ in range(a.streams)]
for fr in frs:
    fr.step()
torch.cuda.synchronize()
ref = [t.clone() for t in frs[0].out] + [t.clone() for t in frs[0].tops]
bad = 0
for it in range(a.iters):
    for fr in frs:
        for t in fr.out[:3]:
            t.fill_(-7.0)                      # a stale result cannot pass
    for fr in frs:
        fr.step()
    torch.cuda.synchronize()
    if it % 50 == 0 or it == a.iters - 1:     # full comparison every 50 iterations, ROI blobs every time
        for fr in frs:
            got = list(fr.out) + list(fr.tops)
            if not all(torch.equal(x, y) for x, y in zip(got, ref)):
                bad += 1
    else:
        for fr in frs:
            if not all(torch.equal(x, y) for x, y in zip(fr.out, ref[:5])):
                bad += 1
print("soak: %d iterations x %d streams (%s cfg, batch %d): %d mismatches" % (a.iters, a.streams, a.cfg, a.batch, bad))
sys.exit(1 if bad else 0)
