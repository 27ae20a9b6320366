#!/usr/bin/env python3
"""Which torch streams really run side by side on this box?  Chains of short spin kernels (torch.cuda._sleep) on subsets of a
pool of streams; serial = S x chain, concurrent = 1 x chain.  Diagnostics for bench.py's choice of streams."""
import itertools
import os
import sys
import time

import torch

dev = torch.device("cuda")
torch.cuda._sleep(1000)
torch.cuda.synchronize()
# calibrate _sleep cycles -> us
t0 = time.perf_counter(); torch.cuda._sleep(20_000_000); torch.cuda.synchronize(); dt = time.perf_counter() - t0
cyc_per_us = 20_000_000 / (dt * 1e6)
print("sleep cycles per us: %.1f" % cyc_per_us)
T, K = float(os.environ.get("T", "20")), int(os.environ.get("K", "40"))
cyc = int(T * cyc_per_us)
pool = [torch.cuda.Stream() for _ in range(int(os.environ.get("POOL", "8")))]
if os.environ.get("HIPRI"):
    pool += [torch.cuda.Stream(priority=-1) for _ in range(4)]
print("stream handles:", [hex(s.cuda_stream)[-6:] for s in pool])


def run(streams):
    for s in streams:
        with torch.cuda.stream(s):
            torch.cuda._sleep(cyc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        for s in streams:
            with torch.cuda.stream(s):
                torch.cuda._sleep(cyc)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6


base = run(pool[:1])
print("one stream: %.0f us for %d x %.0f us" % (base, K, T))
for S in range(2, len(pool) + 1):
    t = run(pool[:S])
    print("first %2d streams: %7.0f us -> concurrency %.2f" % (S, t, S * base / t))
print("pairs (concurrency):")
for a, b in itertools.combinations(range(len(pool)), 2):
    t = run([pool[a], pool[b]])
    sys.stdout.write("  (%d,%d) %.2f" % (a, b, 2 * base / t))
print()
best = None
for combo in itertools.combinations(range(len(pool)), 3):
    t = run([pool[i] for i in combo])
    c = 3 * base / t
    if best is None or c > best[0]:
        best = (c, combo)
print("best triple:", best)
best = None
for combo in itertools.combinations(range(len(pool)), 4):
    t = run([pool[i] for i in combo])
    c = 4 * base / t
    if best is None or c > best[0]:
        best = (c, combo)
print("best quad:", best)
