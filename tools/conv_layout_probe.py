#!/usr/bin/env python3
"""fwd + bwd time of every distinct 3x3 convolution of the two VGG16 trunks (and the RPN 1x1 heads) through torch / MIOpen, in
the layouts networks/mv3d.py could use: which shapes fall to MIOpen's naive kernels."""
import time
import torch
import torch.nn.functional as F

def run(x, w, pad, iters=3):
    for _ in range(2):
        y = F.conv2d(x, w, padding=pad); y.sum().backward()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(iters):
        y = F.conv2d(x, w, padding=pad); y.sum().backward()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e3

shapes = []
for (cin0, H, W) in ((9, 608, 608), (3, 375, 1242)):
    c, h, w = cin0, H, W
    for cout, pool in ((64, 0), (64, 1), (128, 0), (128, 1), (256, 0), (256, 0), (256, 1), (512, 0), (512, 0), (512, 0), (512, 0), (512, 0), (512, 0)):
        shapes.append((c, cout, h, w, 3))
        c = cout
        if pool: h, w = h // 2, w // 2
shapes += [(512, 512, 76, 76, 3), (512, 8, 76, 76, 1), (512, 24, 76, 76, 1)]
seen = set()
tot = {"nchw": 0.0, "cl": 0.0, "mixed": 0.0}
for (cin, cout, H, W, k) in shapes:
    key = (cin, cout, H, W, k)
    res = {}
    for name in ("nchw", "cl", "mixed"):
        x = torch.randn(1, cin, H, W, device="cuda")
        wt = torch.randn(cout, cin, k, k, device="cuda")
        if name != "nchw": x = x.contiguous(memory_format=torch.channels_last)
        if name == "cl": wt = wt.contiguous(memory_format=torch.channels_last)
        res[name] = run(x.requires_grad_(True), wt.requires_grad_(True), k // 2)
        tot[name] += res[name]
    if key not in seen:
        print(key, " ".join("%s %.2f ms" % kv for kv in res.items()))
    seen.add(key)
print("sum over the layers (fwd+bwd):", " ".join("%s %.1f ms" % kv for kv in tot.items()))
