#!/usr/bin/env python3
"""The input layer's kernel (csrc/conv_input.hip) against torch's f32 convolution on given view sizes, grouped and single: max error per
view, and that the output's frame stays zero."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from mv3d_tf_amd import ops

torch.manual_seed(0)
T = torch.float16
SETS = [[(2, 608, 608, 9), (2, 96, 320, 3)], [(2, 608, 608, 9), (2, 96, 320, 3), (2, 64, 512, 3)], [(1, 37, 45, 9)], [(3, 16, 32, 3), (1, 17, 33, 9)]]
for views in SETS:
    xs, ws, bs, refs = [], [], [], []
    for B, H, W, cin in views:
        x = torch.randn((B, H, W, cin), device="cuda")
        w = torch.randn((64, cin, 3, 3), device="cuda") * 0.2
        b = torch.randn(64, device="cuda")
        xs.append(ops.frame_nhwc_f16(x, ops.framed_buffer(B, H, W, 16, "cuda", T)))
        ws.append(ops.pack_conv3x3_weights_input_layer(w, dtype=T))
        bs.append(b)
        refs.append(F.relu(F.conv2d(x.to(T).float().permute(0, 3, 1, 2), w.to(T).float(), b, padding=1)).permute(0, 2, 3, 1))
    outs = [ops.framed_buffer(B, H, W, 64, "cuda", T) for B, H, W, _ in views]
    ops.conv3x3_views([(x, w, b, None, o) for x, w, b, o in zip(xs, ws, bs, outs)])
    torch.cuda.synchronize()
    for (B, H, W, cin), o, r in zip(views, outs, refs):
        inner = o[:, 1:-1, 1:-1].float()
        err = (inner - r).abs()
        frame = o.clone(); frame[:, 1:-1, 1:-1] = 0
        bad = (err > 0.02 * (1 + r.abs())).nonzero()
        print("grouped %s: max err %.4g (ref max %.3g), frame max %.3g, bad %d %s" % ((B, H, W, cin), float(err.max()), float(r.abs().max()),
              float(frame.abs().max()), bad.shape[0], bad[:5].tolist()))
    if views == SETS[0]:
        import numpy as np
        off = 0
        for (B, H, W, cin), o, r in zip(views, outs, refs):
            inner = o[:, 1:-1, 1:-1].float()
            bad = ((inner - r).abs() > 0.02 * (1 + r.abs())) | torch.isnan(inner)
            idx = bad.nonzero().cpu().numpy()
            xblocks, strips = (W + 31) // 32, (H + 15) // 16
            u = off + (idx[:, 0] * strips + idx[:, 1] // 16) * xblocks + idx[:, 2] // 32
            uu, cnt = np.unique(u, return_counts=True)
            print("view", (B, H, W), "bad units", len(uu), "of", B * strips * xblocks, "first", uu[:12].tolist(), "last", uu[-6:].tolist())
            print("   wave-in-WG histogram", np.bincount(uu % 4, minlength=4).tolist(), " WG index min/max", int(uu.min()) // 4, int(uu.max()) // 4)
            print("   rows-in-strip histogram", np.bincount(idx[:, 1] % 16, minlength=16).tolist())
            print("   pixel-in-block histogram", np.bincount(idx[:, 2] % 32, minlength=32).tolist())
            print("   channel histogram", np.bincount(idx[:, 3], minlength=64).tolist())
            off += B * strips * xblocks
        # what ARE the bad values?  the same position of another row (a register read too late / too early), or junk
        (B, H, W, cin), o, r = views[0], outs[0], refs[0]
        inner = o[:, 1:-1, 1:-1].float()
        bad = ((inner - r).abs() > 0.02 * (1 + r.abs())) | torch.isnan(inner)
        idx = bad.nonzero()
        idx = idx[(idx[:, 1] > 2) & (idx[:, 1] < H - 3)]
        g = inner[idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]]
        print("bad values: %d, NaN %d, zero %d" % (g.numel(), int(torch.isnan(g).sum()), int((g == 0).sum())))
        rq = r.to(torch.float16).float()
        for dy in (-2, -1, 1, 2):
            m = (g == rq[idx[:, 0], idx[:, 1] + dy, idx[:, 2], idx[:, 3]])
            print("   equal to the reference of row %+d, same pixel / channel: %d" % (dy, int(m.sum())))
        for dc in (-32, 2, -2, 4, 8):
            cc = idx[:, 3] + dc
            ok = (cc >= 0) & (cc < 64)
            m = (g[ok] == rq[idx[ok, 0], idx[ok, 1], idx[ok, 2], cc[ok]])
            print("   equal to the reference of channel %+d: %d" % (dc, int(m.sum())))
        print("   sample:", [(float(a), float(b_)) for a, b_ in zip(g[:8], rq[idx[:8, 0], idx[:8, 1], idx[:8, 2], idx[:8, 3]])])
