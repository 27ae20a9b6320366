"""cosine / max-error of the training trunk's gradients (mv3d_tf_amd.trunk_train) against torch fp32 autograd, next to what torch's
own bf16 autocast gets on the same trunk:  python tools/trunk_grad_debug.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mv3d_tf_amd import build, trunk_train  # noqa: E402

build.build()
F = torch.nn.functional
layers = [("a", 64, False), ("b", 64, True), ("c", 128, False), ("d", 128, True), ("e", 256, False)]
g = torch.Generator(device="cuda").manual_seed(5)
params, cin = {}, 9
for name, cout, _ in layers:
    params[name] = [(torch.randn((cout, cin, 3, 3), device="cuda", generator=g) * (2.0 / (9 * cin)) ** 0.5).requires_grad_(True),
                    (torch.randn((cout,), device="cuda", generator=g) * 0.1).requires_grad_(True)]
    cin = cout
x = torch.randn((2, 42, 54, 9), device="cuda", generator=g)
R = torch.randn((2, 10, 13, 256), device="cuda", generator=g)


def torch_trunk(amp):
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        t = x.permute(0, 3, 1, 2)
        for name, _, pool in layers:
            w, b = params[name]
            t = torch.relu(F.conv2d(t, w, b, padding=1))
            if pool:
                t = F.max_pool2d(t, 2, 2)
        return t.permute(0, 2, 3, 1).float()


def grads(fn):
    for v in params.values():
        v[0].grad = v[1].grad = None
    out = fn()
    (out * R).sum().backward()
    return out.detach(), {k: (v[0].grad.clone().float(), v[1].grad.clone().float()) for k, v in params.items()}


o32, g32 = grads(lambda: torch_trunk(False))
oam, gam = grads(lambda: torch_trunk(True))
fns = {"mfma+torch-wgrad": trunk_train._wgrad_torch}
if hasattr(trunk_train, "wgrad_mfma"):
    fns["mfma"] = trunk_train.wgrad_mfma
res = {"torch bf16 autocast": (oam, gam)}
for k, fn in fns.items():
    res[k] = grads(lambda: trunk_train.trunk(layers, x, params, "", wgrad=fn))
cos = lambda a, b: float(F.cosine_similarity(a.flatten(), b.flatten(), dim=0))
for k, (o, gr) in res.items():
    print(k, "out max err / max %.4f" % (float((o - o32).abs().max()) / float(o32.abs().max())))
    for name, _, _ in layers:
        print("   %s  w cos %.5f maxerr/max %.4f | b cos %.5f maxerr/max %.4f" % (
            name, cos(gr[name][0], g32[name][0]), float((gr[name][0] - g32[name][0]).abs().max()) / float(g32[name][0].abs().max()),
            cos(gr[name][1], g32[name][1]), float((gr[name][1] - g32[name][1]).abs().max()) / float(g32[name][1].abs().max())))
