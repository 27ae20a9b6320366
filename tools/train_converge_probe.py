"""A few Adam steps on ONE fixed frame with frozen sampling, fp32 vs the bf16 MFMA trunk: do both losses fall alike?
    python tools/train_converge_probe.py [steps] [lr]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mv3d_tf_amd import build, synth  # noqa: E402
from mv3d_tf_amd.fast_rcnn.train_mv import total_loss  # noqa: E402
from mv3d_tf_amd.networks import get_network  # noqa: E402

build.build()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4
rng = np.random.RandomState(2)
gt = synth.gt_cars(np.random.RandomState(31), 4)
feed = {"lidar_bv_data": ((rng.random_sample((1, 608, 608, 9)) < 0.05) * rng.uniform(0, 2.4, (1, 608, 608, 9))).astype(np.float32),
        "image_data": rng.uniform(-1, 1, (1, 375, 1242, 3)).astype(np.float32), "im_info": np.array([[608, 608, 1]], np.float32),
        "calib": synth.KITTI_CALIB[None], "gt_boxes_bv": gt[0], "gt_boxes_3d": gt[1], "gt_boxes_corners": gt[2], "keep_prob": 1.0}
out = {}
for mixed in (False, True):
    net = get_network("MV3D_train")
    g = torch.Generator(device="cuda").manual_seed(21)
    with torch.no_grad():
        for name, (w, b) in net.params.items():
            if w.ndim == 4 and w.shape[2] == 3:
                w.copy_(torch.randn(w.shape, device="cuda", generator=g) * (2.0 / (w.shape[1] * 9)) ** 0.5)
        net.params["rpn_cls_score"][0].mul_(20.0)
    net.mfma_trunk, net.amp_dtype = mixed, (torch.bfloat16 if mixed else None)
    opt = torch.optim.Adam(net.parameters(), lr=lr, fused=True)
    hist = []
    for it in range(steps):
        np.random.seed(4)                      # frozen anchor / ROI sampling
        opt.zero_grad(set_to_none=True)
        loss, parts = total_loss(net.forward(feed))
        loss.backward()
        opt.step()
        hist.append(float(loss.detach()))
    out[mixed] = hist
    print("mixed" if mixed else "fp32 ", " ".join("%.4f" % v for v in hist), flush=True)
