"""Two gloo ranks sharing ONE GPU: the mixed-precision training step with the trunks on one stream vs on side streams (GradBucketer
fencing each bucket by its gradients' stream events): step time, and are the averaged gradients the same bits?
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29617 tools/dp_streams_probe.py"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mv3d_tf_amd import sharding, synth  # noqa: E402
from mv3d_tf_amd.fast_rcnn.train_mv import bench_train_step, stack_blobs, total_loss  # noqa: E402
from mv3d_tf_amd.networks import get_network  # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)


def grads(streams_dp):
    torch.manual_seed(0)
    net = get_network("MV3D_train")
    net.mfma_trunk, net.amp_dtype, net.trunk_streams_dp = True, torch.bfloat16, streams_dp
    params = net.parameters()
    b = sharding.GradBucketer(params, dist)
    rng = np.random.RandomState(10 + rank)
    gt = synth.gt_cars(np.random.RandomState(31 + rank), 4)
    feed = {"lidar_bv_data": ((rng.random_sample((1, 608, 608, 9)) < 0.05) * rng.uniform(0, 2.4, (1, 608, 608, 9))).astype(np.float32),
            "image_data": rng.uniform(-1, 1, (1, 375, 1242, 3)).astype(np.float32), "im_info": np.array([[608, 608, 1]], np.float32),
            "calib": synth.KITTI_CALIB[None], "gt_boxes_bv": gt[0], "gt_boxes_3d": gt[1], "gt_boxes_corners": gt[2], "keep_prob": 1.0}
    out = None
    for _ in range(2):
        np.random.seed(4 + rank)
        b.zero_grad()
        b.reset()
        loss, _ = total_loss(net.forward(feed))
        loss.backward()
        b.finish()
        torch.cuda.synchronize()
        out = [p.grad.detach().clone() for p in params]
    b.close()
    return out


a = grads(False)
c = grads(True)
same = all(torch.equal(x, y) for x, y in zip(a, c))
worst = max(float((x - y).abs().max()) for x, y in zip(a, c))
for mode in (() if "--no-timing" in sys.argv else (True, "dp", True, "dp")):
    r = bench_train_step(rank, world, dist, steps=4, warmup=2, amp=torch.bfloat16, mfma=True, trunk_streams=mode)
    if rank == 0:
        print("streams under DP" if mode == "dp" else "one stream      ", r["ms_per_step"], flush=True)
if rank == 0:
    print("averaged gradients bit-identical:", same, "max abs diff", worst, flush=True)
dist.barrier()
dist.destroy_process_group()
