"""Two gloo ranks sharing ONE GPU run the mixed-precision training step (grouped trunk launches on one stream, bucketed gradient
all-reduce overlapping backward): are the averaged gradients the same bits on both ranks, and the same bits when the step is run again?
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29617 tools/dp_grad_probe.py"""
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


def grads():
    torch.manual_seed(0)
    net = get_network("MV3D_train")
    net.mfma_trunk, net.amp_dtype = True, torch.bfloat16
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


a = grads()
c = grads()
same = all(torch.equal(x, y) for x, y in zip(a, c))
worst = max(float((x - y).abs().max()) for x, y in zip(a, c))
flat = torch.cat([x.reshape(-1).double().cpu() for x in a])
both = [torch.zeros_like(flat) for _ in range(world)]
dist.all_gather(both, flat)
across = all(torch.equal(both[0], t) for t in both[1:])
if "--no-timing" not in sys.argv:
    r = bench_train_step(rank, world, dist, steps=4, warmup=2, amp=torch.bfloat16, mfma=True)
    if rank == 0:
        print("ms per step (2 gloo ranks on one GPU):", r["ms_per_step"], flush=True)
if rank == 0:
    print("averaged gradients bit-identical:", same and across, "(run to run:", same, "max abs diff", worst, "; across ranks:", across, ")", flush=True)
dist.barrier()
dist.destroy_process_group()
