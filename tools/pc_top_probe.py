import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from mv3d_tf_amd import ops, synth
pts = torch.as_tensor(synth.point_cloud(1, 120000)).cuda()
for _ in range(3): ops.point_cloud_2_top(pts)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50): ops.point_cloud_2_top(pts)
b.record(); torch.cuda.synchronize()
print("point_cloud_2_top 120k points: %.1f us/call" % (a.elapsed_time(b) / 50 * 1e3))
