"""`point_cloud_2_top`: interface of lib/utils/read_lidar.py:10-115 (== tools/read_lidar.py).

Only the parameter set MV3D uses (tools/read_lidar.py:121-133) is implemented on the device:
res=0.1, zres=0.3, side_range=(-30,30), fwd_range=(0,60), height_range=(-2,0.4) -> (601,601,9)."""
import numpy as np
import torch

from .. import ops

_MV3D = dict(res=0.1, zres=0.3, side_range=(-30., 30.), fwd_range=(0., 60.), height_range=(-2., 0.4))


def point_cloud_2_top(points, res=0.1, zres=0.3, side_range=(-30., 30.), fwd_range=(0., 60), height_range=(-2, 0.4)):
    got = dict(res=float(res), zres=float(zres), side_range=tuple(map(float, side_range)),
               fwd_range=tuple(map(float, fwd_range)), height_range=tuple(map(float, height_range)))
    if got != _MV3D:
        raise NotImplementedError("point_cloud_2_top: only the MV3D ranges of tools/read_lidar.py:121-133 are built")
    as_numpy = not isinstance(points, torch.Tensor)
    pts = ops._dev(np.ascontiguousarray(points, dtype=np.float32)[:, :4] if as_numpy else points[:, :4])
    top = ops.point_cloud_2_top(pts)
    return top.cpu().numpy() if as_numpy else top
