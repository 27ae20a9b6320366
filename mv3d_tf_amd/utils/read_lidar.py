"""`point_cloud_2_top`: interface of lib/utils/read_lidar.py:10-115 (== tools/read_lidar.py), every parameter of it.

The call MV3D makes (tools/read_lidar.py:121-133: res=0.1, zres=0.3, side_range=(-30,30), fwd_range=(0,60), height_range=(-2,0.4) ->
(601,601,9)) goes through mv3d_point_cloud_2_top, any other parameter set through mv3d_point_cloud_2_top_ranges (which reports the
cells numpy would refuse: IndexError here as there)."""
import numpy as np
import torch

from .. import ops

_MV3D = (0.1, 0.3, -30., 30., 0., 60., -2., 0.4)


def point_cloud_2_top(points, res=0.1, zres=0.3, side_range=(-10., 10.), fwd_range=(-10., 10.), height_range=(-2., 2.)):
    """defaults as lib/utils/read_lidar.py:10-16"""
    got = (float(res), float(zres), float(side_range[0]), float(side_range[1]), float(fwd_range[0]), float(fwd_range[1]),
           float(height_range[0]), float(height_range[1]))
    as_numpy = not isinstance(points, torch.Tensor)
    pts = ops._dev(np.ascontiguousarray(points, dtype=np.float32)[:, :4] if as_numpy else points[:, :4])
    top = ops.point_cloud_2_top(pts, None if got == _MV3D else got)
    return top.cpu().numpy() if as_numpy else top
