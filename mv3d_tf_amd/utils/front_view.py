"""Front-view (FV) ROIs: the third view of MV3D, which the reference leaves a TODO
(`proposal_transform` in lib/networks/network.py:293-315 returns None for anything but 'bv' / 'img').

`rois_3d_to_fv(rois_3d)`: (R,7) [b,x,y,z,l,w,h] -> (R,5) [b,x1,y1,x2,y2] on the 64 x 512 cylindrical front-view
map of the MV3D paper; computed on the device (csrc/front_view.hip).  PARITY UNPINNED: there is no reference code
to pin it to; the definition is documented in the kernel file and restated in the oracle."""
import numpy as np
import torch

from .. import ops

FV_HEIGHT, FV_WIDTH = 64, 512          # BASELINE.json: "512x64 FV"


def rois_3d_to_fv(rois_3d):
    as_numpy = not isinstance(rois_3d, torch.Tensor)
    r3 = ops._dev(np.asarray(rois_3d, np.float32).reshape(-1, 7) if as_numpy else rois_3d.reshape(-1, 7))
    out = ops.rois_3d_to_fv(r3)
    return out.cpu().numpy() if as_numpy else out
