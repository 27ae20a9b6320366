"""`cpu_nms(dets, thresh)` with the semantics of lib/nms/cpu_nms.pyx:17-68 (f32 IoU, +1
pixel convention, Python-float `>=` compare, result = indices into `dets` in descending
score order) -- executed on the MI355X through the C-ABI (mv3d_nms_host).  The name is
kept because callers import it; there is no host implementation in this package."""
import numpy as np

from .. import ops


def cpu_nms(dets, thresh, device_id=0):
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    if dets.ndim != 2 or dets.shape[1] != 5:
        raise ValueError("Buffer has wrong number of dimensions (expected 2) or shape (n, 5)")
    return ops.nms_host(dets, float(thresh), device_id)
