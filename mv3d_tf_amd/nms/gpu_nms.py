"""`gpu_nms(dets, thresh, device_id=0)`: interface of lib/nms/gpu_nms.pyx:16-31.

Like the reference it argsorts on the host, hands pre-sorted boxes to the C symbol `_nms`
(lib/nms/gpu_nms.hpp:1-2, here exported by libmv3d_hip.so) and maps the kept positions back
through `order`.  `_nms` keeps the CUDA kernel's rule (IoU > thresh in f32,
lib/nms/nms_kernel.cu:71), which is NOT the cpu_nms rule; `nms_wrapper.nms` therefore
routes to the CPU-rule device path by default (the parity target)."""
import numpy as np

from .. import ops


def gpu_nms(dets, thresh, device_id=0):
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    order = dets[:, 4].argsort()[::-1]
    sorted_dets = np.ascontiguousarray(dets[order, :])
    keep = ops.nms_gpu_rule_host(sorted_dets, np.float32(thresh), device_id)
    return list(order[keep])
