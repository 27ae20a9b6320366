"""`nms(dets, thresh, force_cpu=False)`: dispatcher interface of lib/fast_rcnn/nms_wrapper.py:13-21.

Both branches run on the MI355X.  cfg.USE_GPU_NMS is honoured exactly like the reference honours it: True (and not
force_cpu) -> the gpu_nms rule (IoU > thresh in f32, lib/nms/nms_kernel.cu:71), otherwise the cpu_nms rule
((double)IoU >= thresh, lib/nms/cpu_nms.pyx:65).  The default of cfg.USE_GPU_NMS here is False because the parity target is
the reference's CPU path (see fast_rcnn/config.py)."""
from ..nms.cpu_nms import cpu_nms
from ..nms.gpu_nms import gpu_nms
from .config import cfg


def nms(dets, thresh, force_cpu=False):
    """Dispatch to either rule; empty input -> [] (nms_wrapper.py:16-17)."""
    if dets.shape[0] == 0:
        return []
    if cfg.USE_GPU_NMS and not force_cpu:
        return gpu_nms(dets, thresh, device_id=cfg.GPU_ID)
    return cpu_nms(dets, thresh, device_id=cfg.GPU_ID)
