"""`nms(dets, thresh, force_cpu=False)`: dispatcher interface of
lib/fast_rcnn/nms_wrapper.py:13-21.

Both branches run on the MI355X.  The north-star parity target is the reference's CPU
path, so the default (and `force_cpu=True`) is the cpu_nms rule; the CUDA kernel's
slightly different rule is reachable with cfg.USE_GPU_NMS == 'cuda_rule'."""
from ..nms.cpu_nms import cpu_nms
from ..nms.gpu_nms import gpu_nms
from .config import cfg


def nms(dets, thresh, force_cpu=False):
    """Dispatch to either rule; empty input -> [] (nms_wrapper.py:16-17)."""
    if dets.shape[0] == 0:
        return []
    if cfg.USE_GPU_NMS == 'cuda_rule' and not force_cpu:
        return gpu_nms(dets, thresh, device_id=cfg.GPU_ID)
    return cpu_nms(dets, thresh, device_id=cfg.GPU_ID)
