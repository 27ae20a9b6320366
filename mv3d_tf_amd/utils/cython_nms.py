"""`nms(dets, thresh)` used for the final per-class detection NMS
(lib/utils/nms.pyx:17-68, imported at lib/fast_rcnn/test_mv.py:6): same algorithm as
cpu_nms, same device implementation."""
from ..nms.cpu_nms import cpu_nms as nms  # noqa: F401
