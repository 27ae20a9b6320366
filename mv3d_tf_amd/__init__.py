"""mv3d_tf_amd -- MI355X (gfx950) implementation of the MV3D RPN -> ROI-pool -> NMS hot
path behind the reference's own operator interfaces (see DESIGN.md / INTEGRATION.md).

Layout mirrors the reference's `lib/` for the path only:
  fast_rcnn/{config,nms_wrapper}.py   rpn_msr/{proposal_layer_tf,anchor_target_layer_tf}.py
  roi_pooling_layer/roi_pooling_op.py nms/{cpu_nms,gpu_nms}.py
  csrc/ (HIP kernels + C-ABI), _lib.py (ctypes), ops.py (tensor-level calls), synth.py
"""
__version__ = "0.1.0"
