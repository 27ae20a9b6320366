"""ctypes binding of libmv3d_hip.so (include/mv3d_hip.h).

The HIP library is the product: there is no CPU or PyTorch fallback.  If the shared
object is missing or cannot be loaded, importing any operator raises immediately.
torch is imported first so that the one HIP runtime already loaded by torch
(libamdhip64.so.7) also serves this library: device pointers and stream handles are
shared with torch tensors, which are used purely as buffers.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede CDLL: provides libamdhip64.so.7)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmv3d_hip.so")

OK, ERR_INVALID_ARG, ERR_WORKSPACE, ERR_HIP, ERR_ZERO_DIVISION = 0, 1, 2, 3, 4


class ProposalParams(C.Structure):
    """mv3d_proposal_params"""
    _fields_ = [("feat_stride", C.c_int32), ("pre_nms_topN", C.c_int32), ("post_nms_topN", C.c_int32),
                ("img_height", C.c_int32), ("img_width", C.c_int32), ("img_padding", C.c_int32),
                ("nms_strict_gt", C.c_int32), ("reserved0", C.c_int32), ("nms_thresh", C.c_double), ("min_size", C.c_double)]


class AnchorTargetParams(C.Structure):
    """mv3d_anchor_target_params"""
    _fields_ = [("feat_stride", C.c_int32), ("clobber_positives", C.c_int32),
                ("negative_overlap", C.c_double), ("positive_overlap", C.c_double)]


class RoiView(C.Structure):
    """mv3d_roi_view"""
    _fields_ = [("bottom_data", C.c_void_p), ("bottom_rois", C.c_void_p), ("top_data", C.c_void_p),
                ("argmax_data", C.c_void_p), ("spatial_scale", C.c_float), ("batch_size", C.c_int32),
                ("num_rois", C.c_int32), ("height", C.c_int32), ("width", C.c_int32), ("channels", C.c_int32)]


class RoiGradView(C.Structure):
    """mv3d_roi_grad_view"""
    _fields_ = [("bottom_diff", C.c_void_p), ("bottom_rois", C.c_void_p), ("top_diff", C.c_void_p),
                ("argmax_data", C.c_void_p), ("spatial_scale", C.c_float), ("batch_size", C.c_int32),
                ("num_rois", C.c_int32), ("height", C.c_int32), ("width", C.c_int32), ("channels", C.c_int32)]


class ProposalTargetParams(C.Structure):
    """mv3d_proposal_target_params"""
    _fields_ = [("num_classes", C.c_int32), ("frame_index", C.c_int32), ("fg_thresh", C.c_double),
                ("bg_thresh_hi", C.c_double), ("bg_thresh_lo", C.c_double)]


class DrawFrame(C.Structure):
    """mv3d_draw_frame"""
    _fields_ = [("n_fg", C.c_int32), ("n_bg", C.c_int32), ("n_low", C.c_int32), ("pt_n_fg", C.c_int32), ("pt_n_bg", C.c_int32),
                ("reserved0", C.c_int32), ("fg_alive", C.c_void_p)]


class DrawParams(C.Structure):
    """mv3d_draw_params"""
    _fields_ = [("rpn_batchsize", C.c_int32), ("rpn_num_fg", C.c_int32), ("rois_per_image", C.c_int32), ("roi_fg_max", C.c_int32)]


class TrainPathConfig(C.Structure):
    """mv3d_train_path_config"""
    _fields_ = [("batch", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("num_classes", C.c_int32), ("proposal_cap", C.c_int32),
                ("anchor_cap", C.c_int32), ("roi_cap", C.c_int32), ("max_gt", C.c_int32), ("proposal", ProposalParams),
                ("anchor", AnchorTargetParams), ("target", ProposalTargetParams), ("draw", DrawParams)]


class TrainPathSlot(C.Structure):
    """mv3d_train_path_slot"""
    _fields_ = [("blob_bv", C.c_void_p), ("blob_img", C.c_void_p), ("blob_3d", C.c_void_p), ("num_proposals", C.c_void_p),
                ("proposal_ws", C.c_void_p), ("proposal_ws_bytes", C.c_size_t), ("rpn_labels", C.c_void_p), ("rpn_targets", C.c_void_p),
                ("anchors", C.c_void_p), ("anchors_3d", C.c_void_p), ("n_anchors", C.c_void_p), ("report", C.c_void_p),
                ("report_row", C.c_size_t), ("pt_counts", C.c_void_p), ("anchor_ws", C.c_void_p), ("anchor_ws_bytes", C.c_size_t),
                ("target_ws", C.c_void_p), ("target_ws_bytes", C.c_size_t), ("rois_bev", C.c_void_p), ("rois_rgb", C.c_void_p),
                ("rois_fv", C.c_void_p), ("rois_3d", C.c_void_p), ("labels", C.c_void_p), ("bbox_targets", C.c_void_p),
                ("lists", C.c_void_p), ("lists_cap", C.c_size_t), ("h_report", C.c_void_p), ("h_report_row", C.c_size_t),
                ("h_pt_counts", C.c_void_p), ("h_num_proposals", C.c_void_p), ("h_lists", C.c_void_p), ("h_scratch", C.c_void_p),
                ("scratch_cap", C.c_size_t)]


class ConvView(C.Structure):
    """mv3d_conv_view"""
    _fields_ = [("x_framed", C.c_void_p), ("w_packed", C.c_void_p), ("bias", C.c_void_p), ("gate_framed", C.c_void_p), ("y", C.c_void_p),
                ("batch", C.c_int32), ("height", C.c_int32), ("width", C.c_int32), ("reserved0", C.c_int32)]


class PoolView(C.Structure):
    """mv3d_pool_view"""
    _fields_ = [("x_framed", C.c_void_p), ("g_pooled_framed", C.c_void_p), ("y_framed", C.c_void_p),
                ("batch", C.c_int32), ("height", C.c_int32), ("width", C.c_int32), ("reserved0", C.c_int32)]


class WgradView(C.Structure):
    """mv3d_wgrad_view"""
    _fields_ = [("x_framed", C.c_void_p), ("dy_framed", C.c_void_p), ("dw", C.c_void_p), ("db", C.c_void_p),
                ("batch", C.c_int32), ("height", C.c_int32), ("width", C.c_int32), ("c_in_real", C.c_int32)]


class PackItem(C.Structure):
    """mv3d_pack_item"""
    _fields_ = [("w_oihw", C.c_void_p), ("fwd_packed", C.c_void_p), ("dgrad_packed", C.c_void_p),
                ("c_out", C.c_int32), ("c_in", C.c_int32), ("c_in_pad", C.c_int32), ("reserved0", C.c_int32)]


class AdamTensor(C.Structure):
    """mv3d_adam_tensor"""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("numel", C.c_longlong),
                ("param_lowp", C.c_void_p)]


_P = C.c_void_p
_SIGS = {
    "mv3d_version": (C.c_int, []),
    "mv3d_status_string": (C.c_char_p, [C.c_int]),
    "mv3d_nms_workspace_bytes": (C.c_size_t, [C.c_int]),
    "mv3d_nms_device": (C.c_int, [_P, C.c_int, C.c_double, C.c_int, _P, _P, _P, _P, C.c_size_t, _P]),
    "mv3d_nms_device_trace": (C.c_int, [_P, C.c_int, C.c_double, C.c_int, _P, _P, _P, _P, C.c_size_t, _P, _P]),
    "mv3d_nms_host": (C.c_int, [_P, _P, _P, C.c_int, C.c_double, C.c_int]),
    "_nms": (None, [_P, _P, _P, C.c_int, C.c_int, C.c_float, C.c_int]),
    "mv3d_proposal_3d_capacity": (C.c_int, [C.c_int, C.c_int, C.POINTER(ProposalParams)]),
    "mv3d_proposal_3d_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.POINTER(ProposalParams)]),
    "mv3d_proposal_3d": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.POINTER(ProposalParams),
                                   _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "mv3d_roi_pool_forward": (C.c_int, [_P, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, _P, _P, _P, _P]),
    "mv3d_roi_pool_backward": (C.c_int, [_P, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_int, _P, _P, _P, _P]),
    "mv3d_anchor_target_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "mv3d_anchor_target_stage1": (C.c_int, [C.c_int, C.c_int, _P, _P, _P, C.c_int, C.POINTER(AnchorTargetParams),
                                            _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "mv3d_anchor_target_stage2": (C.c_int, [C.c_int, C.c_int, C.POINTER(AnchorTargetParams), _P, C.c_int, _P,
                                            C.c_int, _P, C.c_int, _P, _P, _P, _P, C.c_int, _P, C.c_size_t, _P]),
    "mv3d_anchor_target_stage1_batch": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, C.POINTER(AnchorTargetParams), _P, _P, _P, _P,
                                                  _P, C.c_size_t, _P]),
    "mv3d_anchor_target_stage2_batch": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(AnchorTargetParams), _P, _P, _P, _P, _P, _P, _P,
                                                  _P, _P, _P, C.c_int, _P, C.c_size_t, _P]),
    "mv3d_proposal_target_stage1_batch": (C.c_int, [C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mv3d_proposal_target_stage2_batch": (C.c_int, [C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                                    _P, _P, _P]),
    "mv3d_proposal_target_stage1_batch_devn": (C.c_int, [C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mv3d_proposal_target_stage2_batch_devn": (C.c_int, [C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                                         _P, _P, _P, _P, _P]),
    "mv3d_proposal_target_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "mv3d_proposal_target_stage1": (C.c_int, [_P, _P, C.c_int, _P, _P, C.c_int, C.POINTER(ProposalTargetParams), _P, _P,
                                              C.c_size_t, _P]),
    "mv3d_proposal_target_stage2": (C.c_int, [_P, _P, C.c_int, _P, _P, _P, C.c_int, _P, C.POINTER(ProposalTargetParams),
                                              _P, C.c_int, _P, C.c_int, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "mv3d_point_cloud_2_top": (C.c_int, [_P, C.c_int, _P, _P]),
    "mv3d_point_cloud_2_top_shape": (C.c_int, [C.c_double] * 8 + [_P]),
    "mv3d_point_cloud_2_top_ranges": (C.c_int, [_P, C.c_int] + [C.c_double] * 8 + [_P, _P, _P]),
    "mv3d_box_detect_tail": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "mv3d_loss_workspace_bytes": (C.c_size_t, [C.c_int]),
    "mv3d_rpn_loss": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_float, _P, _P, _P, _P, C.c_size_t, _P]),
    "mv3d_rcnn_loss": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_float, _P, _P, _P, _P, C.c_size_t, _P]),
    "mv3d_roi_pool_forward_views": (C.c_int, [C.c_int, C.POINTER(RoiView), C.c_int, C.c_int, _P]),
    "mv3d_roi_pool_forward_views_cold": (C.c_int, [C.c_int, C.POINTER(RoiView), C.c_int, C.c_int, _P]),
    "mv3d_roi_pool_forward_views_half": (C.c_int, [C.c_int, C.POINTER(RoiView), C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mv3d_roi_pool_backward_workspace_bytes": (C.c_size_t, [C.c_int, C.POINTER(RoiGradView), C.c_int, C.c_int]),
    "mv3d_roi_pool_backward_views": (C.c_int, [C.c_int, C.POINTER(RoiGradView), C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "mv3d_roi_pool_pair_workspace_bytes": (C.c_size_t, [C.c_int, C.POINTER(RoiGradView), C.c_int, C.c_int]),
    "mv3d_roi_pool_forward_views_pair": (C.c_int, [C.c_int, C.POINTER(RoiView), C.c_int, C.c_int, C.c_int, _P]),
    "mv3d_roi_pool_backward_views_pair": (C.c_int, [C.c_int, C.POINTER(RoiGradView), C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "mv3d_roi_pool_argmax_decode": (C.c_int, [C.c_int, C.POINTER(RoiView), C.c_int, C.c_int, C.POINTER(C.c_void_p), _P]),
    "mv3d_ROIPoolForwardLaucher": (C.c_int, [_P, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "mv3d_ROIPoolBackwardLaucher": (C.c_int, [_P, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "mv3d_rois_3d_to_fv": (C.c_int, [_P, C.c_int, _P, _P]),
    "mv3d_legacy_permutation": (C.c_int, [_P, C.c_int32, _P]),
    "mv3d_draw_training_subsamples": (C.c_int, [_P, C.c_int, C.POINTER(DrawFrame), C.POINTER(DrawParams), _P, C.c_size_t, _P, _P,
                                                C.c_size_t]),
    "mv3d_train_path_create": (C.c_int, [C.POINTER(TrainPathConfig), C.c_int, C.POINTER(TrainPathSlot), C.c_int, C.POINTER(C.c_void_p)]),
    "mv3d_train_path_configure": (C.c_int, [_P, C.POINTER(TrainPathConfig)]),
    "mv3d_train_path_submit": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mv3d_train_path_finish": (C.c_int, [_P, C.c_int, _P, _P, _P]),
    "mv3d_train_path_host_seconds": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "mv3d_train_path_destroy": (None, [_P]),
    "mv3d_gt_encode": (C.c_int, [_P, _P, C.c_int, _P, _P, _P, _P, _P, _P, _P]),
    "mv3d_conv3x3_f16": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mv3d_conv3x3_f32": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mv3d_maxpool2x2_f32": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mv3d_frame_nhwc_f32": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mv3d_conv3x3_bf16": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mv3d_maxpool2x2_bf16": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mv3d_frame_nhwc_bf16": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mv3d_conv3x3_wgrad_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mv3d_conv3x3_wgrad_bf16": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "mv3d_conv3x3_wgrad_f32_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mv3d_conv3x3_wgrad_f32": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "mv3d_conv3x3_pack_bf16": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "mv3d_conv3x3_gated_bf16": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mv3d_maxpool2x2_bwd_f32": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mv3d_maxpool2x2_bwd_bf16": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mv3d_conv3x3_views_f16": (C.c_int, [C.c_int, C.POINTER(ConvView), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mv3d_conv3x3_views_bf16": (C.c_int, [C.c_int, C.POINTER(ConvView), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mv3d_conv3x3_views_f32": (C.c_int, [C.c_int, C.POINTER(ConvView), C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mv3d_conv3x3_pool_views_f16": (C.c_int, [C.c_int, C.POINTER(ConvView), C.c_int, C.c_int, _P]),
    "mv3d_conv3x3_pool_views_bf16": (C.c_int, [C.c_int, C.POINTER(ConvView), C.c_int, C.c_int, _P]),
    "mv3d_maxpool2x2_views_f16": (C.c_int, [C.c_int, C.POINTER(PoolView), C.c_int, _P]),
    "mv3d_maxpool2x2_views_bf16": (C.c_int, [C.c_int, C.POINTER(PoolView), C.c_int, _P]),
    "mv3d_maxpool2x2_views_f32": (C.c_int, [C.c_int, C.POINTER(PoolView), C.c_int, _P]),
    "mv3d_maxpool2x2_bwd_views_bf16": (C.c_int, [C.c_int, C.POINTER(PoolView), C.c_int, _P]),
    "mv3d_maxpool2x2_bwd_views_f32": (C.c_int, [C.c_int, C.POINTER(PoolView), C.c_int, _P]),
    "mv3d_conv3x3_wgrad_views_workspace_bytes": (C.c_size_t, [C.c_int, C.POINTER(WgradView), C.c_int, C.c_int, C.c_int]),
    "mv3d_conv3x3_wgrad_views_bf16": (C.c_int, [C.c_int, C.POINTER(WgradView), C.c_int, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "mv3d_conv3x3_wgrad_views_f32": (C.c_int, [C.c_int, C.POINTER(WgradView), C.c_int, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "mv3d_conv3x3_pack_many_bf16": (C.c_int, [C.c_int, C.POINTER(PackItem), _P]),
    "mv3d_conv3x3_pack_many_f32": (C.c_int, [C.c_int, C.POINTER(PackItem), _P]),
    "mv3d_maxpool2x2_f16": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mv3d_frame_nhwc_f16": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "mv3d_softmax_rows": (C.c_int, [_P, _P, C.c_longlong, C.c_int, _P]),
    "mv3d_adam_chunk_elements": (C.c_int, []),
    "mv3d_adam_step": (C.c_int, [_P, _P, _P, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, _P]),
}
EXPORTS = tuple(_SIGS)

_lib = None


def lib():
    """The loaded library; raises loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension is the product and has no fallback. "
                "Build it with `python -m mv3d_tf_amd.build` (or __graft_entry__.build()).")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(handle, name)       # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


class Mv3dError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        super().__init__(f"{where}: {lib().mv3d_status_string(status).decode()} (status {status})")


def check(status, where):
    if status == ERR_ZERO_DIVISION:
        raise ZeroDivisionError("float division")     # what lib/nms/cpu_nms.pyx:64 raises
    if status != OK:
        raise Mv3dError(status, where)
