"""`proposal_layer_3d`: the callable that lib/networks/network.py:221-234 wraps in
tf.py_func, same name / arguments / return tuple as lib/rpn_msr/proposal_layer_tf.py:25.

Inputs may be numpy arrays (the py_func contract: host arrays in, fresh host arrays out,
one frame) or torch device tensors (no host round trip, any batch).  All arithmetic runs
in libmv3d_hip.so (csrc/proposal.hip)."""
import ctypes as C

import numpy as np
import torch

from .. import ops
from .._lib import ERR_INVALID_ARG, check, lib
from ..fast_rcnn.config import cfg


def proposal_layer_3d(rpn_cls_prob_reshape, rpn_bbox_pred, im_info, calib, cfg_key, _feat_stride=[8, ],
                      anchor_scales=[1.0, 1.0]):
    """Returns (blob_bv (R,5), blob_img (R,5), blob_3d (R,7)), f32, R <= RPN_POST_NMS_TOP_N.
    Column 0 is the frame index.  Raises like the reference on batch != 1 for numpy input."""
    if isinstance(cfg_key, bytes):
        cfg_key = cfg_key.decode()
    as_numpy = not isinstance(rpn_cls_prob_reshape, torch.Tensor)
    if as_numpy:
        assert rpn_cls_prob_reshape.shape[0] == 1, 'Only single item batches are supported'
    prob = ops._dev(rpn_cls_prob_reshape)
    pred = ops._dev(rpn_bbox_pred, device=prob.device)
    B = prob.shape[0]
    info = ops._dev(im_info, device=prob.device).reshape(-1, 3)
    cal = ops._dev(calib, device=prob.device).reshape(-1, 4, 12)
    if info.shape[0] == 1 and B > 1:
        info = info.expand(B, 3).contiguous()
    if cal.shape[0] == 1 and B > 1:
        cal = cal.expand(B, 4, 12).contiguous()
    stride = int(np.asarray(_feat_stride).reshape(-1)[0])
    params = ops.proposal_params(cfg[cfg_key], feat_stride=stride)
    H, W = int(prob.shape[1]), int(prob.shape[2])
    cap = lib().mv3d_proposal_3d_capacity(H, W, C.byref(params))
    if cap < 0:
        check(ERR_INVALID_ARG, "mv3d_proposal_3d_capacity")
    pack, out = ops.proposal_3d_outputs(B, cap, prob.device)
    bv, img, b3, num, status = ops.proposal_3d(prob, pred, info, cal, params, out=out)
    if as_numpy:
        # numpy contract (one frame): ONE device-to-host copy of the packed outputs, sliced on the host
        host = pack.cpu().numpy()
        n5, n7 = cap * 5, cap * 7
        tail = host[2 * n5 + n7:].view(np.int32)
        if int(tail[1]) & 1:
            raise ZeroDivisionError("float division")
        r = int(tail[0])
        return (host[:n5].reshape(cap, 5)[:r], host[n5:2 * n5].reshape(cap, 5)[:r],
                host[2 * n5:2 * n5 + n7].reshape(cap, 7)[:r])
    counts = num.cpu().numpy()                     # the only host sync: ROI counts
    if int(status.max().item()) & 1:
        raise ZeroDivisionError("float division")
    if B == 1:
        r = int(counts[0])
        return (bv[0, :r], img[0, :r], b3[0, :r])
    return tuple(torch.cat([t[b, :int(counts[b])] for b in range(B)], 0) for t in (bv, img, b3))


def proposal_layer_3d_fixed(prob, pred, info, cal, cfg_key, feat_stride=8):
    """The serving form of proposal_layer_3d for a captured graph: NO host round trip, fixed shapes.  Device tensors in; returns
    (blob_bv (B * cap, 5), blob_img (B * cap, 5), blob_3d (B * cap, 7), num_out (B) i32, status (B) i32, cap): frame b's proposals
    are rows [b * cap, b * cap + num_out[b]), the rows behind them are ZERO boxes of frame 0 (the kernel writes them: they are pooled
    and scored like any row and masked by whoever reads num_out -- once, after the step).  status bit 0 = the reference's
    ZeroDivisionError (lib/nms/cpu_nms.pyx:64), to be checked with num_out."""
    params = ops.proposal_params(cfg[cfg_key], feat_stride=int(feat_stride))
    B, H, W = int(prob.shape[0]), int(prob.shape[1]), int(prob.shape[2])
    cap = lib().mv3d_proposal_3d_capacity(H, W, C.byref(params))
    if cap < 0:
        check(ERR_INVALID_ARG, "mv3d_proposal_3d_capacity")
    _, out = ops.proposal_3d_outputs(B, cap, prob.device)
    bv, img, b3, num, status = ops.proposal_3d(prob, pred, info, cal, params, out=out)
    return bv.reshape(B * cap, 5), img.reshape(B * cap, 5), b3.reshape(B * cap, 7), num, status, cap
