"""`proposal_target_layer_3d`: the callable that lib/networks/network.py:256-273 wraps in
tf.py_func; same name / arguments / 5-tuple as lib/rpn_msr/proposal_target_layer_tf.py:19.

IoU, assignment, list compaction, corners, corner targets and the image projection run on
the device (csrc/proposal_target.hip); the two `npr.choice(..., replace=False)` draws come
from the numpy GLOBAL RNG on the host, in the reference's order (fg first, then bg), as
`inds[npr.permutation(len(inds))[:k]]`."""
import numpy as np
import numpy.random as npr
import torch

from .. import ops
from .._lib import ProposalTargetParams
from ..fast_rcnn.config import cfg


def draw_samples(counts):
    """The two `npr.choice(..., replace=False)` draws of _sample_rois_3d (:246-269) from the numpy GLOBAL RNG given stage 1's
    device counts: positions in the fg / bg candidate lists."""
    T = cfg.TRAIN
    _, n_fg, n_bg, _ = (int(v) for v in counts.cpu().numpy())                    # the one host sync
    rois_per_image = T.BATCH_SIZE // 1                                            # :55-57
    fg_rois_per_image = np.round(T.FG_FRACTION * rois_per_image)
    fg_n = int(min(fg_rois_per_image, n_fg))                                      # :251
    fg_pick = npr.permutation(n_fg)[:fg_n] if n_fg > 0 else np.zeros(0, np.int64)   # :253-255
    bg_n = int(min(rois_per_image - fg_n, n_bg))                                  # :264-266
    bg_pick = npr.permutation(n_bg)[:bg_n] if n_bg > 0 else np.zeros(0, np.int64)   # :268-269
    return fg_pick, bg_pick


def proposal_target_layer_3d(rpn_rois_bv, rpn_rois_3d, gt_boxes_bv, gt_boxes_3d, gt_boxes_corners, calib, _num_classes):
    """Returns (rois_bv (S,5) f32, rois_img (S,5) f32, labels (S,1) i32, bbox_targets (S,24*nc) f32,
    rois_3d (S,7) f32), S <= cfg.TRAIN.BATCH_SIZE, foreground rows first."""
    as_numpy = not isinstance(rpn_rois_bv, torch.Tensor)
    dev = rpn_rois_bv.device if not as_numpy else torch.device("cuda", cfg.GPU_ID)
    if as_numpy:
        # numpy contract: the six inputs travel in one upload
        f = lambda a, c: np.asarray(a, np.float32).reshape(-1, c)
        rois_bv, rois_3d, gt_bv, gt_3d, gt_cnr, cal = ops.upload_packed(
            [f(rpn_rois_bv, 5), f(rpn_rois_3d, 7), f(gt_boxes_bv, 5), f(gt_boxes_3d, 7), f(gt_boxes_corners, 25),
             np.asarray(calib, np.float32).reshape(4, 12)], dev)
    else:
        rois_bv = ops._dev(rpn_rois_bv, device=dev).reshape(-1, 5)
        rois_3d = ops._dev(rpn_rois_3d, device=dev).reshape(-1, 7)
        gt_bv = ops._dev(gt_boxes_bv, device=dev).reshape(-1, 5)
        gt_3d = ops._dev(gt_boxes_3d, device=dev).reshape(-1, 7)
        gt_cnr = ops._dev(gt_boxes_corners, device=dev).reshape(-1, 25)
        cal = ops._dev(calib, device=dev).reshape(4, 12)
    T = cfg.TRAIN
    if as_numpy:
        # Sanity check of the reference (:52-53): single batch only
        assert np.all(np.asarray(rpn_rois_bv).reshape(-1, 5)[:, 0] == 0), 'Only single item batches are supported'
    nc = int(_num_classes)
    params = ProposalTargetParams(nc, 0, float(T.FG_THRESH), float(T.BG_THRESH_HI), float(T.BG_THRESH_LO))
    counts, ws = ops.proposal_target_stage1(rois_bv, rois_3d, gt_bv, gt_3d, params)
    fg_pick, bg_pick = draw_samples(counts)
    if as_numpy:
        spec = ops.proposal_target_spec(len(fg_pick) + len(bg_pick), nc)
        pack, views = ops.packed_views(spec, dev)
        ops.proposal_target_stage2(rois_bv, rois_3d, gt_bv, gt_3d, gt_cnr, cal, params, fg_pick, bg_pick, ws, out=tuple(views))
        return tuple(ops.unpack_host(pack, spec))              # ONE device-to-host copy
    return ops.proposal_target_stage2(rois_bv, rois_3d, gt_bv, gt_3d, gt_cnr, cal, params, fg_pick, bg_pick, ws)
