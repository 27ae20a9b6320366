"""`anchor_target_layer`: the callable that lib/networks/network.py:237-253 wraps in
tf.py_func; same name / arguments / return tuple as
lib/rpn_msr/anchor_target_layer_tf.py:21.

The arithmetic (inside filter, IoU, label rules, 6-d targets, list compaction) runs on
the device (csrc/anchor_target.hip).  The three random subsamplings draw from the numpy
GLOBAL RNG here on the host, in the reference's order, so a seeded run reproduces the
reference draw for draw: `npr.choice(x, size=k, replace=False)` on the legacy generator
is `x[npr.permutation(len(x))[:k]]`."""
import numpy as np
import numpy.random as npr
import torch

from .. import ops
from .._lib import AnchorTargetParams
from ..fast_rcnn.config import cfg


def draw_subsamples(cf, fg_hi, N):
    """The three random subsamplings of anchor_target_layer_tf.py:146-183, drawn from the numpy GLOBAL RNG in the
    reference's order given stage 1's device results (`cf` = counts + per-foreground flags in one buffer).  Returns the
    positions to disable in the fg list, the first bg list and the second bg list (None = no subsampling)."""
    T = cfg.TRAIN
    # host sync #1: the counts and (usually all of) the foreground flags in one copy
    first = min(32 + N, 4096)
    head = cf[:first].cpu().numpy()
    n_inside, n_fg, n_bg, n_low = (int(v) for v in head[:16].view(np.int32))
    # anchor_target_layer_tf.py:146-151
    num_fg = int(T.RPN_FG_FRACTION * T.RPN_BATCHSIZE)
    dis_fg = None
    if n_fg > num_fg:
        dis_fg = npr.permutation(n_fg)[:n_fg - num_fg]
    # :154-159
    num_bg = T.RPN_BATCHSIZE - min(n_fg, num_fg)
    dis_bg1 = None
    if n_bg > num_bg:
        dis_bg1 = npr.permutation(n_bg)[:n_bg - num_bg]
    # :176-183 positives that survive `labels[max_overlaps < RPN_NEGATIVE_OVERLAP] = 0`
    if n_fg:
        if 32 + n_fg <= first:
            alive = head[32:32 + n_fg].astype(bool)
        else:
            alive = fg_hi[:n_fg].cpu().numpy().astype(bool)                        # host sync #2 (rare: > 4064 positives)
        if dis_fg is not None:
            alive[dis_fg] = False
        n_pos = int(alive.sum())
    else:
        n_pos = 0
    num_bg2 = T.RPN_BATCHSIZE - n_pos
    dis_bg2 = None
    if n_low > num_bg2:
        dis_bg2 = npr.permutation(n_low)[:n_low - num_bg2]
    return dis_fg, dis_bg1, dis_bg2


def anchor_target_layer(rpn_cls_score, gt_boxes, gt_boxes_3d, im_info, _feat_stride=[8, ], anchor_scales=[1.0, 1.0]):
    """Returns (rpn_labels (N,), rpn_bbox_targets (N,6), anchors (M,5), anchors_3d (M,7)), f32."""
    assert rpn_cls_score.shape[0] == 1, 'Only single item batches are supported'
    as_numpy = not isinstance(rpn_cls_score, torch.Tensor)
    H, W = int(rpn_cls_score.shape[1]), int(rpn_cls_score.shape[2])
    dev = rpn_cls_score.device if not as_numpy else torch.device("cuda", cfg.GPU_ID)
    if as_numpy:
        # numpy contract: the three small inputs travel in one upload
        gt_bv, gt_3d, info = ops.upload_packed([np.asarray(gt_boxes, np.float32), np.asarray(gt_boxes_3d, np.float32),
                                                np.asarray(im_info, np.float32).reshape(-1)[:3]], dev)
    else:
        gt_bv = ops._dev(gt_boxes, device=dev)
        gt_3d = ops._dev(gt_boxes_3d, device=dev)
        info = ops._dev(im_info, device=dev).reshape(-1)[:3].contiguous()
    T = cfg.TRAIN
    stride = int(np.asarray(_feat_stride).reshape(-1)[0])
    params = AnchorTargetParams(stride, 1 if T.RPN_CLOBBER_POSITIVES else 0, float(T.RPN_NEGATIVE_OVERLAP),
                                float(T.RPN_POSITIVE_OVERLAP))
    N = H * W * 4
    cap = max(int(T.RPN_BATCHSIZE), 1) * 2
    spec = [((N,), torch.float32), ((N, 6), torch.float32), ((cap, 5), torch.float32), ((cap, 7), torch.float32),
            ((1,), torch.int32)]
    pack, (labels, targets, anchors, anchors_3d, n_anc) = ops.packed_views(spec, dev)   # all outputs: one buffer
    labels, targets, counts, fg_hi, ws, cf = ops.anchor_target_stage1(H, W, info, gt_bv, gt_3d, params, labels, targets)
    dis_fg, dis_bg1, dis_bg2 = draw_subsamples(cf, fg_hi, N)
    ops.anchor_target_stage2(H, W, params, dis_fg, dis_bg1, dis_bg2, labels, ws, cap, out=(anchors, anchors_3d, n_anc))
    if as_numpy:
        h_labels, h_targets, h_anchors, h_anchors_3d, h_n = ops.unpack_host(pack, spec)   # ONE device-to-host copy
        m = int(h_n[0])
        if m > cap:
            raise RuntimeError("anchor_target_layer: %d labelled anchors exceed capacity %d" % (m, cap))
        return (h_labels, h_targets, h_anchors[:m], h_anchors_3d[:m])
    m = int(n_anc.item())
    if m > cap:                                      # more than 2*RPN_BATCHSIZE debug rows
        raise RuntimeError("anchor_target_layer: %d labelled anchors exceed capacity %d" % (m, cap))
    return (labels, targets, anchors[:m], anchors_3d[:m])
