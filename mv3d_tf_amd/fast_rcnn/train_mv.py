"""The loss side of lib/fast_rcnn/train_mv.py: `modified_smooth_l1` (:74-90), the four
losses of `train_model` (:92-130) as autograd functions over the fused device kernels (csrc/losses.hip), the
snapshot file name (:48-65) and the `.npy` weight-dict format that `network.load` reads (network.py:45-64).
The optimiser / data-layer loop of `train_model` is out of scope."""
import os

import numpy as np
import torch

from .. import ops
from .config import cfg


class _RpnLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cls_score, labels, bbox_pred, bbox_targets, sigma):
        losses, d_cls, d_pred = ops.rpn_loss(cls_score.contiguous(), labels.contiguous(), bbox_pred.contiguous(),
                                             bbox_targets.contiguous(), sigma)
        ctx.save_for_backward(d_cls, d_pred)
        return losses[0], losses[1]

    @staticmethod
    def backward(ctx, g_ce, g_box):
        d_cls, d_pred = ctx.saved_tensors
        return d_cls * g_ce, None, d_pred * g_box, None, None


class _RcnnLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cls_score, labels, bbox_pred, bbox_targets, sigma):
        losses, d_cls, d_pred = ops.rcnn_loss(cls_score.contiguous(), labels.contiguous(), bbox_pred.contiguous(),
                                              bbox_targets.contiguous(), sigma)
        ctx.save_for_backward(d_cls, d_pred)
        return losses[0], losses[1]

    @staticmethod
    def backward(ctx, g_ce, g_box):
        d_cls, d_pred = ctx.saved_tensors
        return d_cls * g_ce, None, d_pred * g_box, None, None


def modified_smooth_l1(sigma, bbox_pred, bbox_targets):
    """Element-wise form of train_mv.py:74-90 (torch expression of the same formula; the fused kernels are used by
    the losses below)."""
    sigma2 = sigma * sigma
    diffs = bbox_pred - bbox_targets
    sign = (diffs.abs() < 1.0 / sigma2).to(diffs.dtype)
    return (diffs * diffs) * (0.5 * sigma2) * sign + (diffs.abs() - 0.5 / sigma2) * (sign - 1.0).abs()


def rpn_losses(rpn_cls_score_reshape, rpn_data, rpn_bbox_pred, sigma=3.0):
    """train_mv.py:92-113: (rpn_cross_entropy, rpn_loss_box) from the `rpn_cls_score_reshape` output (..., 2), the
    `rpn_data` tuple (labels, bbox_targets, ...) of anchor_target_layer and `rpn_bbox_pred`."""
    cls = rpn_cls_score_reshape.reshape(-1, 2)
    labels = torch.as_tensor(rpn_data[0], dtype=torch.float32, device=cls.device).reshape(-1)
    targets = torch.as_tensor(rpn_data[1], dtype=torch.float32, device=cls.device).reshape(-1, 6)
    return _RpnLoss.apply(cls, labels, rpn_bbox_pred.reshape(-1, 6), targets, float(sigma))


def rcnn_losses(cls_score, roi_data_3d, bbox_pred, sigma=3.0):
    """train_mv.py:115-127: (cross_entropy, loss_box) from `cls_score`, the `roi_data_3d` tuple (.., labels, targets, ..)
    of proposal_target_layer_3d and `bbox_pred`."""
    labels = torch.as_tensor(roi_data_3d[2], device=cls_score.device).reshape(-1).to(torch.int32)
    targets = torch.as_tensor(roi_data_3d[3], dtype=torch.float32, device=cls_score.device)
    return _RcnnLoss.apply(cls_score, labels, bbox_pred, targets, float(sigma))


def total_loss(net_layers, sigma=3.0):
    """train_mv.py:130: cross_entropy + loss_box + rpn_cross_entropy + rpn_loss_box from the train graph's layers."""
    rpn_ce, rpn_box = rpn_losses(net_layers['rpn_cls_score_reshape'], net_layers['rpn_data'], net_layers['rpn_bbox_pred'], sigma)
    ce, box = rcnn_losses(net_layers['cls_score'], net_layers['roi_data_3d'], net_layers['bbox_pred'], sigma)
    return ce + box + rpn_ce + rpn_box, (ce, box, rpn_ce, rpn_box)


def snapshot_filename(output_dir, iter):
    """train_mv.py:56-61: <output_dir>/<SNAPSHOT_PREFIX>[_<SNAPSHOT_INFIX>]_iter_<iter+1>.ckpt"""
    infix = ('_' + cfg.TRAIN.SNAPSHOT_INFIX if cfg.TRAIN.SNAPSHOT_INFIX != '' else '')
    return os.path.join(output_dir, cfg.TRAIN.SNAPSHOT_PREFIX + infix + '_iter_{:d}'.format(iter + 1) + '.ckpt')


def save_weights_npy(net, path):
    """The `.npy` dict {layer: {'weights': ..., 'biases': ...}} that network.load() reads (network.py:45-64; the
    reference writes the same structure at test_mv.py:345-372)."""
    d = {name: {'weights': w.detach().cpu().numpy(), 'biases': b.detach().cpu().numpy()} for name, (w, b) in net.params.items()}
    np.save(path, d, allow_pickle=True)
    return path
