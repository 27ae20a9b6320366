"""`box_detect(sess, net, im, bv, calib, boxes=None)`: interface of
lib/fast_rcnn/test_mv.py:149-264.  `sess` is an opaque context (ignored: there is no TF
session).  Returns (scores (R,K), pred_boxes_bv (R,4K) f64, pred_boxes_cnr (R,24K) f32,
pred_boxes_cnr_r (R,24K) f32) like the reference, K = 2; the geometric tail runs in
libmv3d_hip.so (mv3d_box_detect_tail)."""
import numpy as np
import torch

from .. import ops
from .config import cfg
from ..networks.mv3d import n_classes


def box_detect(sess, net, im, bv, calib, boxes=None):
    im_blob = (np.asarray(im, np.float64) - cfg.PIXEL_MEANS).astype(np.float32)      # :162
    im_blob = im_blob.reshape((1,) + im_blob.shape)
    bv_blob = np.asarray(bv, np.float32).reshape((1,) + tuple(np.shape(bv)))
    im_info = np.array([[bv_blob.shape[1], bv_blob.shape[2], 1]], dtype=np.float32)   # :175-177 (BEV size, scale 1)
    with torch.no_grad():
        L = net.forward({"image_data": im_blob, "lidar_bv_data": bv_blob, "im_info": im_info, "calib": calib,
                         "keep_prob": 1.0})
        rois = L["rois"]
        scores = L["cls_prob"]
        # :240-261 -- the corner regression is deliberately NOT applied to pred_boxes_cnr (sic)
        cnr, pred_r, pred_bv, _ = ops.box_detect_tail(rois[2].contiguous(), L["bbox_pred"].contiguous(), n_classes)
        pred_cnr = torch.cat([cnr] * n_classes, dim=1)
    return (scores.cpu().numpy(), pred_bv.cpu().numpy().astype(np.float64), pred_cnr.cpu().numpy(), pred_r.cpu().numpy())
