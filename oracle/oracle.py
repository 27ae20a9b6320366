"""ctypes front-end of the CPU oracle (oracle/mv3d_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never from mv3d_tf_amd/ (the product).  The heavy
arithmetic lives in the C restatement; this file only marshals numpy arrays and
restates the reference's random-subsampling glue (numpy global RNG, draw for draw).

Function names / argument order mirror the reference callables:
  proposal_layer_3d        lib/rpn_msr/proposal_layer_tf.py:25
  anchor_target_layer      lib/rpn_msr/anchor_target_layer_tf.py:21
  proposal_target_layer_3d lib/rpn_msr/proposal_target_layer_tf.py:19
  cpu_nms / nms            lib/nms/cpu_nms.pyx:17, lib/utils/nms.pyx:17
  bbox_overlaps            lib/utils/bbox.pyx:15
  roi_pool / roi_pool_grad lib/roi_pooling_layer/roi_pooling_op.cc:74,319
"""
import ctypes as C
import os
import subprocess

import numpy as np
import numpy.random as npr

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmv3d_oracle.so")


def build(force=False):
    """Compile the C restatement (gcc, see Makefile)."""
    src = os.path.join(_HERE, "mv3d_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.mv3d_ref_expf.restype = C.c_float
        _lib.mv3d_ref_expf.argtypes = [C.c_float]
        _lib.mv3d_ref_log.restype = C.c_double
        _lib.mv3d_ref_log.argtypes = [C.c_double]
        _lib.mv3d_ref_floor_divide.restype = C.c_double
        _lib.mv3d_ref_floor_divide.argtypes = [C.c_double, C.c_double]
    return _lib


class _ProposalCfg(C.Structure):
    _fields_ = [("feat_stride", C.c_int), ("pre_nms_topN", C.c_int), ("post_nms_topN", C.c_int),
                ("nms_thresh", C.c_double), ("min_size", C.c_double)]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# reference defaults: lib/fast_rcnn/config.py:126-148,184-192
CFG = {
    "TRAIN": dict(RPN_PRE_NMS_TOP_N=12000, RPN_POST_NMS_TOP_N=2000, RPN_NMS_THRESH=0.7, RPN_MIN_SIZE=5),
    "TEST": dict(RPN_PRE_NMS_TOP_N=12000, RPN_POST_NMS_TOP_N=2000, RPN_NMS_THRESH=0.7, RPN_MIN_SIZE=5),
}
TRAIN = dict(RPN_CLOBBER_POSITIVES=False, RPN_NEGATIVE_OVERLAP=0.5, RPN_POSITIVE_OVERLAP=0.7,
             RPN_FG_FRACTION=0.25, RPN_BATCHSIZE=128, BATCH_SIZE=128, FG_FRACTION=0.25,
             FG_THRESH=0.5, BG_THRESH_HI=0.5, BG_THRESH_LO=0.1)


def generate_anchors_bv():
    out = np.zeros((4, 4), np.int64)
    lib().mv3d_ref_generate_anchors_bv(_p(out))
    return out


def bbox_overlaps(boxes, query_boxes):
    b = np.ascontiguousarray(boxes, np.float64)
    q = np.ascontiguousarray(query_boxes, np.float64)
    out = np.zeros((b.shape[0], q.shape[0]), np.float64)
    lib().mv3d_ref_bbox_overlaps(_p(b), C.c_int(b.shape[0]), _p(q), C.c_int(q.shape[0]), _p(out))
    return out


def cpu_nms(dets, thresh, presorted=False):
    d = _f32(dets)
    n = d.shape[0]
    if n == 0:
        return []
    keep = np.zeros(n, np.int32)
    nk = lib().mv3d_ref_cpu_nms(_p(d), C.c_int(n), C.c_double(float(thresh)),
                                C.c_int(1 if presorted else 0), _p(keep))
    if nk < 0:
        raise ZeroDivisionError("float division")
    return [int(i) for i in keep[:nk]]


nms = cpu_nms


def gpu_nms_rule(sorted_dets, thresh):
    """lib/nms/nms_kernel.cu rule on pre-sorted boxes (IoU > thresh in f32): kept positions.  Parity unpinned."""
    d = _f32(sorted_dets)
    keep = np.zeros(max(d.shape[0], 1), np.int32)
    nk = lib().mv3d_ref_gpu_nms_rule(_p(d), C.c_int(d.shape[0]), C.c_float(thresh), _p(keep))
    return keep[:nk].tolist()


def proposal_layer_3d(rpn_cls_prob_reshape, rpn_bbox_pred, im_info, calib, cfg_key,
                      _feat_stride=(8,), anchor_scales=(1.0, 1.0), cfg=None, debug=False):
    cfg = (cfg or CFG)[cfg_key]
    prob, pred = _f32(rpn_cls_prob_reshape), _f32(rpn_bbox_pred)
    assert prob.shape[0] == 1, 'Only single item batches are supported'
    H, W = prob.shape[1:3]
    N = H * W * 4
    info = _f32(im_info).reshape(-1)[:3].copy()
    cal = _f32(calib).reshape(48)
    c = _ProposalCfg(int(_feat_stride[0]), int(cfg["RPN_PRE_NMS_TOP_N"]), int(cfg["RPN_POST_NMS_TOP_N"]),
                     float(cfg["RPN_NMS_THRESH"]), float(cfg["RPN_MIN_SIZE"]))
    cap = c.post_nms_topN if c.post_nms_topN > 0 else N
    bv = np.zeros((cap, 5), np.float32)
    img = np.zeros((cap, 5), np.float32)
    b3 = np.zeros((cap, 7), np.float32)
    n_out = C.c_int(0)
    dbg = {}
    if debug:
        dbg = dict(anchors3d=np.zeros((N, 6), np.float64), props3d=np.zeros((N, 6), np.float32),
                   bv_raw=np.zeros((N, 4), np.float32), corners=np.zeros((N, 24), np.float32),
                   img=np.zeros((N, 4), np.int32), valid=np.zeros(N, np.uint8),
                   order=np.zeros(N, np.int32), nms_keep=np.zeros(N, np.int32))
    n_order = C.c_int(0)
    rc = lib().mv3d_ref_proposal_layer_3d(
        _p(prob), _p(pred), C.c_int(H), C.c_int(W), _p(info), _p(cal), C.byref(c),
        _p(bv), _p(img), _p(b3), C.byref(n_out),
        _p(dbg.get("anchors3d")), _p(dbg.get("props3d")), _p(dbg.get("bv_raw")),
        _p(dbg.get("corners")), _p(dbg.get("img")), _p(dbg.get("valid")),
        _p(dbg.get("order")), C.byref(n_order), _p(dbg.get("nms_keep")))
    if rc != 0:
        raise ZeroDivisionError("float division")
    r = n_out.value
    if debug:
        dbg["order"] = dbg["order"][:n_order.value]
        dbg["nms_keep"] = dbg["nms_keep"][:r]
        return bv[:r], img[:r], b3[:r], dbg
    return bv[:r], img[:r], b3[:r]


def lidar_3d_to_corners(boxes3d):
    b = _f32(boxes3d)
    out = np.zeros((b.shape[0], 24), np.float32)
    lib().mv3d_ref_lidar_3d_to_corners(_p(b), C.c_int(b.shape[0]), _p(out))
    return out


def lidar_cnr_to_img(corners, calib):
    c = _f32(corners)
    out = np.zeros((c.shape[0], 4), np.int32)
    lib().mv3d_ref_lidar_cnr_to_img(_p(c), C.c_int(c.shape[0]), _p(_f32(calib).reshape(48)), _p(out))
    return out


def _choice_without_replacement(inds, size):
    """legacy RandomState.choice(x, size=k, replace=False) == x[permutation(len(x))[:k]]"""
    return inds[npr.permutation(len(inds))[:size]]


def anchor_target_stage1(H, W, gt_boxes, gt_boxes_3d, im_info, feat_stride=8, train=None):
    t = train or TRAIN
    gt_bv, gt_3d = _f32(gt_boxes), _f32(gt_boxes_3d)
    G = gt_bv.shape[0]
    N = H * W * 4
    inds = np.zeros(N, np.int32)
    am = np.zeros(N, np.int32)
    mo = np.zeros(N, np.float64)
    lab = np.zeros(N, np.float32)
    tg = np.zeros((N, 6), np.float32)
    info = _f32(im_info).reshape(-1)[:3].copy()
    ni = lib().mv3d_ref_anchor_target_stage1(
        C.c_int(H), C.c_int(W), C.c_int(feat_stride), _p(info), _p(gt_bv), _p(gt_3d), C.c_int(G),
        C.c_double(t["RPN_NEGATIVE_OVERLAP"]), C.c_double(t["RPN_POSITIVE_OVERLAP"]),
        C.c_int(1 if t["RPN_CLOBBER_POSITIVES"] else 0), _p(inds), _p(am), _p(mo), _p(lab), _p(tg))
    return inds[:ni], am[:ni], mo[:ni], lab[:ni], tg[:ni]


def anchor_target_layer(rpn_cls_score, gt_boxes, gt_boxes_3d, im_info, _feat_stride=(8,),
                        anchor_scales=(1.0, 1.0), train=None, debug=False):
    """anchor_target_layer_tf.py:21-250; random draws from the numpy global RNG in the
    reference's order (fg choice, bg choice, second bg choice)."""
    t = train or TRAIN
    assert rpn_cls_score.shape[0] == 1, 'Only single item batches are supported'
    H, W = rpn_cls_score.shape[1:3]
    stride = int(_feat_stride[0])
    N = H * W * 4
    inds, argmax, max_ov, labels, targets = anchor_target_stage1(H, W, gt_boxes, gt_boxes_3d, im_info,
                                                                 stride, t)
    labels = labels.copy()
    dbg = dict(inds_inside=inds.copy(), argmax_overlaps=argmax.copy(), max_overlaps=max_ov.copy(),
               labels_stage1=labels.copy())
    # :146-151
    num_fg = int(t["RPN_FG_FRACTION"] * t["RPN_BATCHSIZE"])
    fg_inds = np.where(labels == 1)[0]
    if len(fg_inds) > num_fg:
        labels[_choice_without_replacement(fg_inds, len(fg_inds) - num_fg)] = -1
    # :154-159
    num_bg = t["RPN_BATCHSIZE"] - np.sum(labels == 1)
    bg_inds = np.where(labels == 0)[0]
    if len(bg_inds) > num_bg:
        labels[_choice_without_replacement(bg_inds, len(bg_inds) - num_bg)] = -1
    # :170-174 debug outputs are taken here
    base = generate_anchors_bv()
    sel = np.where(labels != -1)[0]
    n_sel = inds[sel]
    a, cell = n_sel % 4, n_sel // 4
    shifts = np.stack([cell % W, cell // W, cell % W, cell // W], 1).astype(np.int64) * stride
    anc = base[a] + shifts
    anc3d = np.zeros((len(sel), 6), np.float64)
    if len(sel):
        anc_c = np.ascontiguousarray(anc, np.int64)
        lib().mv3d_ref_bv_anchor_to_lidar(_p(anc_c), C.c_int(len(sel)), _p(anc3d))
    zeros = np.zeros((len(sel), 1), np.float32)
    anchors = np.hstack((zeros, anc)).astype(np.float32)
    anchors_3d = np.hstack((zeros, anc3d)).astype(np.float32)
    # :176-183
    labels[max_ov < t["RPN_NEGATIVE_OVERLAP"]] = 0
    num_bg = t["RPN_BATCHSIZE"] - np.sum(labels == 1)
    bg_inds = np.where(labels == 0)[0]
    if len(bg_inds) > num_bg:
        labels[_choice_without_replacement(bg_inds, len(bg_inds) - num_bg)] = -1
    # :225-226 unmap
    rpn_labels = np.full((N,), -1, np.float32)
    rpn_labels[inds] = labels
    rpn_targets = np.zeros((N, 6), np.float32)
    rpn_targets[inds] = targets
    if debug:
        return rpn_labels, rpn_targets, anchors, anchors_3d, dbg
    return rpn_labels, rpn_targets, anchors, anchors_3d


def proposal_target_layer_3d(rpn_rois_bv, rpn_rois_3d, gt_boxes_bv, gt_boxes_3d, gt_boxes_corners,
                             calib, _num_classes, train=None, debug=False):
    """proposal_target_layer_tf.py:19-94 + _sample_rois_3d :227-298 (inputs as the f32
    tensors the graph feeds)."""
    t = train or TRAIN
    rois_bv_in, rois_3d_in = _f32(rpn_rois_bv), _f32(rpn_rois_3d)
    gt_bv, gt_3d, gt_cnr = _f32(gt_boxes_bv), _f32(gt_boxes_3d), _f32(gt_boxes_corners)
    G = gt_bv.shape[0]
    zeros = np.zeros((G, 1), np.float32)
    all_bv = np.vstack((rois_bv_in, np.hstack((zeros, gt_bv[:, :-1]))))
    all_3d = np.vstack((rois_3d_in, np.hstack((zeros, gt_3d[:, :-1]))))
    assert np.all(all_bv[:, 0] == 0), 'Only single item batches are supported'
    rois_per_image = t["BATCH_SIZE"] // 1
    fg_rois_per_image = np.round(t["FG_FRACTION"] * rois_per_image)
    ov = bbox_overlaps(all_bv[:, 1:5], gt_bv[:, :4])
    gt_assignment = ov.argmax(axis=1)
    max_ov = ov.max(axis=1)
    labels = gt_bv[gt_assignment, 4]
    fg_inds = np.where(max_ov >= t["FG_THRESH"])[0]
    fg_n = int(min(fg_rois_per_image, fg_inds.size))
    if fg_inds.size > 0:
        fg_inds = _choice_without_replacement(fg_inds, fg_n)
    bg_inds = np.where((max_ov < t["BG_THRESH_HI"]) & (max_ov >= t["BG_THRESH_LO"]))[0]
    bg_n = min(rois_per_image - fg_n, bg_inds.size)
    if bg_inds.size > 0:
        bg_inds = _choice_without_replacement(bg_inds, bg_n)
    keep = np.append(fg_inds, bg_inds).astype(np.int64)
    labels = labels[keep].copy()
    labels[fg_n:] = 0
    rois_bv = all_bv[keep]
    rois_3d = all_3d[keep]
    cnr = lidar_3d_to_corners(rois_3d[:, 1:7])
    tg = np.zeros((len(keep), 24), np.float32)
    gsel = np.ascontiguousarray(gt_cnr[gt_assignment[keep], :24])
    if len(keep):
        lib().mv3d_ref_bbox_transform_cnr(_p(cnr), _p(gsel), C.c_int(len(keep)), _p(tg))
    # _get_bbox_regression_labels_3d :172-194
    clss = labels.astype(np.uint16)
    bbox_targets = np.zeros((len(keep), 24 * _num_classes), np.float32)
    for i in np.where(clss > 0)[0]:
        bbox_targets[i, 24 * clss[i]:24 * clss[i] + 24] = tg[i]
    img = lidar_cnr_to_img(cnr, calib)
    rois_img = np.hstack((rois_bv[:, 0].reshape(-1, 1), img.astype(np.float64))).astype(np.float32)
    out = (rois_bv.reshape(-1, 5).astype(np.float32), rois_img.reshape(-1, 5),
           labels.reshape(-1, 1).astype(np.int32), bbox_targets, rois_3d.reshape(-1, 7).astype(np.float32))
    if debug:
        return out + (dict(gt_assignment=gt_assignment, max_overlaps=max_ov, keep=keep, fg_n=fg_n),)
    return out


def roi_pool(bottom_data, bottom_rois, pooled_height, pooled_width, spatial_scale):
    d, r = _f32(bottom_data), _f32(bottom_rois)
    B, H, W, Cc = d.shape
    R = r.shape[0]
    top = np.zeros((R, pooled_height, pooled_width, Cc), np.float32)
    am = np.zeros((R, pooled_height, pooled_width, Cc), np.int32)
    rc = lib().mv3d_ref_roi_pool_forward(_p(d), C.c_int(B), C.c_int(H), C.c_int(W), C.c_int(Cc), _p(r),
                                         C.c_int(R), C.c_int(pooled_height), C.c_int(pooled_width),
                                         C.c_float(spatial_scale), _p(top), _p(am))
    if rc != 0:
        raise ValueError("roi batch index out of range")
    return top, am


def roi_pool_grad(bottom_data, bottom_rois, argmax, grad, pooled_height, pooled_width, spatial_scale):
    d, r = _f32(bottom_data), _f32(bottom_rois)
    B, H, W, Cc = d.shape
    R = r.shape[0]
    g = _f32(grad)
    am = np.ascontiguousarray(argmax, np.int32)
    out = np.zeros((B, H, W, Cc), np.float32)
    lib().mv3d_ref_roi_pool_backward(_p(g), _p(am), C.c_int(B), C.c_int(H), C.c_int(W), C.c_int(Cc), _p(r),
                                     C.c_int(R), C.c_int(pooled_height), C.c_int(pooled_width),
                                     C.c_float(spatial_scale), _p(out))
    return out


def expf(x):
    return np.array([lib().mv3d_ref_expf(float(v)) for v in np.asarray(x, np.float32).ravel()],
                    np.float32).reshape(np.shape(x))


def floor_divide(a, b):
    return np.array([lib().mv3d_ref_floor_divide(float(u), float(b)) for u in np.asarray(a, np.float64).ravel()],
                    np.float64).reshape(np.shape(a))


def log(x):
    return np.array([lib().mv3d_ref_log(float(v)) for v in np.asarray(x, np.float64).ravel()],
                    np.float64).reshape(np.shape(x))


def box_tail(rois_3d, deltas, nc=2):
    """(corners (R,24), pred_cnr (R,24*nc), pred_cnr_r, pred_bv (R,4*nc) f64, pred_bv_r) of
    lib/fast_rcnn/test_mv.py:240-261."""
    r3, dl = _f32(rois_3d), _f32(deltas)
    R = r3.shape[0]
    cnr = np.zeros((R, 24), np.float32); pr = np.zeros((R, 24 * nc), np.float32)
    bv = np.zeros((R, 4 * nc), np.float32); bvr = np.zeros((R, 4 * nc), np.float32)
    lib().mv3d_ref_box_tail(_p(r3), _p(dl), C.c_int(R), C.c_int(nc), _p(cnr), _p(pr), _p(bv), _p(bvr))
    return cnr, np.hstack([cnr] * nc), pr, bv.astype(np.float64), bvr.astype(np.float64)


def point_cloud_2_top(points, res=0.1, zres=0.3, side_range=(-30., 30.), fwd_range=(0., 60.), height_range=(-2., 0.4)):
    """lib/utils/read_lidar.py:10-115 (defaults: the call MV3D makes); raises IndexError where numpy's fancy assignment would"""
    p = _f32(points)
    d = [C.c_double(float(v)) for v in (res, zres, side_range[0], side_range[1], fwd_range[0], fwd_range[1], height_range[0], height_range[1])]
    dims = (C.c_int * 3)()
    err = C.c_int(0)
    fn = lib().mv3d_ref_point_cloud_2_top_ranges
    fn.restype = None
    fn(_p(p), C.c_int(p.shape[0]), *d, dims, None, C.byref(err))
    top = np.zeros(tuple(dims), np.float32)
    fn(_p(p), C.c_int(p.shape[0]), *d, dims, _p(top), C.byref(err))
    if err.value:
        raise IndexError("point_cloud_2_top: a point's cell lies outside the map (numpy raises IndexError there)")
    return top


def test_net_frame(scores, boxes_bv, boxes_cnr, boxes_cnr_r, num_classes, nms_thresh, max_per_image=300):
    """lib/fast_rcnn/test_mv.py:420-444, 483-499 for one frame: per-class score cut (thresh is re-set to 0.05 at :421),
    cpu_nms on the (N,5) BEV dets, cap over all classes.  Returns (dets[cls], dets_cnr[cls]) with index 0 empty."""
    thresh = 0.05
    dets, dets_cnr = [[]], [[]]
    for j in range(1, num_classes):
        inds = np.where(scores[:, j] > thresh)[0]
        cls_scores = scores[inds, j]
        cls_dets = np.hstack((boxes_bv[inds, j * 4:(j + 1) * 4], cls_scores[:, np.newaxis])).astype(np.float32, copy=False)
        cls_dets_cnr = np.hstack((boxes_cnr[inds, j * 24:(j + 1) * 24], cls_scores[:, np.newaxis])).astype(np.float32, copy=False)
        keep = cpu_nms(cls_dets, nms_thresh)
        dets.append(cls_dets[keep, :]); dets_cnr.append(cls_dets_cnr[keep, :])
    if max_per_image > 0:
        image_scores = np.hstack([dets[j][:, -1] for j in range(1, num_classes)])
        if len(image_scores) > max_per_image:
            image_thresh = np.sort(image_scores)[-max_per_image]
            for j in range(1, num_classes):
                keep = np.where(dets[j][:, -1] >= image_thresh)[0]
                dets[j] = dets[j][keep, :]
                dets_cnr[j] = dets_cnr[j][keep, :]
    return dets, dets_cnr


def detection_losses(cls_score, labels, bbox_pred, bbox_targets, rpn, sigma=3.0):
    """lib/fast_rcnn/train_mv.py:74-127 in numpy f32 (parity unpinned: the reference's losses are TensorFlow ops and
    TensorFlow is not available here; formula restated).  rpn: rows label != -1 for the cross-entropy, label == 1 for
    the box loss; otherwise every row.  Returns (cross_entropy, loss_box, d_cls, d_pred)."""
    z = _f32(cls_score); p = _f32(bbox_pred); t = _f32(bbox_targets)
    lab = np.asarray(labels).reshape(-1)
    keep = (lab != -1) if rpn else np.ones(lab.shape, bool)
    pos = (lab == 1) if rpn else np.ones(lab.shape, bool)
    sigma2 = np.float32(sigma * sigma)
    m = z.max(1, keepdims=True)
    e = np.exp(z - m)
    se = e.sum(1, keepdims=True)
    li = lab.astype(np.int64).clip(0, z.shape[1] - 1)
    ce_rows = np.log(se[:, 0]) - (z[np.arange(len(li)), li] - m[:, 0])
    with np.errstate(all="ignore"):
        ce = np.float32(ce_rows[keep].astype(np.float64).sum() / keep.sum())
        d = p - t
        quad = np.abs(d) < np.float32(1.0) / sigma2
        l1 = np.where(quad, (d * d) * (np.float32(0.5) * sigma2), np.abs(d) - np.float32(0.5) / sigma2)
        box = np.float32(l1.sum(1)[pos].astype(np.float64).sum() / pos.sum())
        onehot = np.zeros_like(z); onehot[np.arange(len(li)), li] = 1
        d_cls = np.where(keep[:, None], (e / se - onehot) * np.float32(1.0 / keep.sum()), 0).astype(np.float32)
        d_pred = np.where(pos[:, None], np.where(quad, sigma2 * d, np.sign(d)) * np.float32(1.0 / pos.sum()), 0).astype(np.float32)
    return ce, box, d_cls, d_pred



# ------------------------------------------------------------------ X1: front-view ROI (parity unpinned: no reference code)
def rois_3d_to_fv(rois_3d):
    """rois_3d (R,7) [b,x,y,z,l,w,h] -> rois_fv (R,5) [b,x1,y1,x2,y2] on the 64 x 512 cylindrical front-view map
    (definition: mv3d_tf_amd/csrc/front_view.hip; the reference leaves this view a TODO, network.py:293-315)."""
    r3 = _f32(rois_3d).reshape(-1, 7)
    out = np.zeros((r3.shape[0], 5), np.float32)
    lib().mv3d_ref_rois_3d_to_fv(_p(r3), C.c_int(r3.shape[0]), _p(out))
    return out


def fv_atan2(y, x):
    f = lib().mv3d_ref_fv_atan2
    f.restype = C.c_double
    f.argtypes = [C.c_double, C.c_double]
    return np.array([f(float(a), float(b)) for a, b in zip(np.ravel(y), np.ravel(x))])


# ------------------------------------------------------------------ f3: KITTI label rows -> ground-truth encodings
def gt_encode(box_cam, ry, Tr):
    """box_cam (G,6) f32 [tx,ty,tz,l,w,h], ry (G,) Python floats, Tr (3,4) f32 -> (boxes3D_cam_corners (G,24),
    boxes_corners (G,24), boxes_3D (G,6), boxes_bv (G,4)) f32, as lib/datasets/kitti_mv3d.py:240-272 fills them."""
    b = _f32(box_cam).reshape(-1, 6)
    G = b.shape[0]
    ry = np.asarray(ry, np.float64).reshape(-1)
    cs = np.ascontiguousarray(np.stack([np.cos(ry), np.sin(ry)], 1), np.float64)
    tr = _f32(Tr).reshape(3, 4)
    inv = np.ascontiguousarray(np.linalg.inv(tr[:, :3]), np.float32)
    out = [np.zeros((G, 24), np.float32), np.zeros((G, 24), np.float32), np.zeros((G, 6), np.float32), np.zeros((G, 4), np.float32)]
    lib().mv3d_ref_gt_encode(_p(b), _p(cs), C.c_int(G), _p(inv), _p(tr), *(_p(o) for o in out))
    return tuple(out)
