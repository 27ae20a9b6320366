"""Tensor-level calls into libmv3d_hip.so.

torch CUDA(ROCm) tensors are used only as device buffers: every function passes
`tensor.data_ptr()` and the current stream handle through the C-ABI and returns tensors
holding the results.  Nothing here synchronises with the host unless it says so.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import AnchorTargetParams, ProposalParams, ProposalTargetParams, RoiGradView, RoiView, check, lib


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _dev(t, dtype=torch.float32, device=None):
    """numpy / tensor -> contiguous device tensor of dtype (the placeholder cast of the graph)."""
    if isinstance(t, torch.Tensor):
        return t.to(device=device or (t.device if t.is_cuda else "cuda"), dtype=dtype).contiguous()
    return torch.as_tensor(np.ascontiguousarray(t), dtype=dtype).to(device or "cuda")


_ws_cache = {}


def upload_packed(arrays, device):
    """Several small host arrays (f32 / i32, any shape) -> ONE host-to-device copy; returns device views with the
    arrays' shapes (4-byte slots; an int32 array comes back as an int32 view)."""
    flat = [np.ascontiguousarray(a, np.int32 if np.issubdtype(np.asarray(a).dtype, np.integer) else np.float32) for a in arrays]
    sizes = [a.size for a in flat]
    host = np.empty((max(sum(sizes), 1),), np.float32)
    o = 0
    for a, n in zip(flat, sizes):
        host[o:o + n] = a.reshape(-1).view(np.float32)
        o += n
    devbuf = torch.from_numpy(host).to(device)
    views, o = [], 0
    for a, n in zip(flat, sizes):
        v = devbuf[o:o + n]
        if a.dtype == np.int32:
            v = v.view(torch.int32)
        views.append(v.view(a.shape))
        o += n
    return views


def packed_views(spec, device):
    """spec: [(shape, torch dtype of 4 bytes)] -> (pack f32 buffer, views): outputs that a host consumer fetches with
    ONE device-to-host copy (`unpack_host`)."""
    sizes = [int(np.prod(sh)) for sh, _ in spec]
    pack = torch.empty((max(sum(sizes), 1),), dtype=torch.float32, device=device)
    views, o = [], 0
    for (sh, dt), n in zip(spec, sizes):
        v = pack[o:o + n]
        if dt != torch.float32:
            v = v.view(dt)
        views.append(v.view(sh))
        o += n
    return pack, views


def unpack_host(pack, spec):
    host = pack.cpu().numpy()
    outs, o = [], 0
    for sh, dt in spec:
        n = int(np.prod(sh))
        a = host[o:o + n]
        if dt != torch.float32:
            a = a.view(np.int32)
        outs.append(a.reshape(sh))
        o += n
    return outs


def _workspace(nbytes, device, tag, zero=False):
    """Reused per (device, stream, tag) so hot calls never allocate (buffers are 256-B aligned).  zero=True: zero-filled
    when it is (re)allocated (workspaces whose header the kernels expect zero on first use and leave zero)."""
    key = (str(device), torch.cuda.current_stream().cuda_stream, tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = (torch.zeros if zero else torch.empty)(max(nbytes, 256), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


# ------------------------------------------------------------------ proposal_layer_3d
def proposal_params(cfg_section, feat_stride=8, img_height=375, img_width=1242, img_padding=50, use_gpu_nms=None):
    """cfg[cfg_key] section (lib/rpn_msr/proposal_layer_tf.py:52-55) -> mv3d_proposal_params.  `use_gpu_nms` (default:
    cfg.USE_GPU_NMS) selects the rule of the NMS inside the layer the way nms_wrapper.py:13-21 does."""
    if use_gpu_nms is None:
        from .fast_rcnn.config import cfg
        use_gpu_nms = bool(cfg.USE_GPU_NMS)
    return ProposalParams(int(feat_stride), int(cfg_section["RPN_PRE_NMS_TOP_N"]),
                          int(cfg_section["RPN_POST_NMS_TOP_N"]), int(img_height), int(img_width),
                          int(img_padding), 1 if use_gpu_nms else 0, 0, float(cfg_section["RPN_NMS_THRESH"]),
                          float(cfg_section["RPN_MIN_SIZE"]))


def proposal_3d_outputs(B, cap, dev):
    """The five outputs of proposal_3d as views of ONE buffer (bv | img | 3d | num | status, 4-byte slots), so that
    a host consumer fetches them with a single device-to-host copy.  Returns (pack, (bv, img, b3, num, status))."""
    n5, n7 = B * cap * 5, B * cap * 7
    pack = torch.empty((2 * n5 + n7 + 2 * B,), dtype=torch.float32, device=dev)
    bv = pack[:n5].view(B, cap, 5)
    img = pack[n5:2 * n5].view(B, cap, 5)
    b3 = pack[2 * n5:2 * n5 + n7].view(B, cap, 7)
    tail = pack[2 * n5 + n7:].view(torch.int32)
    return pack, (bv, img, b3, tail[:B], tail[B:])


def proposal_3d(prob, pred, im_info, calib, params, out=None):
    """prob (B,H,W,8), pred (B,H,W,24), im_info (B,3), calib (B,4,12) device f32 tensors.
    Returns (blob_bv (B,cap,5), blob_img (B,cap,5), blob_3d (B,cap,7), num_out (B) i32,
    status (B) i32); rows >= num_out[b] are zero.  Asynchronous."""
    B, H, W, _ = prob.shape
    dev = prob.device
    cap = lib().mv3d_proposal_3d_capacity(H, W, C.byref(params))
    if cap < 0:
        check(_lib.ERR_INVALID_ARG, "mv3d_proposal_3d_capacity")
    nbytes = lib().mv3d_proposal_3d_workspace_bytes(B, H, W, C.byref(params))
    ws = _workspace(nbytes, dev, "proposal")
    if out is None:
        out = proposal_3d_outputs(B, cap, dev)[1]
    bv, img, b3, num, status = out
    rc = lib().mv3d_proposal_3d(_ptr(prob), _ptr(pred), B, H, W, _ptr(im_info), _ptr(calib), C.byref(params),
                                _ptr(bv), _ptr(img), _ptr(b3), _ptr(num), _ptr(status), _ptr(ws), ws.numel(),
                                _stream())
    check(rc, "mv3d_proposal_3d")
    return out


# ------------------------------------------------------------------ NMS
def nms_device(dets, thresh, max_keep=0):
    """dets (n,5) device f32 in processing order -> (keep (n) i32, num_keep (1) i32, status (1) i32)."""
    n = dets.shape[0]
    dev = dets.device
    keep = torch.empty((max(n, 1),), dtype=torch.int32, device=dev)
    cnt = torch.zeros((2,), dtype=torch.int32, device=dev)
    ws = _workspace(lib().mv3d_nms_workspace_bytes(n), dev, "nms")
    rc = lib().mv3d_nms_device(_ptr(dets), n, float(thresh), int(max_keep), _ptr(keep), _ptr(cnt[0:1]),
                               _ptr(cnt[1:2]), _ptr(ws), ws.numel(), _stream())
    check(rc, "mv3d_nms_device")
    return keep, cnt[0:1], cnt[1:2]


def nms_host(dets, thresh, device_id=0):
    """numpy (n,5) f32, unsorted -> list of kept indices (cpu_nms semantics, computed on the GPU)."""
    d = np.ascontiguousarray(dets, dtype=np.float32)
    n = d.shape[0]
    keep = np.zeros(max(n, 1), np.int32)
    num = C.c_int32(0)
    rc = lib().mv3d_nms_host(keep.ctypes.data_as(C.c_void_p), C.byref(num), d.ctypes.data_as(C.c_void_p), n,
                             float(thresh), int(device_id))
    check(rc, "mv3d_nms_host")
    return [int(i) for i in keep[:num.value]]


def nms_gpu_rule_host(sorted_dets, thresh, device_id=0):
    """_nms of lib/nms/gpu_nms.hpp: host pointers, pre-sorted boxes, IoU > thresh in f32."""
    d = np.ascontiguousarray(sorted_dets, dtype=np.float32)
    n = d.shape[0]
    keep = np.zeros(max(n, 1), np.int32)
    num = C.c_int(0)
    lib()._nms(keep.ctypes.data_as(C.c_void_p), C.byref(num), d.ctypes.data_as(C.c_void_p), n, d.shape[1],
               C.c_float(thresh), int(device_id))
    return keep[:num.value]


# ------------------------------------------------------------------ RoiPool
def roi_pool_forward(data, rois, pooled_height, pooled_width, spatial_scale, want_argmax=True):
    B, H, W, Cc = data.shape
    R = rois.shape[0]
    top = torch.empty((R, pooled_height, pooled_width, Cc), dtype=torch.float32, device=data.device)
    argmax = torch.empty((R, pooled_height, pooled_width, Cc), dtype=torch.int32, device=data.device) if want_argmax else None
    rc = lib().mv3d_roi_pool_forward(_ptr(data), C.c_float(spatial_scale), B, R, H, W, Cc, pooled_height,
                                     pooled_width, _ptr(rois), _ptr(top), _ptr(argmax), _stream())
    check(rc, "mv3d_roi_pool_forward")
    return top, argmax


def roi_pool_backward(top_diff, rois, argmax, data_shape, pooled_height, pooled_width, spatial_scale):
    B, H, W, Cc = data_shape
    R = rois.shape[0]
    out = torch.empty((B, H, W, Cc), dtype=torch.float32, device=top_diff.device)
    rc = lib().mv3d_roi_pool_backward(_ptr(top_diff), C.c_float(spatial_scale), B, R, H, W, Cc, pooled_height,
                                      pooled_width, _ptr(rois), _ptr(out), _ptr(argmax), _stream())
    check(rc, "mv3d_roi_pool_backward")
    return out


# ------------------------------------------------------------------ anchor_target_layer
def anchor_target_stage1(H, W, im_info, gt_bv, gt_3d, params, labels=None, targets=None):
    dev = gt_bv.device
    N = H * W * 4
    G = gt_bv.shape[0]
    if labels is None:
        labels = torch.empty((N,), dtype=torch.float32, device=dev)
        targets = torch.empty((N, 6), dtype=torch.float32, device=dev)
    # counts (8 x i32) and the per-foreground flags in ONE buffer: the host reads both with one copy
    cf = torch.empty((32 + N,), dtype=torch.uint8, device=dev)
    counts = cf[:32].view(torch.int32)
    fg_hi = cf[32:]
    ws = _workspace(lib().mv3d_anchor_target_workspace_bytes(H, W, G), dev, "anchor_target")
    rc = lib().mv3d_anchor_target_stage1(H, W, _ptr(im_info), _ptr(gt_bv), _ptr(gt_3d), G, C.byref(params),
                                         _ptr(labels), _ptr(targets), _ptr(counts), _ptr(fg_hi), _ptr(ws),
                                         ws.numel(), _stream())
    check(rc, "mv3d_anchor_target_stage1")
    return labels, targets, counts, fg_hi, ws, cf


def anchor_target_stage2(H, W, params, dis_fg, dis_bg1, dis_bg2, labels, ws, anchors_cap, out=None):
    dev = labels.device
    lists = [np.zeros(0, np.int32) if a is None else np.ascontiguousarray(a, np.int32) for a in (dis_fg, dis_bg1, dis_bg2)]
    t_fg = t_b1 = t_b2 = None
    if sum(len(a) for a in lists):
        v = upload_packed(lists, dev)                          # the three index lists in one upload
        t_fg, t_b1, t_b2 = (x if x.numel() else None for x in v)
    if out is None:
        anchors = torch.empty((anchors_cap, 5), dtype=torch.float32, device=dev)
        anchors_3d = torch.empty((anchors_cap, 7), dtype=torch.float32, device=dev)
        n_anchors = torch.empty((1,), dtype=torch.int32, device=dev)
    else:
        anchors, anchors_3d, n_anchors = out
    rc = lib().mv3d_anchor_target_stage2(H, W, C.byref(params), _ptr(t_fg), 0 if t_fg is None else t_fg.numel(),
                                         _ptr(t_b1), 0 if t_b1 is None else t_b1.numel(),
                                         _ptr(t_b2), 0 if t_b2 is None else t_b2.numel(),
                                         _ptr(labels), _ptr(anchors), _ptr(anchors_3d), _ptr(n_anchors),
                                         anchors_cap, _ptr(ws), ws.numel(), _stream())
    check(rc, "mv3d_anchor_target_stage2")
    return anchors, anchors_3d, n_anchors


# ------------------------------------------------------------------ proposal_target_layer_3d
def proposal_target_stage1(rois_bv, rois_3d, gt_bv, gt_3d, params):
    dev = gt_bv.device
    R, G = rois_bv.shape[0], gt_bv.shape[0]
    counts = torch.empty((4,), dtype=torch.int32, device=dev)
    ws = _workspace(lib().mv3d_proposal_target_workspace_bytes(R, G), dev, "proposal_target")
    rc = lib().mv3d_proposal_target_stage1(_ptr(rois_bv), _ptr(rois_3d), R, _ptr(gt_bv), _ptr(gt_3d), G, C.byref(params),
                                           _ptr(counts), _ptr(ws), ws.numel(), _stream())
    check(rc, "mv3d_proposal_target_stage1")
    return counts, ws


def proposal_target_spec(S, nc):
    return [((S, 5), torch.float32), ((S, 5), torch.float32), ((S, 1), torch.int32), ((S, 24 * nc), torch.float32),
            ((S, 7), torch.float32)]


def proposal_target_stage2(rois_bv, rois_3d, gt_bv, gt_3d, gt_cnr, calib, params, fg_pick, bg_pick, ws, out=None):
    dev = gt_bv.device
    R, G = rois_bv.shape[0], gt_bv.shape[0]
    t_fg = t_bg = None
    if len(fg_pick) + len(bg_pick):
        v = upload_packed([np.ascontiguousarray(fg_pick, np.int32), np.ascontiguousarray(bg_pick, np.int32)], dev)
        t_fg, t_bg = (x if x.numel() else None for x in v)
    S = len(fg_pick) + len(bg_pick)
    nc = params.num_classes
    if out is None:
        out = tuple(torch.empty(sh, dtype=dt, device=dev) for sh, dt in proposal_target_spec(S, nc))
    rc = lib().mv3d_proposal_target_stage2(_ptr(rois_bv), _ptr(rois_3d), R, _ptr(gt_bv), _ptr(gt_3d), _ptr(gt_cnr), G,
                                           _ptr(calib), C.byref(params), _ptr(t_fg), len(fg_pick), _ptr(t_bg), len(bg_pick),
                                           _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _ptr(out[3]), _ptr(out[4]),
                                           _ptr(ws), ws.numel(), _stream())
    check(rc, "mv3d_proposal_target_stage2")
    return out


# ------------------------------------------------------------------ SURVEY §8(f) next rows
def point_cloud_2_top(points, ranges=None):
    """points (P,4) f32 device tensor -> top f32 device tensor: (601,601,9) for the call MV3D makes (ranges=None), or the map of
    ranges = (res, zres, side_lo, side_hi, fwd_lo, fwd_hi, height_lo, height_hi) (lib/utils/read_lidar.py:10-16); that form reads
    one status word back and raises IndexError where numpy's fancy assignment would (a cell outside the map)."""
    pts = points.contiguous()
    if ranges is None:
        top = torch.empty((601, 601, 9), dtype=torch.float32, device=pts.device)
        check(lib().mv3d_point_cloud_2_top(_ptr(pts), pts.shape[0], _ptr(top), _stream()), "mv3d_point_cloud_2_top")
        return top
    r = [float(v) for v in ranges]
    dims = (C.c_int * 3)()
    check(lib().mv3d_point_cloud_2_top_shape(*r, dims), "mv3d_point_cloud_2_top_shape")
    top = torch.empty(tuple(dims), dtype=torch.float32, device=pts.device)
    status = torch.empty(1, dtype=torch.int32, device=pts.device)
    check(lib().mv3d_point_cloud_2_top_ranges(_ptr(pts), pts.shape[0], *r, _ptr(top), _ptr(status), _stream()), "mv3d_point_cloud_2_top_ranges")
    if int(status.item()):
        raise IndexError("point_cloud_2_top: a point's cell lies outside the map (numpy raises IndexError there)")
    return top


def box_detect_tail(rois_3d, bbox_pred, num_classes):
    R = rois_3d.shape[0]
    dev = rois_3d.device
    cnr = torch.empty((R, 24), dtype=torch.float32, device=dev)
    pr = torch.empty((R, 24 * num_classes), dtype=torch.float32, device=dev)
    bv = torch.empty((R, 4 * num_classes), dtype=torch.float32, device=dev)
    bvr = torch.empty((R, 4 * num_classes), dtype=torch.float32, device=dev)
    check(lib().mv3d_box_detect_tail(_ptr(rois_3d), _ptr(bbox_pred), R, num_classes, _ptr(cnr), _ptr(pr), _ptr(bv), _ptr(bvr),
                                     _stream()), "mv3d_box_detect_tail")
    return cnr, pr, bv, bvr


def roi_pool_forward_views(views, pooled_height, pooled_width, outs=None, cold_maps=False, want_argmax=True, top_dtype=None):
    """views: list of (data (B,H,W,C), rois (R,5), spatial_scale); one launch for all of them.
    Returns [(top, argmax), ...]; pass `outs` (same structure) to reuse output tensors.  cold_maps: the maps are not
    cache-resident (mv3d_roi_pool_forward_views_cold: same results, prefetch workgroups in front of the launch).
    want_argmax=False (inference): argmax_data = NULL, the entries are (top, None).  top_dtype = torch.float16 / bfloat16 (inference,
    want_argmax=False): `top` in that type (mv3d_roi_pool_forward_views_half: the values a cast of the f32 top gives)."""
    half = top_dtype in (torch.float16, torch.bfloat16)
    if half and want_argmax:
        raise ValueError("16-bit tops are an inference output: want_argmax=False")
    arr = (RoiView * len(views))()
    res = []
    for k, (data, rois, scale) in enumerate(views):
        B, H, W, Cc = data.shape
        R = rois.shape[0]
        if outs is not None:
            top, am = outs[k]
        else:
            top = torch.empty((R, pooled_height, pooled_width, Cc), dtype=top_dtype if half else torch.float32, device=data.device)
            am = torch.empty((R, pooled_height, pooled_width, Cc), dtype=torch.int32, device=data.device) if want_argmax else None
        arr[k] = RoiView(data.data_ptr(), rois.data_ptr(), top.data_ptr(), am.data_ptr() if am is not None else None, float(scale), B, R, H, W, Cc)
        res.append((top, am))
    if half:
        check(lib().mv3d_roi_pool_forward_views_half(len(views), arr, pooled_height, pooled_width, 1 if top_dtype == torch.float16 else 2,
                                                     1 if cold_maps else 0, _stream()), "mv3d_roi_pool_forward_views_half")
        return res
    fn = lib().mv3d_roi_pool_forward_views_cold if cold_maps else lib().mv3d_roi_pool_forward_views
    check(fn(len(views), arr, pooled_height, pooled_width, _stream()), "mv3d_roi_pool_forward_views")
    return res


def roi_pool_backward_views(views, pooled_height, pooled_width, outs=None):
    """views: list of (top_diff (R,PH,PW,C), rois (R,5), argmax (R,PH,PW,C) i32, data_shape (B,H,W,C), spatial_scale);
    RoiPoolGrad of all of them behind one call.  Returns [bottom_diff, ...]; pass `outs` to reuse output tensors."""
    arr = (RoiGradView * len(views))()
    res = []
    for k, (top_diff, rois, argmax, shape, scale) in enumerate(views):
        B, H, W, Cc = shape
        out = outs[k] if outs is not None else torch.empty((B, H, W, Cc), dtype=torch.float32, device=top_diff.device)
        arr[k] = RoiGradView(out.data_ptr(), rois.data_ptr(), top_diff.data_ptr(), argmax.data_ptr(), float(scale), B,
                             rois.shape[0], H, W, Cc)
        res.append(out)
    nbytes = lib().mv3d_roi_pool_backward_workspace_bytes(len(views), arr, pooled_height, pooled_width)
    ws = _workspace(nbytes, views[0][0].device, "roi_bwd", zero=True)
    check(lib().mv3d_roi_pool_backward_views(len(views), arr, pooled_height, pooled_width, _ptr(ws), ws.numel(), _stream()),
          "mv3d_roi_pool_backward_views")
    return res


# ---- the RoiPool pair of a training step: private compact argmax plane, index + fill and gather behind one backward call
_PAIR_WS_FREE = {}                 # (device, bytes, stream) -> workspaces whose last user was enqueued on that stream


def _stream_key():
    return int(torch.cuda.current_stream().cuda_stream)


def roi_pair_workspace(nbytes, device):
    """A workspace for mv3d_roi_pool_backward_views_pair (no initialisation needed), recycled through release_roi_pair_workspace.
    The pool is per STREAM: a released workspace is handed out again only to a call enqueued on the stream its last launches
    run on (stream order is what makes the reuse safe; another stream could start while they are still running)."""
    free = _PAIR_WS_FREE.setdefault((str(device), int(nbytes), _stream_key()), [])
    return free.pop() if free else torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def release_roi_pair_workspace(ws):
    """ws goes back to the pool of the CURRENT stream (call it on the stream the workspace's launches were enqueued on)."""
    free = _PAIR_WS_FREE.setdefault((str(ws.device), int(ws.numel()), _stream_key()), [])
    if len(free) < 4:
        free.append(ws)


def roi_pool_forward_views_pair(views, pooled_height, pooled_width, outs=None, cold_maps=False):
    """views as roi_pool_forward_views; one launch.  Returns [(top, argmax_private), ...]: the second tensor is the pair's PRIVATE
    argmax plane (one-byte codes, 16-bit for bins of > 255 pixels, in an int32-shaped buffer for the shapes the pair's kernels take) -- hand it to
    roi_pool_backward_views_pair only; roi_pool_argmax_decode gives the reference's int32 plane."""
    arr = (RoiView * len(views))()
    res = []
    for k, (data, rois, scale) in enumerate(views):
        B, H, W, Cc = data.shape
        R = rois.shape[0]
        if outs is not None:
            top, am = outs[k]
        else:
            top = torch.empty((R, pooled_height, pooled_width, Cc), dtype=torch.float32, device=data.device)
            am = torch.empty((R, pooled_height, pooled_width, Cc), dtype=torch.int32, device=data.device)
        arr[k] = RoiView(data.data_ptr(), rois.data_ptr(), top.data_ptr(), am.data_ptr(), float(scale), B, R, H, W, Cc)
        res.append((top, am))
    check(lib().mv3d_roi_pool_forward_views_pair(len(views), arr, pooled_height, pooled_width, 1 if cold_maps else 0, _stream()),
          "mv3d_roi_pool_forward_views_pair")
    return res


def roi_pool_argmax_decode(views, res, pooled_height, pooled_width):
    """The reference's int32 argmax planes of a roi_pool_forward_views_pair call: views / res as given to / returned by it.  Tests and
    verification only."""
    arr = (RoiView * len(views))()
    outs = []
    for k, ((data, rois, scale), (top, am)) in enumerate(zip(views, res)):
        B, H, W, Cc = data.shape
        arr[k] = RoiView(data.data_ptr(), rois.data_ptr(), top.data_ptr(), am.data_ptr(), float(scale), B, rois.shape[0], H, W, Cc)
        outs.append(torch.empty(tuple(top.shape), dtype=torch.int32, device=data.device))
    ptrs = (C.c_void_p * len(views))(*[o.data_ptr() for o in outs])
    check(lib().mv3d_roi_pool_argmax_decode(len(views), arr, pooled_height, pooled_width, ptrs, _stream()), "mv3d_roi_pool_argmax_decode")
    return outs


def roi_pool_backward_views_pair(views, pooled_height, pooled_width, outs=None, workspace=None):
    """views as roi_pool_backward_views, of a roi_pool_forward_views_pair call (the argmax tensors it returned): RoiPoolGrad of all of
    them behind one call.  workspace = None: a recycled workspace (index + zero fill, gather); a uint8 tensor: that workspace;
    False: NO workspace -- the entry's single-launch path (map tiles in LDS, csrc/roi_grad_tiles.hip), same results.
    Returns [bottom_diff, ...]."""
    arr = (RoiGradView * len(views))()
    res = []
    for k, (top_diff, rois, argmax, shape, scale) in enumerate(views):
        B, H, W, Cc = shape
        out = outs[k] if outs is not None else torch.empty((B, H, W, Cc), dtype=torch.float32, device=top_diff.device)
        arr[k] = RoiGradView(out.data_ptr(), rois.data_ptr(), top_diff.data_ptr(), argmax.data_ptr(), float(scale), B,
                             rois.shape[0], H, W, Cc)
        res.append(out)
    if workspace is False:
        check(lib().mv3d_roi_pool_backward_views_pair(len(views), arr, pooled_height, pooled_width, None, 0, _stream()),
              "mv3d_roi_pool_backward_views_pair")
        return res
    ws = workspace
    if ws is None:
        ws = roi_pair_workspace(lib().mv3d_roi_pool_pair_workspace_bytes(len(views), arr, pooled_height, pooled_width), views[0][0].device)
    check(lib().mv3d_roi_pool_backward_views_pair(len(views), arr, pooled_height, pooled_width, _ptr(ws), ws.numel(), _stream()),
          "mv3d_roi_pool_backward_views_pair")
    if workspace is None:
        release_roi_pair_workspace(ws)        # (stream-ordered reuse: the next call on this stream runs behind these launches)
    return res


def rois_3d_to_fv(rois_3d, out=None):
    """rois_3d (R,7) device f32 [b,x,y,z,l,w,h] -> rois_fv (R,5) [b,x1,y1,x2,y2] on the 64 x 512 front-view map."""
    r3 = rois_3d.contiguous()
    R = r3.shape[0]
    if out is None:
        out = torch.empty((R, 5), dtype=torch.float32, device=r3.device)
    check(lib().mv3d_rois_3d_to_fv(_ptr(r3), R, _ptr(out), _stream()), "mv3d_rois_3d_to_fv")
    return out


def gt_encode(box_cam, cos_sin, inv_rot, tr):
    """box_cam (G,6) f32, cos_sin (G,2) f64, inv_rot (9) f32, tr (12) f32 device tensors -> (corners_cam (G,24),
    corners_lidar (G,24), boxes_3d (G,6), boxes_bv (G,4)) f32 in ONE buffer (pack, views): a host caller fetches it with
    one copy."""
    G = box_cam.shape[0]
    spec = [((G, 24), torch.float32), ((G, 24), torch.float32), ((G, 6), torch.float32), ((G, 4), torch.float32)]
    pack, views = packed_views(spec, box_cam.device)
    check(lib().mv3d_gt_encode(_ptr(box_cam), _ptr(cos_sin), G, _ptr(inv_rot), _ptr(tr), _ptr(views[0]), _ptr(views[1]),
                               _ptr(views[2]), _ptr(views[3]), _stream()), "mv3d_gt_encode")
    return pack, spec, views


# ------------------------------------------------------------------ training losses
def _loss_call(fn, name, cls, labels, pred, tgt, extra, sigma, want_grad):
    dev = cls.device
    R = cls.shape[0]
    losses = torch.empty((2,), dtype=torch.float32, device=dev)
    d_cls = torch.empty_like(cls) if want_grad else None
    d_pred = torch.empty_like(pred) if want_grad else None
    ws = _workspace(lib().mv3d_loss_workspace_bytes(R), dev, "loss")
    check(fn(_ptr(cls), _ptr(labels), _ptr(pred), _ptr(tgt), R, *extra, C.c_float(sigma), _ptr(losses), _ptr(d_cls), _ptr(d_pred),
             _ptr(ws), ws.numel(), _stream()), name)
    return losses, d_cls, d_pred


def softmax_rows(logits):
    """softmax over the last axis of an f32 device tensor (any leading shape), forward only: mv3d_softmax_rows"""
    x = logits.detach().contiguous()
    y = torch.empty_like(x)
    K = int(x.shape[-1])
    check(lib().mv3d_softmax_rows(_ptr(x), _ptr(y), x.numel() // max(K, 1), K, _stream()), "mv3d_softmax_rows")
    return y


def rpn_loss(rpn_cls_score, rpn_labels, rpn_bbox_pred, rpn_bbox_targets, sigma=3.0, want_grad=True):
    """(N,2) logits, (N) f32 labels in {-1,0,1}, (N,6) pred / targets -> (losses[2] = [cross-entropy, box], d_cls, d_pred)."""
    return _loss_call(lib().mv3d_rpn_loss, "mv3d_rpn_loss", rpn_cls_score, rpn_labels, rpn_bbox_pred, rpn_bbox_targets, (),
                      sigma, want_grad)


def rcnn_loss(cls_score, labels, bbox_pred, bbox_targets, sigma=3.0, want_grad=True):
    """(S,K) logits, (S) i32 labels, (S,D) pred / targets -> (losses[2], d_cls, d_pred)."""
    return _loss_call(lib().mv3d_rcnn_loss, "mv3d_rcnn_loss", cls_score, labels, bbox_pred, bbox_targets,
                      (cls_score.shape[1], bbox_pred.shape[1]), sigma, want_grad)



# ------------------------------------------------------------------ the serving trunk's contraction (f16 MFMA)
# (the `_f16` functions take float16, bfloat16 or float32 tensors and pick the C entry by the tensor's dtype: f16 = the serving trunk,
# bf16 = the training trunk, f32 = the serving trunk in the reference's precision)
def _half_entry(stem, dtype):
    if dtype == torch.float16:
        return getattr(lib(), stem + "_f16"), stem + "_f16"
    if dtype == torch.bfloat16:
        return getattr(lib(), stem + "_bf16"), stem + "_bf16"
    if dtype == torch.float32:
        return getattr(lib(), stem + "_f32"), stem + "_f32"
    raise TypeError("float16, bfloat16 or float32 tensors expected, got %s" % dtype)


def pack_conv3x3_weights(w_oihw, c_in_pad=None, dtype=torch.float16):
    """torch (O, I, 3, 3) f32 -> (O, 9 * I') f16 / bf16 with k = (ky * 3 + kx) * I' + c, I' = I zero-padded to c_in_pad."""
    O, I = w_oihw.shape[:2]
    Ip = c_in_pad or I
    w = torch.zeros((O, 3, 3, Ip), dtype=dtype, device=w_oihw.device)
    w[..., :I] = w_oihw.detach().permute(0, 2, 3, 1).to(dtype)
    return w.reshape(O, 9 * Ip).contiguous()


def pack_conv3x3_weights_input_layer(w_oihw, dtype=torch.float16):
    """the input layer's packing (c_in <= 16, see mv3d_conv3x3_f16): (O, I, 3, 3) f32 -> (O, 12 * 16), k = 16 * tap + c"""
    O, I = w_oihw.shape[:2]
    w = torch.zeros((O, 12, 16), dtype=dtype, device=w_oihw.device)
    w[:, :9, :I] = w_oihw.detach().permute(0, 2, 3, 1).reshape(O, 9, I).to(dtype)
    return w.reshape(O, 192).contiguous()


def framed_buffer(B, H, W, C, device, dtype=torch.float16):
    """zeroed (B, H + 2, W + 2, C): the kernels below only ever write its interior, so the SAME-padding frame stays 0"""
    return torch.zeros((B, H + 2, W + 2, C), dtype=dtype, device=device)


def frame_nhwc_f16(x_nhwc, out):
    """(B, H, W, C) f32 -> interior / first C channels of the framed f16 / bf16 buffer `out` (B, H + 2, W + 2, C' >= C)"""
    B, H, W, Cc = x_nhwc.shape
    fn, name = _half_entry("mv3d_frame_nhwc", out.dtype)
    check(fn(_ptr(x_nhwc), _ptr(out), B, H, W, Cc, out.shape[3], _stream()), name)
    return out


def conv3x3_f16(x_framed, w_packed, bias, out=None, out_framed=True, out_f32=False, relu=True):
    """x_framed (B, H + 2, W + 2, Cin) f16 / bf16, w_packed (Cout, 9 Cin) same type, bias (Cout) f32 -> framed map of that type
    (default), or the bare (B, H, W, Cout) map in that type / f32."""
    B, Hp, Wp, cin = x_framed.shape
    H, W, cout = Hp - 2, Wp - 2, w_packed.shape[0]
    if w_packed.dtype != x_framed.dtype:
        raise TypeError("activations %s, weights %s" % (x_framed.dtype, w_packed.dtype))
    fn, name = _half_entry("mv3d_conv3x3", x_framed.dtype)
    f32 = x_framed.dtype == torch.float32
    if f32:
        out_f32 = True                                            # f32 maps in, f32 maps out
    if out is None:
        if out_framed:
            out = framed_buffer(B, H, W, cout, x_framed.device, x_framed.dtype)
        else:
            out = torch.empty((B, H, W, cout), dtype=torch.float32 if out_f32 else x_framed.dtype, device=x_framed.device)
    # the kernel addresses each buffer with 32-bit offsets: frames are independent, so a larger batch goes in chunks
    per_frame = max(Hp * Wp * cin * x_framed.element_size(), out[0].numel() * out.element_size())
    step = max(1, min(B, (2 ** 31 - 1) // per_frame))
    for b0 in range(0, B, step):
        nb = min(step, B - b0)
        if f32:
            check(fn(_ptr(x_framed[b0:b0 + nb]), _ptr(w_packed), _ptr(bias), _ptr(out[b0:b0 + nb]), nb, H, W, cin, cout, int(out_framed), int(relu),
                     _stream()), name)
        else:
            check(fn(_ptr(x_framed[b0:b0 + nb]), _ptr(w_packed), _ptr(bias), _ptr(out[b0:b0 + nb]), nb, H, W, cin, cout, int(out_framed),
                     int(out_f32), int(relu), _stream()), name)
    return out


def maxpool2x2_f16(x_framed, out=None):
    B, Hp, Wp, Cc = x_framed.shape
    H, W = Hp - 2, Wp - 2
    if out is None:
        out = framed_buffer(B, H // 2, W // 2, Cc, x_framed.device, x_framed.dtype)
    fn, name = _half_entry("mv3d_maxpool2x2", x_framed.dtype)
    check(fn(_ptr(x_framed), _ptr(out), B, H, W, Cc, _stream()), name)
    return out



def maxpool2x2_bwd_bf16(y_framed, g_pooled_framed, out):
    """gradient of ReLU + 2x2 max pool into the zero-initialised framed buffer `out` (same shape / type as y_framed: bf16 or f32)"""
    B, Hp, Wp, Cc = y_framed.shape
    if y_framed.dtype not in (torch.bfloat16, torch.float32):
        raise TypeError("bfloat16 or float32 maps expected")
    fn, name = _half_entry("mv3d_maxpool2x2_bwd", y_framed.dtype)
    check(fn(_ptr(y_framed), _ptr(g_pooled_framed), _ptr(out), B, Hp - 2, Wp - 2, Cc, _stream()), name)
    return out


_WGRAD_CHUNK_BYTES = 2 ** 31 - 512          # what one weight-gradient launch may address per map


def conv3x3_wgrad_bf16(x_framed, dy_framed, c_in_real=None, want_bias=False):
    """x_framed (B, H + 2, W + 2, Cin), dy_framed (B, H + 2, W + 2, Cout) with a zero frame, both bf16 or both f32 -> the filter
    gradient (Cout, c_in_real, 3, 3) f32 (c_in_real <= Cin: the input layer's padding channels are dropped) [, the bias gradient
    (Cout) f32]"""
    B, Hp, Wp, cin = x_framed.shape
    cout = dy_framed.shape[3]
    if x_framed.dtype != dy_framed.dtype or x_framed.dtype not in (torch.bfloat16, torch.float32):
        raise TypeError("bfloat16 or float32 maps of one type expected, got %s / %s" % (x_framed.dtype, dy_framed.dtype))
    f32 = x_framed.dtype == torch.float32
    ws_fn = lib().mv3d_conv3x3_wgrad_f32_workspace_bytes if f32 else lib().mv3d_conv3x3_wgrad_workspace_bytes
    fn, name = (lib().mv3d_conv3x3_wgrad_f32, "mv3d_conv3x3_wgrad_f32") if f32 else (lib().mv3d_conv3x3_wgrad_bf16, "mv3d_conv3x3_wgrad_bf16")
    creal = cin if c_in_real is None else int(c_in_real)
    # the kernel addresses each map with 32-bit offsets: the gradient is a sum over the frames, so a larger batch goes in chunks
    per_frame = Hp * Wp * max(cin, cout) * x_framed.element_size()
    step = max(1, _WGRAD_CHUNK_BYTES // per_frame)
    if B > step:
        dw = db = None
        for b0 in range(0, B, step):
            r = conv3x3_wgrad_bf16(x_framed[b0:b0 + step], dy_framed[b0:b0 + step], creal, want_bias=want_bias)
            w_, b_ = r if want_bias else (r, None)
            dw = w_ if dw is None else dw.add_(w_)
            if want_bias:
                db = b_ if db is None else db.add_(b_)
        return (dw, db) if want_bias else dw
    need = ws_fn(B, Hp - 2, Wp - 2, cin, cout)
    if need == 0:
        raise _lib.Mv3dError(_lib.ERR_INVALID_ARG, name + "_workspace_bytes")
    ws = torch.empty(need, dtype=torch.uint8, device=x_framed.device)
    dw = torch.empty((cout, creal, 3, 3), dtype=torch.float32, device=x_framed.device)
    db = torch.empty((cout,), dtype=torch.float32, device=x_framed.device) if want_bias else None
    check(fn(_ptr(x_framed), _ptr(dy_framed), _ptr(dw), _ptr(db), B, Hp - 2, Wp - 2, cin, creal, cout, _ptr(ws), need, _stream()), name)
    return (dw, db) if want_bias else dw


def pack_conv3x3_train_bf16(w_oihw, c_in_pad=None, want_dgrad=True):
    """fp32 (O, I, 3, 3) -> (forward packing (O, 9 * I') bf16, data-gradient packing (I, 9 * O) bf16 | None) in one launch"""
    O, I = w_oihw.shape[:2]
    Ip = c_in_pad or I
    w = w_oihw.detach().contiguous()
    fwd = (torch.zeros if Ip > I else torch.empty)((O, 9 * Ip), dtype=torch.bfloat16, device=w.device)
    dg = torch.empty((I, 9 * O), dtype=torch.bfloat16, device=w.device) if want_dgrad else None
    check(lib().mv3d_conv3x3_pack_bf16(_ptr(w), _ptr(fwd), _ptr(dg), O, I, Ip, _stream()), "mv3d_conv3x3_pack_bf16")
    return fwd, dg


def conv3x3_gated_bf16(x_framed, w_packed, bias, gate_framed, out):
    """(conv3x3(x) + bias) gated by gate > 0 into the framed bf16 buffer `out` (the training trunk's data gradient + ReLU mask)"""
    B, Hp, Wp, cin = x_framed.shape
    cout = w_packed.shape[0]
    check(lib().mv3d_conv3x3_gated_bf16(_ptr(x_framed), _ptr(w_packed), _ptr(bias), _ptr(gate_framed), _ptr(out), B, Hp - 2, Wp - 2, cin, cout,
                                        _stream()), "mv3d_conv3x3_gated_bf16")
    return out


# ------------------------------------------------------------------ several views of one layer shape behind one launch
# (the BEV / image / front-view trunks at one VGG depth: lib/networks/MV3D_train.py:44-81 -- same channel counts, different map
# sizes; at a training batch one view's tiles do not fill the chip, together they do, on one stream)
_OFF32 = 2 ** 31 - 1


def conv3x3_views(views, out_framed=True, out_f32=False, relu=True):
    """views: list of (x_framed, w_packed, bias, gate_framed | None, out); the same (c_in, c_out) and type in every view.
    One launch (mv3d_conv3x3_views_*); `out` is written (framed of the operand type, or bare in that type / f32)."""
    x0, w0 = views[0][0], views[0][1]
    cin, cout, dt = x0.shape[3], w0.shape[0], x0.dtype
    f32 = dt == torch.float32
    big = False
    for x, w, b, g, out in views:
        if x.dtype != dt or w.dtype != dt or x.shape[3] != cin or w.shape[0] != cout:
            raise TypeError("views of one layer shape and type expected")
        big = big or x.numel() * x.element_size() > _OFF32 or out.numel() * out.element_size() > _OFF32
    if big or len(views) > 3:
        # beyond the kernel's 32-bit offsets (or its three view slots): view by view, each in batch chunks that fit, through the
        # one-view form of this same entry -- so the gated (data-gradient) views keep their own type (bf16 or f32: ADVICE r04)
        for x, w, b, g, out in views:
            per_frame = max(x[0].numel() * x.element_size(), out[0].numel() * out.element_size())
            step = _OFF32 // per_frame
            if step < 1:
                raise ValueError("one frame of %d bytes is beyond the convolution kernel's 32-bit offsets" % per_frame)
            for b0 in range(0, x.shape[0], step):
                conv3x3_views([(x[b0:b0 + step], w, b, None if g is None else g[b0:b0 + step], out[b0:b0 + step])],
                              out_framed=out_framed, out_f32=out_f32, relu=relu)
        return [v[4] for v in views]
    arr = (_lib.ConvView * len(views))()
    for k, (x, w, b, g, out) in enumerate(views):
        B, Hp, Wp, _ = x.shape
        arr[k] = _lib.ConvView(x.data_ptr(), w.data_ptr(), b.data_ptr(), g.data_ptr() if g is not None else None, out.data_ptr(),
                               B, Hp - 2, Wp - 2, 0)
    fn, name = _half_entry("mv3d_conv3x3_views", dt)
    if f32:
        check(fn(len(views), arr, cin, cout, int(out_framed), int(relu), _stream()), name)
    else:
        check(fn(len(views), arr, cin, cout, int(out_framed), int(out_f32), int(relu), _stream()), name)
    return [v[4] for v in views]


def conv3x3_pool_views(views):
    """views: list of (x_framed, w_packed, bias, out_pooled_framed): convolution + bias + ReLU + the 2x2 max pool behind it in one
    launch (mv3d_conv3x3_pool_views_*); `out` (B, H / 2 + 2, W / 2 + 2, Cout) is the POOLED framed map."""
    x0, w0 = views[0][0], views[0][1]
    cin, cout, dt = x0.shape[3], w0.shape[0], x0.dtype
    arr = (_lib.ConvView * len(views))()
    for k, (x, w, b, out) in enumerate(views):
        B, Hp, Wp, _ = x.shape
        if x.dtype != dt or w.dtype != dt or x.shape[3] != cin or w.shape[0] != cout or tuple(out.shape) != (B, (Hp - 2) // 2 + 2, (Wp - 2) // 2 + 2, cout):
            raise TypeError("views of one layer shape and type, pooled framed outputs expected")
        if x.numel() * x.element_size() > _OFF32:
            raise ValueError("map beyond the kernel's 32-bit offsets: pool separately")
        arr[k] = _lib.ConvView(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, out.data_ptr(), B, Hp - 2, Wp - 2, 0)
    fn, name = _half_entry("mv3d_conv3x3_pool_views", dt)
    check(fn(len(views), arr, cin, cout, _stream()), name)
    return [v[3] for v in views]


def maxpool2x2_views(views):
    """views: list of (x_framed, out_framed): the 2x2 pools of several maps of one channel count in one launch"""
    arr = (_lib.PoolView * len(views))()
    for k, (x, out) in enumerate(views):
        B, Hp, Wp, _ = x.shape
        arr[k] = _lib.PoolView(x.data_ptr(), None, out.data_ptr(), B, Hp - 2, Wp - 2, 0)
    fn, name = _half_entry("mv3d_maxpool2x2_views", views[0][0].dtype)
    check(fn(len(views), arr, views[0][0].shape[3], _stream()), name)
    return [v[1] for v in views]


def maxpool2x2_bwd_views(views):
    """views: list of (y_framed, g_pooled_framed, out): gradient of ReLU + 2x2 max pool of several maps in one launch"""
    dt = views[0][0].dtype
    if dt not in (torch.bfloat16, torch.float32):
        raise TypeError("bfloat16 or float32 maps expected")
    arr = (_lib.PoolView * len(views))()
    for k, (y, g, out) in enumerate(views):
        B, Hp, Wp, _ = y.shape
        arr[k] = _lib.PoolView(y.data_ptr(), g.data_ptr(), out.data_ptr(), B, Hp - 2, Wp - 2, 0)
    fn, name = _half_entry("mv3d_maxpool2x2_bwd_views", dt)
    check(fn(len(views), arr, views[0][0].shape[3], _stream()), name)
    return [v[2] for v in views]


def conv3x3_wgrad_views(views, c_in_real=None, want_bias=False):
    """views: list of (x_framed, dy_framed) of one layer shape -> [(dw (Cout, c_in_real, 3, 3) f32, db (Cout) f32 | None), ...]:
    one weight-gradient launch + one reduce launch for all of them.  c_in_real: one number, or one per view (the input layers' 9 / 3
    real channels inside their 64-channel buffers)."""
    x0, dy0 = views[0]
    cin, cout, dt = x0.shape[3], dy0.shape[3], x0.dtype
    if dt not in (torch.bfloat16, torch.float32):
        raise TypeError("bfloat16 or float32 maps expected")
    reals = list(c_in_real) if isinstance(c_in_real, (list, tuple)) else [cin if c_in_real is None else int(c_in_real)] * len(views)
    creal = reals[0]
    if len(views) > 3 or any(x.numel() * x.element_size() > _OFF32 - 256 or dy.numel() * dy.element_size() > _OFF32 - 256 for x, dy in views):
        return [conv3x3_wgrad_bf16(x, dy, r, want_bias=True) if want_bias else (conv3x3_wgrad_bf16(x, dy, r), None) for (x, dy), r in zip(views, reals)]
    dev = x0.device
    arr = (_lib.WgradView * len(views))()
    res = []
    for k, (x, dy) in enumerate(views):
        if x.dtype != dt or dy.dtype != dt or x.shape[3] != cin or dy.shape[3] != cout:
            raise TypeError("views of one layer shape and type expected")
        B, Hp, Wp, _ = x.shape
        dw = torch.empty((cout, reals[k], 3, 3), dtype=torch.float32, device=dev)
        db = torch.empty((cout,), dtype=torch.float32, device=dev) if want_bias else None
        arr[k] = _lib.WgradView(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr() if db is not None else None, B, Hp - 2, Wp - 2, reals[k])
        res.append((dw, db))
    f32 = dt == torch.float32
    need = lib().mv3d_conv3x3_wgrad_views_workspace_bytes(len(views), arr, cin, cout, int(f32))
    if need == 0:
        raise _lib.Mv3dError(_lib.ERR_INVALID_ARG, "mv3d_conv3x3_wgrad_views_workspace_bytes")
    ws = _workspace(need, dev, "wgrad")
    fn, name = (lib().mv3d_conv3x3_wgrad_views_f32, "mv3d_conv3x3_wgrad_views_f32") if f32 else \
        (lib().mv3d_conv3x3_wgrad_views_bf16, "mv3d_conv3x3_wgrad_views_bf16")
    check(fn(len(views), arr, cin, creal, cout, _ptr(ws), ws.numel(), _stream()), name)
    return res


def pack_conv3x3_train_many(items, dtype=torch.bfloat16):
    """items: list of (w_oihw f32 (O, I, 3, 3), c_in_pad | None, want_dgrad) -> [(fwd (O, 9 I') of `dtype` (bf16 | f32), dgrad (I, 9 O)
    | None)]: every filter of a training step packed by ONE launch."""
    arr = (_lib.PackItem * len(items))()
    res, keep = [], []
    for k, (w_oihw, c_in_pad, want_dgrad) in enumerate(items):
        O, I = w_oihw.shape[:2]
        Ip = c_in_pad or I
        w = w_oihw.detach().contiguous()
        fwd = (torch.zeros if Ip > I else torch.empty)((O, 9 * Ip), dtype=dtype, device=w.device)
        dg = torch.empty((I, 9 * O), dtype=dtype, device=w.device) if want_dgrad else None
        arr[k] = _lib.PackItem(w.data_ptr(), fwd.data_ptr(), dg.data_ptr() if dg is not None else None, O, I, Ip, 0)
        keep.append(w)
        res.append((fwd, dg))
    fn, name = _half_entry("mv3d_conv3x3_pack_many", dtype)
    check(fn(len(items), arr, _stream()), name)
    return res


pack_conv3x3_train_many_bf16 = pack_conv3x3_train_many


# type-neutral names (the `_f16` / `_bf16` suffixes above are historical: each function picks the C entry by its tensors' dtype)
frame_nhwc = frame_nhwc_f16
conv3x3 = conv3x3_f16
maxpool2x2 = maxpool2x2_f16
maxpool2x2_bwd = maxpool2x2_bwd_bf16
conv3x3_wgrad = conv3x3_wgrad_bf16
