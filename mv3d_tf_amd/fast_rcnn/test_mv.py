"""Test-time entry points of lib/fast_rcnn/test_mv.py: `box_detect` (:149-264), the per-frame post-processing of
`test_net` (:420-499: per-class score cut, NMS, cap over all classes) and the `test_net` loop itself (:321-518, without
the drawing).

`box_detect(sess, net, im, bv, calib, boxes=None)`: interface of
lib/fast_rcnn/test_mv.py:149-264.  `sess` is an opaque context (ignored: there is no TF
session).  Returns (scores (R,K), pred_boxes_bv (R,4K) f64, pred_boxes_cnr (R,24K) f32,
pred_boxes_cnr_r (R,24K) f32) like the reference, K = 2; the geometric tail runs in
libmv3d_hip.so (mv3d_box_detect_tail)."""
import os
import time
import pickle

import numpy as np
import torch

from .. import ops
from .config import cfg, get_output_dir
from .nms_wrapper import nms
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


def class_detections(scores, boxes_bv, boxes_cnr, boxes_cnr_r, num_classes, thresh=0.05):
    """One frame of test_net's inner loop (lib/fast_rcnn/test_mv.py:423-444): for every foreground class j the rows
    with score > thresh, as (N,5) BEV dets / (N,25) corner dets / (N,25) regressed-corner dets [.., score], after
    `nms(cls_dets, cfg.TEST.NMS)` on the BEV boxes.  Index 0 (background) is an empty list, like all_boxes[0][i]."""
    dets, dets_cnr, dets_cnr_r = [[]], [[]], [[]]
    for j in range(1, num_classes):                                    # :423 skip j = 0, the background class
        inds = np.where(scores[:, j] > thresh)[0]
        cls_scores = scores[inds, j]
        cls_dets = np.hstack((boxes_bv[inds, j * 4:(j + 1) * 4], cls_scores[:, np.newaxis])).astype(np.float32, copy=False)
        cls_dets_cnr = np.hstack((boxes_cnr[inds, j * 24:(j + 1) * 24], cls_scores[:, np.newaxis])).astype(np.float32, copy=False)
        cls_dets_cnr_r = np.hstack((boxes_cnr_r[inds, j * 24:(j + 1) * 24], cls_scores[:, np.newaxis])).astype(np.float32, copy=False)
        keep = nms(cls_dets, cfg.TEST.NMS)                             # :440 (greedy NMS on the device)
        dets.append(cls_dets[keep, :]); dets_cnr.append(cls_dets_cnr[keep, :]); dets_cnr_r.append(cls_dets_cnr_r[keep, :])
    return dets, dets_cnr, dets_cnr_r


def limit_detections(dets, dets_cnr, max_per_image):
    """lib/fast_rcnn/test_mv.py:488-499: keep at most max_per_image detections over all classes (score >= the
    max_per_image-th largest score; ties may keep more, as in the reference)."""
    if max_per_image > 0:
        image_scores = np.hstack([dets[j][:, -1] for j in range(1, len(dets))]) if len(dets) > 1 else np.zeros(0)
        if len(image_scores) > max_per_image:
            image_thresh = np.sort(image_scores)[-max_per_image]
            for j in range(1, len(dets)):
                keep = np.where(dets[j][:, -1] >= image_thresh)[0]
                dets[j] = dets[j][keep, :]
                dets_cnr[j] = dets_cnr[j][keep, :]
    return dets, dets_cnr


def test_net(sess, net, imdb, weights_filename, max_per_image=300, thresh=0.05, vis=False):
    """lib/fast_rcnn/test_mv.py:321-518 without the drawing: runs `box_detect` on every frame of `imdb`, collects
    all_boxes[cls][image] (N,5) and all_boxes_cnr[cls][image] (N,25), pickles them under get_output_dir() and calls
    imdb.evaluate_detections.  `imdb` is duck-typed: image_index, num_classes, name, calib_at(i), evaluate_detections,
    and either image_at(i) / bv_at(i) (arrays) or image_path_at(i) / lidar_path_at(i) (files: .npy for the BEV, an
    image readable by numpy / PIL).  As in the reference the `thresh` argument is shadowed by 0.05 (:421)."""
    if hasattr(net, "mfma_trunk") and (cfg.TEST.get("MFMA_TRUNK", False) or cfg.TEST.get("PRECISION", "fp32") != "fp32"):
        net.amp_dtype = {"fp32": None, "fp16": torch.float16, "bf16": torch.bfloat16}[cfg.TEST.get("PRECISION", "fp32")]
        net.mfma_trunk = bool(cfg.TEST.get("MFMA_TRUNK", False))
    num_images = len(imdb.image_index)
    all_boxes = [[[] for _ in range(num_images)] for _ in range(imdb.num_classes)]
    all_boxes_cnr = [[[] for _ in range(num_images)] for _ in range(imdb.num_classes)]
    output_dir = get_output_dir(imdb, weights_filename)
    t_detect = t_misc = 0.0                                            # the reference's _t['im_detect'] / _t['misc'] timers (:355, :433-436, :482-502)
    for i in range(num_images):
        if hasattr(imdb, "image_at"):
            im, bv = imdb.image_at(i), imdb.bv_at(i)
        else:
            bv = np.load(imdb.lidar_path_at(i))
            path = imdb.image_path_at(i)
            if path.endswith(".npy"):
                im = np.load(path)
            else:
                from PIL import Image                               # (the reference uses cv2.imread: BGR)
                im = np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1]
        calib = imdb.calib_at(i)
        t0 = time.time()
        scores, boxes_bv, boxes_cnr, boxes_cnr_r = box_detect(sess, net, im, bv, calib, None)
        t1 = time.time()
        t_detect += t1 - t0
        dets, dets_cnr, _ = class_detections(scores, boxes_bv, boxes_cnr, boxes_cnr_r, imdb.num_classes, 0.05)
        dets, dets_cnr = limit_detections(dets, dets_cnr, max_per_image)
        for j in range(1, imdb.num_classes):
            all_boxes[j][i] = dets[j]
            all_boxes_cnr[j][i] = dets_cnr[j]
        t_misc += time.time() - t1
        print('im_detect: {:d}/{:d} {:.3f}s {:.3f}s'.format(i + 1, num_images, t_detect / (i + 1), t_misc / (i + 1)))   # lib/fast_rcnn/test_mv.py:504-506
    with open(os.path.join(output_dir, 'detections.pkl'), 'wb') as f:
        pickle.dump(all_boxes, f, pickle.HIGHEST_PROTOCOL)
    with open(os.path.join(output_dir, 'detections_cnr.pkl'), 'wb') as f:
        pickle.dump(all_boxes_cnr, f, pickle.HIGHEST_PROTOCOL)
    print('Evaluating detections')
    imdb.evaluate_detections(all_boxes, all_boxes_cnr, output_dir)
    return all_boxes, all_boxes_cnr



class ServeGraph:
    """BASELINE configs[4]'s serving step as ONE captured hipGraph: trunks, RPN heads + softmax, proposal_layer_3d (TEST cfg), front-view
    ROIs, RoiPool of every view, the fusion head and the box tail (lib/fast_rcnn/test_mv.py:149-264 for a batch), no host round trip
    inside.  Shapes are fixed: every frame owns `rois_per_frame` (= RPN_POST_NMS_TOP_N) ROI rows; the rows behind a frame's num_rois are
    zero boxes that are pooled and scored like any other and dropped by `detections()` -- the ROI counts are read ONCE, after the replay.
    Inputs are copied into the graph's static buffers on the replay stream; outputs are the graph's static tensors (valid until the next
    replay)."""

    def __init__(self, net, feed, warmup=2):
        self.net = net
        dev = net.device
        self.static = {k: (v.to(dev).clone() if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v, np.float32)).to(dev))
                       for k, v in feed.items() if k != "keep_prob"}
        self.static["keep_prob"] = 1.0
        self.stream = torch.cuda.Stream(device=dev)
        self.out = None
        net.fixed_rois = True
        try:
            with torch.cuda.stream(self.stream):
                for _ in range(warmup):                             # (weights cast / packed, workspaces allocated: nothing of that is captured)
                    self._step()
            self.stream.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=self.stream):
                self.out = self._step()
        finally:
            net.fixed_rois = False

    def _step(self):
        with torch.no_grad():
            L = self.net.forward(self.static)
            rois3 = L["rois"][2].contiguous()
            cnr, pred_r, pred_bv, _ = ops.box_detect_tail(rois3, L["bbox_pred"].contiguous(), n_classes)
            return {"cls_prob": L["cls_prob"], "bbox_pred": L["bbox_pred"], "rois_bv": L["rois"][0], "rois_img": L["rois"][1], "rois_3d": rois3,
                    "corners": cnr, "pred_corners_r": pred_r, "pred_bv": pred_bv, "num_rois": L["num_rois"], "status": L["rois_status"],
                    "rois_per_frame": L["rois_per_frame"]}

    def replay(self, feed=None):
        """enqueue one step (asynchronous); feed: new inputs for the static buffers (same shapes), or None to reuse them"""
        with torch.cuda.stream(self.stream):
            if feed is not None:
                for k, v in feed.items():
                    if k != "keep_prob":
                        self.static[k].copy_(v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v, np.float32)), non_blocking=True)
            self.graph.replay()
        return self.out

    def detections(self):
        """after a replay: per frame (scores, pred_bv, corners, pred_corners_r) of its num_rois rows -- the ONE host round trip of the step"""
        self.stream.synchronize()
        o = self.out
        num, status, cap = o["num_rois"].cpu().numpy(), o["status"].cpu().numpy(), int(o["rois_per_frame"])
        if int(status.max()) & 1:
            raise ZeroDivisionError("float division")
        host = {k: o[k].float().cpu().numpy() for k in ("cls_prob", "pred_bv", "corners", "pred_corners_r")}
        return [tuple(host[k][b * cap:b * cap + int(num[b])] for k in ("cls_prob", "pred_bv", "corners", "pred_corners_r")) for b in range(len(num))]


def bench_serve_step(rank, world, dist, batch=16, seconds=1.0, warmup=2, dtypes=("fp32", "fp32_mfma", "fp16", "fp16_mfma", "fp16_mfma_graph"),
                     reduce_device="cuda", views=3, steps=None):
    """Full MV3D_test forward WITH the dense layers, for bench.py's `serving_with_trunk` key (BASELINE configs[4]: batch
    16 / GPU, TEST cfg 6000 -> 300, "fp16 VGG16"): `batch` synthetic KITTI-shaped frames per step through the trunks /
    FC head, proposal_layer_3d, RoiPool of both views and the box tail.  fp32 is the reference's precision (torch: MIOpen /
    rocBLAS); fp16 = autocast of the DENSE layers only (MIOpen / rocBLAS half kernels); fp16_mfma = the 27 3x3 convolutions
    on this repository's MFMA kernel (mv3d_conv3x3_f16, mv3d_tf_amd.trunk), the FC head still autocast rocBLAS;
    fp16_mfma_graph = the same step captured once as a hipGraph over fixed shapes (ServeGraph: no host sync per step, the ROI counts
    read after the window).  The hot-path layers stay f32 in every variant; the f16 variants are a lower precision than the reference,
    reported next to fp32, never the headline.  Every variant is timed over >= `seconds` (a fixed step count per variant, the same on
    every rank) and reports the minimum and median step beside the mean."""
    from .. import sharding, synth
    from ..networks import get_network
    from ..utils.timing import step_stats, timed_steps
    net = get_network("MV3D_test_3view" if views == 3 else "MV3D_test")     # configs[4] serves the full 3-view model
    rng = np.random.RandomState(200 + rank)
    bev = torch.as_tensor(((rng.random_sample((batch, 608, 608, 9)) < 0.03) * rng.uniform(0, 2.4, (batch, 608, 608, 9))).astype(np.float32)).cuda()
    img = torch.as_tensor((rng.randint(0, 255, (batch, 375, 1242, 3)) - cfg.PIXEL_MEANS).astype(np.float32)).cuda()
    feed = {"lidar_bv_data": bev, "image_data": img, "im_info": np.array([[608, 608, 1]] * batch, np.float32),
            "calib": np.stack([synth.KITTI_CALIB] * batch), "keep_prob": 1.0}
    if views == 3:
        feed["lidar_fv_data"] = torch.as_tensor(rng.uniform(0, 1, (batch, 64, 512, 3)).astype(np.float32)).cuda()
    # BASELINE configs[4]: "300 proposals/frame" = the 6000 -> 300 TEST setting the reference's config.py keeps as a comment
    # (lib/fast_rcnn/config.py:186-190; its live default is 12000 -> 2000)
    saved = (cfg.TEST.RPN_PRE_NMS_TOP_N, cfg.TEST.RPN_POST_NMS_TOP_N)
    cfg.TEST.RPN_PRE_NMS_TOP_N, cfg.TEST.RPN_POST_NMS_TOP_N = 6000, 300
    out = {"workload": "MV3D_test%s full forward incl. VGG16 trunks + FC head + proposal_layer_3d (TEST cfg 6000 -> 300) + "
                       "RoiPool x%d + box tail: batch %d / GPU, 608x608x9 BEV + 375x1242x3 image%s"
                       % ("_3view" if views == 3 else "", views, batch, " + 64x512x3 front view" if views == 3 else "")}
    rois = [0]
    # steps per variant for a >= `seconds` window at batch 16 on one MI355X (ms per step measured in round 5: 130 / 101 / 47 / 13.4);
    # fixed numbers, not a calibration run, so that every rank times the same window
    ms_guess = {"fp32": 130.0, "fp32_mfma": 101.0, "fp16": 47.0, "bf16": 47.0, "fp16_mfma": 13.4, "fp16_mfma_graph": 12.5}

    def barrier():
        torch.cuda.synchronize()
        if world > 1 and dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def eager_step():
        with torch.no_grad():
            L = net.forward(feed)
            ops.box_detect_tail(L["rois"][2].contiguous(), L["bbox_pred"].contiguous(), n_classes)
            rois[0] = int(L["rois"][2].shape[0])

    try:
        for name in dtypes:
            net.amp_dtype = {"fp32": None, "fp32_mfma": None, "fp16": torch.float16, "bf16": torch.bfloat16, "fp16_mfma": torch.float16,
                             "fp16_mfma_graph": torch.float16}[name]
            net.mfma_trunk = "_mfma" in name
            n = steps or max(8, int(np.ceil(seconds * 1e3 / (ms_guess[name] * batch / 16.0))))
            extra = {}
            if name.endswith("_graph"):
                sg = ServeGraph(net, feed)
                with torch.cuda.stream(sg.stream):
                    dt, ms = timed_steps(sg.replay, n, barrier)
                rois[0] = int(sg.out["num_rois"].sum().item())           # (the ONE read of the counts, after the window)
                extra = {"rows_per_step": int(sg.out["cls_prob"].shape[0]),
                         "launch": "one hipGraph replay per step, fixed shapes (%d ROI rows per frame), no host sync inside the window" % int(sg.out["rois_per_frame"])}
                del sg
            else:
                for _ in range(warmup):
                    eager_step()
                dt, ms = timed_steps(eager_step, n, barrier)
            dt = sharding.max_over_ranks(dt, dist if world > 1 else None, device=reduce_device)
            out[name] = dict({"frames_per_s": round(n * batch * world / dt, 2), "ms_per_step": round(dt / n * 1e3, 3), "timed_s": round(dt, 3),
                              "rois_per_step": rois[0]}, **step_stats(ms), **extra)
            torch.cuda.empty_cache()
    finally:
        cfg.TEST.RPN_PRE_NMS_TOP_N, cfg.TEST.RPN_POST_NMS_TOP_N = saved
        net.fixed_rois = False
    out["note"] = ("fp32_mfma = the reference's precision with the 3x3 convolutions on this library's exact-f32 MFMA kernel; "
                   "fp16 = autocast of the dense layers only; fp16_mfma = 3x3 convolutions on the hand-written f16 MFMA kernel "
                   "(f32 accumulate), FC head autocast; fp16_mfma_graph = that step as one captured hipGraph; all 16-bit variants are a lower "
                   "precision than the reference's fp32; the hot-path layers run in f32.  ms_per_step = wall time of the window / steps; "
                   "_min / _median from one HIP event per step")
    return out


def bench_config1_latency(reps=30, warmup=5, seed=11):
    """BASELINE configs[1] for bench.py's `secondary.config1_latency`: ONE synthetic KITTI frame, BEV view only --
    608 x 608 x 9 BEV -> VGG16 trunk (conv1_1 .. conv5_3, lib/networks/MV3D_test.py:34-58) and rpn_conv/3x3 on this library's
    MFMA convolution -> the two 1 x 1 RPN heads + pairwise softmax (:82-93) -> mv3d_proposal_3d with the TEST cfg (23 104
    anchors -> 6000 pre-NMS -> HIP NMS 0.7 -> 300 proposals, projection to the image included), on ONE stream, batch 1.
    Per precision (f32 = the reference's, on the exact-f32 MFMA; f16 = f16 operands / f32 accumulation): the frame's latency
    with eager launches (host enqueue + device time, synchronised after every frame) and as a captured hipGraph replayed and
    synchronised per frame."""
    import ctypes
    import time
    import torch.nn.functional as F
    from .. import synth
    from ..networks import get_network
    from ..networks.mv3d import _VGG
    from ..trunk import MfmaTrunks
    dev = torch.device("cuda", torch.cuda.current_device())
    net = get_network("MV3D_test")
    rng = np.random.RandomState(seed)
    bev = torch.as_tensor(((rng.random_sample((1, 608, 608, 9)) < 0.03) * rng.uniform(0, 2.4, (1, 608, 608, 9))).astype(np.float32)).to(dev)
    info = torch.as_tensor(np.array([[608, 608, 1]], np.float32)).to(dev)
    cal = torch.as_tensor(synth.KITTI_CALIB[None].astype(np.float32)).to(dev)
    saved = (cfg.TEST.RPN_PRE_NMS_TOP_N, cfg.TEST.RPN_POST_NMS_TOP_N)
    cfg.TEST.RPN_PRE_NMS_TOP_N, cfg.TEST.RPN_POST_NMS_TOP_N = 6000, 300
    params = ops.proposal_params(cfg.TEST, feat_stride=8)
    out = {"workload": "BASELINE configs[1]: 1 frame, BEV only: 608x608x9 -> VGG16 trunk + rpn_conv/3x3 (14 MFMA convolutions, 3 pools) -> "
                       "1x1 RPN heads + softmax -> mv3d_proposal_3d TEST cfg (23104 anchors -> 6000 -> NMS 0.7 -> 300); one stream, batch 1"}
    try:
        for name, dt in (("f32", torch.float32), ("f16", torch.float16)):
            tr = MfmaTrunks(net, _VGG, dtype=dt)
            heads = [(net.params[k][0].detach().reshape(net.params[k][0].shape[0], -1).to(dt).contiguous(),
                      net.params[k][1].detach().to(dt).contiguous()) for k in ("rpn_cls_score", "rpn_bbox_pred")]
            cap = ops.lib().mv3d_proposal_3d_capacity(76, 76, ctypes.byref(params))
            outs = ops.proposal_3d_outputs(1, cap, dev)[1]

            def frame():
                with torch.no_grad():
                    rpn = tr.rpn_conv(tr.trunk(bev, "", last_framed=True))                       # (1, 76, 76, 512)
                    score = F.linear(rpn, *heads[0]).float()
                    pred = F.linear(rpn, *heads[1]).float().contiguous()
                    prob = F.softmax(score.reshape(-1, 2), dim=1).reshape(1, 76, 76, 8)          # reshape_layer(2) + softmax
                    return ops.proposal_3d(prob, pred, info, cal, params, out=outs)

            side = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    frame()
                side.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    frame()
                    side.synchronize()
                eager = (time.perf_counter() - t0) / reps
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(side)
                for _ in range(reps):
                    frame()
                e1.record(side)
                side.synchronize()
                device_ms = e0.elapsed_time(e1) / reps
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                res = frame()
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    g.replay()
                side.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    g.replay()
                    side.synchronize()
                graph = (time.perf_counter() - t0) / reps
            out[name] = {"ms_per_frame_eager": round(eager * 1e3, 4), "ms_per_frame_graph": round(graph * 1e3, 4),
                         "ms_per_frame_device_back_to_back": round(device_ms, 4), "proposals": int(res[3][0].item())}
            del g, tr
    finally:
        cfg.TEST.RPN_PRE_NMS_TOP_N, cfg.TEST.RPN_POST_NMS_TOP_N = saved
    return out
