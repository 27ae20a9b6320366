/*
 * mv3d_hip.h -- C-ABI of libmv3d_hip.so, the MI355X (gfx950) implementation of the
 * MV3D RPN -> ROI-pool -> NMS hot path.
 *
 * Drop-in boundary: every entry point below replaces one native or
 * numpy interface of the reference (cited per function, paths relative to the upstream
 * repository root).  Plain C: raw device pointers, sizes, an opaque stream handle
 * (a hipStream_t passed as void*; NULL = the null stream).  No torch types, no
 * exceptions, no exit(): every call returns an mv3d_status.  Hot calls never allocate:
 * the caller owns outputs and a workspace whose size the *_workspace_bytes() queries
 * return.  All calls are asynchronous on `stream` unless stated otherwise and are
 * re-entrant per (stream, workspace).
 *
 * Frames are an outer, independent dimension (`batch`): the reference asserts batch==1
 * in its RPN layers (lib/rpn_msr/proposal_layer_tf.py:48-49); here frame b of a batch
 * gives exactly what the reference gives for that frame alone, with ROI column 0 = b.
 */
#ifndef MV3D_HIP_H
#define MV3D_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    MV3D_OK = 0,
    MV3D_ERR_INVALID_ARG = 1,   /* bad shape / NULL pointer / unsupported size */
    MV3D_ERR_WORKSPACE = 2,     /* workspace too small or misaligned (256 B) */
    MV3D_ERR_HIP = 3,           /* a HIP runtime call or launch failed */
    MV3D_ERR_ZERO_DIVISION = 4  /* host entries only: the reference's ZeroDivisionError */
} mv3d_status;

int mv3d_version(void);                      /* 100 * major + minor */
const char *mv3d_status_string(int status);

/* ------------------------------------------------------------------ NMS
 * Replaces: lib/nms/cpu_nms.pyx:17-68 (== lib/utils/nms.pyx:17-68), dispatcher
 * lib/fast_rcnn/nms_wrapper.py:13-21, and the CUDA path lib/nms/nms_kernel.cu:34-144 /
 * lib/nms/gpu_nms.hpp:1-2.
 *
 * Semantics (CPU path = the parity target): f32 IoU with the +1 pixel convention,
 * separate IEEE f32 operations, box j suppressed by an earlier kept box iff
 * (double)IoU >= thresh.  The device-status word receives bit 0 = "a union was 0"
 * (the reference raises ZeroDivisionError there; the pair is treated as not
 * suppressing and the flag is raised). */

/* Device entry: dets_dev (n,5) f32 [x1,y1,x2,y2,score], ALREADY in processing order
 * (descending score).  keep_dev receives positions 0..n-1 of the kept boxes in order,
 * at most max_keep of them (max_keep <= 0: no cap; the result equals the reference's
 * keep[:max_keep]); num_keep_dev[0] the count; status_dev (may be NULL): flag bits are OR-ed into
 * status_dev[0], which the caller zeroes once when it allocates it (the call is kernel launches only:
 * no memset node, so it can be captured in a hipGraph and replayed). */
size_t mv3d_nms_workspace_bytes(int max_boxes);
int mv3d_nms_device(const float *dets_dev, int n, double thresh, int max_keep,
                    int32_t *keep_dev, int32_t *num_keep_dev, int32_t *status_dev,
                    void *workspace, size_t workspace_bytes, void *stream);

/* Diagnostics: mv3d_nms_device plus a per-64-box-block trace of the greedy chain,
 * trace_dev[4*b + 0..3] = {cycle at block start, cycle after waiting for the bulk workers,
 * cycle after the diagonal fixed point, (iterations << 32) | kept-in-block}, followed by 8
 * phase stamps of the first round's chain kernel {start, first tiles in LDS, chain done, end};
 * trace_dev holds 4*ceil(n/64) + 8 entries.  Used by tools/ and profiles/ only. */
int mv3d_nms_device_trace(const float *dets_dev, int n, double thresh, int max_keep,
                          int32_t *keep_dev, int32_t *num_keep_dev, int32_t *status_dev,
                          void *workspace, size_t workspace_bytes, void *stream, int64_t *trace_dev);

/* Host entry with cpu_nms(dets, thresh) semantics: host pointers, dets unsorted;
 * sorts on the device (descending score, ties by descending index -- the reference
 * leaves ties to numpy's unstable argsort), runs the NMS, returns indices into dets in
 * processing order.  Synchronous; allocates and frees its own device buffers (like
 * _nms below).  Returns MV3D_ERR_ZERO_DIVISION where the reference raises. */
int mv3d_nms_host(int32_t *keep_out, int32_t *num_out, const float *dets_host, int n,
                  double thresh, int device_id);

/* Symbol-compatible replacement for lib/nms/gpu_nms.hpp:1-2 (bound by gpu_nms.pyx:13-14):
 * host pointers, boxes pre-sorted by the caller, synchronous, selects device_id.  Keeps
 * the CUDA path's own rule (nms_kernel.cu:71): suppressed iff IoU > thresh compared in
 * f32 -- NOT identical to the CPU path; parity of this entry is
 * unpinned (no CUDA device to run the original on). */
void _nms(int *keep_out, int *num_out, const float *boxes_host, int boxes_num,
          int boxes_dim, float nms_overlap_thresh, int device_id);

/* ------------------------------------------------------------------ proposal_layer_3d
 * Replaces the numpy py_func body lib/rpn_msr/proposal_layer_tf.py:25-202 and all it
 * calls (bv_anchor_to_lidar, bbox_transform_inv_3d, lidar_3d_to_bv, lidar_3d_to_corners,
 * lidar_cnr_to_img, clip_boxes, _filter_boxes, _filter_img_boxes, argsort top-N, nms).
 *   prob_dev    (batch,H,W,8)  f32  rpn_cls_prob_reshape, channel 2a = bg, 2a+1 = fg
 *   pred_dev    (batch,H,W,24) f32  rpn_bbox_pred, channels 6a..6a+5
 *   im_info_dev (batch,3)      f32  [H_bev, W_bev, scale]
 *   calib_dev   (batch,4,12)   f32  rows P2, P3, R0(9)+000, Tr_velo_to_cam
 * Outputs, row capacity cap = mv3d_proposal_3d_capacity(): blob_bv (batch,cap,5),
 * blob_img (batch,cap,5), blob_3d (batch,cap,7), rows >= num_out[b] zero-filled;
 * num_out_dev (batch) i32; status_dev (batch) i32 flag bits as for the NMS (may be NULL). */
typedef struct {
    int32_t feat_stride;     /* 8: lib/networks/MV3D_train.py:5 */
    int32_t pre_nms_topN;    /* cfg[key].RPN_PRE_NMS_TOP_N  (<=0: keep all) */
    int32_t post_nms_topN;   /* cfg[key].RPN_POST_NMS_TOP_N (<=0: keep all) */
    int32_t img_height;      /* 375  hard-coded at proposal_layer_tf.py:147 */
    int32_t img_width;       /* 1242 */
    int32_t img_padding;     /* 50: proposal_layer_tf.py:345 */
    int32_t nms_strict_gt;   /* 0: cpu_nms rule, (double)IoU >= thresh (cfg.USE_GPU_NMS False, the parity target);
                                1: gpu_nms rule, IoU > thresh in f32 (what nms_wrapper.py:19-20 picks when USE_GPU_NMS) */
    int32_t reserved0;
    double nms_thresh;       /* cfg[key].RPN_NMS_THRESH */
    double min_size;         /* cfg[key].RPN_MIN_SIZE   */
} mv3d_proposal_params;

int mv3d_proposal_3d_capacity(int H, int W, const mv3d_proposal_params *p);
size_t mv3d_proposal_3d_workspace_bytes(int batch, int H, int W, const mv3d_proposal_params *p);
int mv3d_proposal_3d(const float *prob_dev, const float *pred_dev, int batch, int H, int W,
                     const float *im_info_dev, const float *calib_dev,
                     const mv3d_proposal_params *p,
                     float *blob_bv_dev, float *blob_img_dev, float *blob_3d_dev,
                     int32_t *num_out_dev, int32_t *status_dev,
                     void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------ RoiPool / RoiPoolGrad
 * Replace lib/roi_pooling_layer/roi_pooling_op_gpu.h:18-27 (ROIPoolForwardLaucher /
 * ROIPoolBackwardLaucher, bodies roi_pooling_op_gpu.cu.cc:20-110,113-215) and the CPU
 * kernels roi_pooling_op.cc:74-190,319-452: the launchers' argument lists with a stream
 * handle where the reference takes `const Eigen::GpuDevice&` and a status instead of
 * exit(-1).  ONE DELIBERATE DEVIATION from roi_pooling_op_gpu.h:18-22: the forward takes
 * `batch_size` after `spatial_scale` (the reference's forward launcher has no such argument;
 * its backward launcher has it in the same place), see below.
 * NHWC f32 data, rois (R,5) [batch_idx,x1,y1,x2,y2], argmax = flat index inside the
 * frame (h*W+w)*C+c or -1; argmax_data may be NULL on forward.  `batch_size` lets the
 * kernel refuse out-of-range batch indices (outputs 0 / -1) instead of reading out of
 * bounds as the reference would.  Backward is the reference's deterministic gather
 * (ROIs ascending, then ph, pw ascending): bit-identical f32 sums, no atomics. */
int mv3d_roi_pool_forward(const float *bottom_data, float spatial_scale, int batch_size,
                          int num_rois, int height, int width, int channels,
                          int pooled_height, int pooled_width, const float *bottom_rois,
                          float *top_data, int32_t *argmax_data, void *stream);
int mv3d_roi_pool_backward(const float *top_diff, float spatial_scale, int batch_size,
                           int num_rois, int height, int width, int channels,
                           int pooled_height, int pooled_width, const float *bottom_rois,
                           float *bottom_diff, const int32_t *argmax_data, void *stream);

/* Several views in one launch (e.g. the BEV and RGB RoiPool layers of one step, MV3D_test.py:95-107):
 * same results as one mv3d_roi_pool_forward per view. */
#define MV3D_MAX_ROI_VIEWS 4
typedef struct {
    const float *bottom_data;    /* (batch_size, height, width, channels) NHWC */
    const float *bottom_rois;    /* (num_rois, 5) */
    float *top_data;             /* (num_rois, pooled_height, pooled_width, channels) */
    int32_t *argmax_data;        /* same shape, may be NULL */
    float spatial_scale;
    int32_t batch_size, num_rois, height, width, channels;
} mv3d_roi_view;
int mv3d_roi_pool_forward_views(int num_views, const mv3d_roi_view *views, int pooled_height, int pooled_width,
                                void *stream);
/* The same call for maps that are NOT cache-resident (inputs written long ago, e.g. a ring of resident batches): prefetch
 * workgroups at the front of the same launch stream the map pixels the ROIs cover into the memory-side cache ahead of the
 * pooling workgroups (58 -> 40 us on the training batch of bench.py).  Same results; pointless -- a few us slower -- when the
 * maps were just produced by the previous layer.  (No counterpart in the reference.) */
int mv3d_roi_pool_forward_views_cold(int num_views, const mv3d_roi_view *views, int pooled_height, int pooled_width,
                                     void *stream);
/* Inference in 16-bit mode: the same pooling with `top_data` written as f16 (top_type 1) or bf16 (top_type 2) -- exactly the values a
 * cast of the f32 output gives (the maximum is taken in f32, then rounded once), without the f32 output and without the cast launch in
 * front of the head's first GEMM.  Every view's argmax_data must be NULL and the shapes must be the XCD-sliced kernels'
 * (C in {256, 512, 1024}, 16-byte aligned buffers): INVALID_ARG otherwise.  (No counterpart in the reference: its op is f32.) */
int mv3d_roi_pool_forward_views_half(int num_views, const mv3d_roi_view *views, int pooled_height, int pooled_width,
                                     int top_type, int cold_maps, void *stream);

/* RoiPoolGrad of several views behind one call (the three RoiPool layers of a training step): same results as
 * one mv3d_roi_pool_backward per view.  In a backward view `top_data` is READ (top_diff, (num_rois,PH,PW,C)),
 * `argmax_data` is read and `bottom_data` is WRITTEN (bottom_diff, (batch_size,H,W,C)); the struct is shared with the
 * forward so that a caller fills it once per layer. */
typedef struct {
    float *bottom_diff;          /* out: (batch_size, height, width, channels) */
    const float *bottom_rois;    /* (num_rois, 5) */
    const float *top_diff;       /* (num_rois, pooled_height, pooled_width, channels) */
    const int32_t *argmax_data;  /* same shape */
    float spatial_scale;
    int32_t batch_size, num_rois, height, width, channels;
} mv3d_roi_grad_view;
size_t mv3d_roi_pool_backward_workspace_bytes(int num_views, const mv3d_roi_grad_view *views, int pooled_height,
                                              int pooled_width);
/* workspace (optional, 256-B aligned): with at least mv3d_roi_pool_backward_workspace_bytes() bytes the call runs as three
 * launches -- per-pixel candidate index (sizes, then lists), then a gather that reads every (roi, bin) record slice on one
 * XCD -- which is the fast path for ANY argmax plane (same C for all views, C in {64, 128, 256, 512}, pooled sizes <= 15,
 * 16-byte aligned buffers).
 * The workspace needs no initialisation (every word is written before it is read).  With workspace == NULL the call needs
 * no scratch memory and is slower (one launch, XCD-sliced, geometry recomputed per slice). */
int mv3d_roi_pool_backward_views(int num_views, const mv3d_roi_grad_view *views, int pooled_height, int pooled_width,
                                 void *workspace, size_t workspace_bytes, void *stream);

/* The PAIR: RoiPool and RoiPoolGrad of a training step with a PRIVATE argmax plane between them (the fast training path).
 * `top_data` and `bottom_diff` are bit-identical to mv3d_roi_pool_forward_views / mv3d_roi_pool_backward_views.
 *   argmax_data     (required) belongs to the pair: the forward writes it, the backward reads it, nobody else needs it (the
 *                   reference's own graph uses the op's second output only in RoiPoolGrad, roi_pooling_op_grad.py:7-43).  The
 *                   pair's kernels store it as ONE-BYTE codes -- the position of the first maximum in its bin's scan order
 *                   (h - hstart) * (wend - wstart) + (w - wstart), 0xFF for the reference's -1 -- in the first quarter of the
 *                   caller's (num_rois, PH, PW, C) int32 buffer (bins of more than 255 pixels, i.e. ROIs far larger than the
 *                   map: 16-bit codes in the two quarters behind it): 3 / 8 of the record bytes (8 -> 5 B per pooled value)
 *                   are neither written by the forward nor read by the backward.  The LAST quarter of view 0's buffer carries the
 *                   backward's work list (a few KB): one planning workgroup at the front of the forward launch estimates every map
 *                   tile's entry stream from the ROIs, cuts the tiles under long streams into four sub-tiles (first in the list: a
 *                   stream is one wave's serial work and the launch's makespan) and flags the tiles no ROI touches (written as
 *                   zeros without a ROI filter).  Both launches decide from the shapes alone whether the list exists (<= 3072 tiles
 *                   in all views, the list fits the quarter; otherwise the static tile grid).  mv3d_roi_pool_argmax_decode returns
 *                   the reference's int32 plane (tests, verification).
 *   backward        ONE launch (csrc/roi_grad_tiles.hip): a wave owns a unit of the work list -- a tile of <= 16 map pixels x 64
 *                   channels in LDS, or a sub-tile of a cut one (a single pixel keeps its sum in a register) --, finds the ROIs that
 *                   reach it, streams their (roi, bin) records in the reference's order (roi_pooling_op.cc:392-431) and routes every
 *                   value to the pixel its code names; a unit is written out whole (no index, no fill, no scratch memory).  The
 *                   views must be the forward's (same order, shapes, scale, ROIs) with the argmax buffers it wrote.  (For a
 *                   foreign argmax plane use mv3d_roi_pool_backward_views: int32 argmax.)
 *   workspace       optional and unused by the pair's own kernels since round 6 (the one-launch backward needs no scratch memory;
 *                   measured against the round-5 index + gather structure behind a workspace: level alone, 3 - 7 % more frames/s
 *                   with eight batches in flight); checked for alignment, handed on to mv3d_roi_pool_backward_views for shapes
 *                   outside the pair's kernels.
 *   cold_maps       != 0: as mv3d_roi_pool_forward_views_cold.
 * Shapes outside the pair's kernels (C not in {256, 512} / not the same for all views, pooled sizes > 15, a map of more than 65534
 * pixels, an empty view) take the plain forward (int32 argmax) and mv3d_roi_pool_backward_views behind the same entries.
 * (Measured and dropped, round 5: the candidate index built by workgroups inside the FORWARD launch, and a single-pass index with
 * an in-launch look-back -- profiles/EXPERIMENTS.md, round 5.) */
size_t mv3d_roi_pool_pair_workspace_bytes(int num_views, const mv3d_roi_grad_view *views, int pooled_height, int pooled_width);
int mv3d_roi_pool_forward_views_pair(int num_views, const mv3d_roi_view *views, int pooled_height, int pooled_width,
                                     int cold_maps, void *stream);
int mv3d_roi_pool_backward_views_pair(int num_views, const mv3d_roi_grad_view *views, int pooled_height, int pooled_width,
                                      void *workspace, size_t workspace_bytes, void *stream);
/* views: as passed to mv3d_roi_pool_forward_views_pair (after it ran); argmax_out[k]: (num_rois, PH, PW, C) int32, a buffer of its
 * own: the reference's argmax plane of view k (flat index inside the frame, -1 for an empty bin). */
int mv3d_roi_pool_argmax_decode(int num_views, const mv3d_roi_view *views, int pooled_height, int pooled_width,
                                int32_t *const *argmax_out, void *stream);

/* Call-compatible aliases of the reference's two launchers (roi_pooling_op_gpu.h:18-27): EXACTLY their argument order -- the
 * forward WITHOUT a batch size (out-of-range batch indices are then the caller's problem, as in the reference) -- with `void *stream`
 * where the reference passes `const Eigen::GpuDevice &d` (a maintainer's call site passes `d.stream()`), returning `bool`-like
 * 1 on success and 0 on failure where the reference returns d.ok().  For a port of roi_pooling_op.cc:250-287 / :505-554 that
 * keeps its call sites untouched. */
int mv3d_ROIPoolForwardLaucher(const float *bottom_data, float spatial_scale, int num_rois, int height, int width, int channels,
                               int pooled_height, int pooled_width, const float *bottom_rois, float *top_data,
                               int32_t *argmax_data, void *stream);
int mv3d_ROIPoolBackwardLaucher(const float *top_diff, float spatial_scale, int batch_size, int num_rois, int height, int width,
                                int channels, int pooled_height, int pooled_width, const float *bottom_rois,
                                float *bottom_diff, const int32_t *argmax_data, void *stream);

/* ------------------------------------------------------------------ third (front-view) ROI
 * rois_3d_dev (R,7) [b,x,y,z,l,w,h] -> rois_fv_dev (R,5) [b,x1,y1,x2,y2] on the 64 x 512 cylindrical front-view map
 * of the MV3D paper (azimuth [-45,+45] deg -> 512 columns, elevation [-24.9,+2] deg -> 64 rows; min/max over the 8
 * corners, clipped to the map).  Fills the hook the reference leaves empty: `proposal_transform` returns None for
 * anything but 'bv' / 'img' (lib/networks/network.py:293-315).  PARITY UNPINNED (no reference code); the definition
 * is restated in oracle/mv3d_oracle.c and both agree bit for bit. */
int mv3d_rois_3d_to_fv(const float *rois_3d_dev, int num_rois, float *rois_fv_dev, void *stream);

/* ------------------------------------------------------------------ anchor_target_layer
 * Replaces the deterministic part of lib/rpn_msr/anchor_target_layer_tf.py:21-250 plus
 * lib/utils/bbox.pyx:15-55 and lib/fast_rcnn/bbox_transform.py:32-58 (one frame):
 * stage1 = inside filter, f64 IoU vs gt_boxes_bv, argmax / max, gt-argmax flood, labels
 * before any random subsampling, 6-d targets; it also compacts the three candidate
 * lists the reference subsamples from.  The draws themselves come from the caller's
 * numpy RNG (draw-for-draw parity needs the host's MT19937 stream): the caller reads
 * counts_dev, draws the permutations, and stage2 applies them.
 *   gt_bv_dev (G,5) f32, gt_3d_dev (G,7) f32, im_info_dev (3) f32.
 *   labels_dev (N) f32 in {-1,0,1}; targets_dev (N,6) f32; N = H*W*4, anchor order (h,w,a).
 *   counts_dev[8] i32: [0] n_inside, [1] n_fg (labels==1 before subsampling),
 *     [2] n_bg (labels==0 before subsampling), [3] n_low (max_overlap < NEGATIVE_OVERLAP
 *     among inside anchors), rest reserved.
 *   fg_hi_dev (N) u8: for the k-th fg candidate, 1 iff its max_overlap >= NEGATIVE_OVERLAP
 *     (lets the host know how many positives survive step 9 of SURVEY A.1). */
typedef struct {
    int32_t feat_stride;
    int32_t clobber_positives;   /* cfg.TRAIN.RPN_CLOBBER_POSITIVES */
    double negative_overlap;     /* cfg.TRAIN.RPN_NEGATIVE_OVERLAP  */
    double positive_overlap;     /* cfg.TRAIN.RPN_POSITIVE_OVERLAP  */
} mv3d_anchor_target_params;

/* The workspace size grows with G; nothing in it has to be initialised (stage 1 = three launches, no memset). */
size_t mv3d_anchor_target_workspace_bytes(int H, int W, int G);
int mv3d_anchor_target_stage1(int H, int W, const float *im_info_dev, const float *gt_bv_dev,
                              const float *gt_3d_dev, int G, const mv3d_anchor_target_params *p,
                              float *labels_dev, float *targets_dev, int32_t *counts_dev,
                              uint8_t *fg_hi_dev, void *workspace, size_t workspace_bytes,
                              void *stream);
/* stage2: disable_fg_dev[n_dis_fg] = positions in the fg candidate list to set to -1,
 * disable_bg1_dev[n_dis_bg1] positions in the first bg list; then the debug outputs
 * anchors_dev (<=cap,5) / anchors_3d_dev (<=cap,7) / n_anchors_dev are taken (labels != -1),
 * labels[max_overlap < NEGATIVE_OVERLAP] = 0, and disable_bg2_dev[n_dis_bg2] positions in
 * the second bg list are set to -1 (anchor_target_layer_tf.py:146-183). */
int mv3d_anchor_target_stage2(int H, int W, const mv3d_anchor_target_params *p,
                              const int32_t *disable_fg_dev, int n_dis_fg,
                              const int32_t *disable_bg1_dev, int n_dis_bg1,
                              const int32_t *disable_bg2_dev, int n_dis_bg2,
                              float *labels_dev, float *anchors_dev, float *anchors_3d_dev,
                              int32_t *n_anchors_dev, int anchors_cap,
                              void *workspace, size_t workspace_bytes, void *stream);

/* Frames of a batch behind one launch of every kernel (the reference's layer is single-frame, lib/rpn_msr/
 * anchor_target_layer_tf.py:42-45; the per-frame entry points above are these with batch = 1).  Host-side arrays of `batch`
 * entries (device pointers / counts); labels_dev (batch,N), targets_dev (batch,N,6), im_info_dev (batch,3),
 * anchors_dev (batch,cap,5), anchors_3d_dev (batch,cap,7), n_anchors_dev (batch) are contiguous over the frames; one
 * workspace per frame, each of at least workspace_bytes = mv3d_anchor_target_workspace_bytes(H, W, max_b G[b]) bytes.
 * batch <= 16.  stage2 = two launches: the three disable lists, then debug rows + final labels.
 * ONE-SHOT CONTRACT: stage 2 CONSUMES the workspace stage 1 left (the second background draw marks its anchors in the
 * workspace's argmax array, the list look-back words are only cleared by stage 1's overlap launch): exactly one stage-2 call
 * per stage-1 call on a workspace; to apply other disable lists, run stage 1 again.  (Same for the per-frame entries.) */
int mv3d_anchor_target_stage1_batch(int batch, int H, int W, const float *im_info_dev, const float *const *gt_bv_dev,
                                    const float *const *gt_3d_dev, const int *G, const mv3d_anchor_target_params *p,
                                    float *labels_dev, float *targets_dev, int32_t *const *counts_dev,
                                    uint8_t *const *fg_hi_dev, void *const *workspace, size_t workspace_bytes, void *stream);
int mv3d_anchor_target_stage2_batch(int batch, int H, int W, const mv3d_anchor_target_params *p,
                                    const int32_t *const *disable_fg_dev, const int *n_dis_fg,
                                    const int32_t *const *disable_bg1_dev, const int *n_dis_bg1,
                                    const int32_t *const *disable_bg2_dev, const int *n_dis_bg2, float *labels_dev,
                                    float *anchors_dev, float *anchors_3d_dev, int32_t *n_anchors_dev, int anchors_cap,
                                    void *const *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------ proposal_target_layer_3d
 * Replaces lib/rpn_msr/proposal_target_layer_tf.py:19-94 with _sample_rois_3d (:227-298),
 * _compute_targets_cnr (:211-225), _get_bbox_regression_labels_3d (:172-194) and the image
 * projection (one frame).  stage1: candidate set = proposals followed by the GT boxes, f64 IoU
 * vs gt_boxes_bv, first-argmax / max, ordered fg / bg candidate lists; counts_dev[4] =
 * [n_candidates, n_fg (max_ov >= FG_THRESH), n_bg (LO <= max_ov < HI), 0].  The caller draws
 * npr.permutation(n_fg)[:k_fg] and npr.permutation(n_bg)[:k_bg] (legacy RandomState.choice
 * without replacement) and stage2 gathers the sampled ROIs: rois_bv (S,5), rois_img (S,5),
 * labels (S) i32, bbox_targets (S, 24*num_classes), rois_3d (S,7), S = n_fg + n_bg rows, fg first. */
typedef struct {
    int32_t num_classes;         /* n_classes = 2: lib/networks/MV3D_train.py:4 */
    int32_t frame_index;         /* batch column written for the appended ground-truth rows: 0 = the reference
                                    (single-frame batches, :38-44); a batched caller passes the frame's index */
    double fg_thresh;            /* cfg.TRAIN.FG_THRESH    */
    double bg_thresh_hi;         /* cfg.TRAIN.BG_THRESH_HI */
    double bg_thresh_lo;         /* cfg.TRAIN.BG_THRESH_LO */
} mv3d_proposal_target_params;

size_t mv3d_proposal_target_workspace_bytes(int num_rois, int G);
int mv3d_proposal_target_stage1(const float *rois_bv_dev, const float *rois_3d_dev, int num_rois,
                                const float *gt_bv_dev, const float *gt_3d_dev, int G,
                                const mv3d_proposal_target_params *p, int32_t *counts_dev,
                                void *workspace, size_t workspace_bytes, void *stream);
int mv3d_proposal_target_stage2(const float *rois_bv_dev, const float *rois_3d_dev, int num_rois,
                                const float *gt_bv_dev, const float *gt_3d_dev,
                                const float *gt_corners_dev, int G, const float *calib_dev,
                                const mv3d_proposal_target_params *p,
                                const int32_t *fg_pick_dev, int n_fg, const int32_t *bg_pick_dev, int n_bg,
                                float *rois_bv_out, float *rois_img_out, int32_t *labels_out,
                                float *bbox_targets_out, float *rois_3d_out,
                                void *workspace, size_t workspace_bytes, void *stream);

/* The same for `batch` frames behind one launch of every kernel (host-side arrays of per-frame device pointers / counts /
 * parameter structs -- params[b].frame_index = the batch column of frame b's appended ground-truth rows); batch <= 16.
 * stage 2: rois_fv_out (may be NULL, as may its entries) = per-frame (S_b,5) buffers for the third view's ROIs of the sampled
 * boxes, the values mv3d_rois_3d_to_fv gives for rois_3d_out (one launch less on the training path). */
int mv3d_proposal_target_stage1_batch(int batch, const float *const *rois_bv_dev, const float *const *rois_3d_dev,
                                      const int *num_rois, const float *const *gt_bv_dev, const float *const *gt_3d_dev,
                                      const int *G, const mv3d_proposal_target_params *params, int32_t *const *counts_dev,
                                      void *const *workspace, const size_t *workspace_bytes, void *stream);
int mv3d_proposal_target_stage2_batch(int batch, const float *const *rois_bv_dev, const float *const *rois_3d_dev,
                                      const int *num_rois, const float *const *gt_bv_dev, const float *const *gt_3d_dev,
                                      const float *const *gt_corners_dev, const int *G, const float *const *calib_dev,
                                      const mv3d_proposal_target_params *params, const int32_t *const *fg_pick_dev,
                                      const int *n_fg, const int32_t *const *bg_pick_dev, const int *n_bg,
                                      float *const *rois_bv_out, float *const *rois_img_out, int32_t *const *labels_out,
                                      float *const *bbox_targets_out, float *const *rois_3d_out, float *const *rois_fv_out,
                                      void *const *workspace, const size_t *workspace_bytes, void *stream);
/* The batched entries for proposals whose NUMBER stays on the device (the `num_out_dev` of mv3d_proposal_3d): num_rois_cap[b]
 * = the capacity of frame b's proposal blobs (sizes the workspace: mv3d_proposal_target_workspace_bytes(num_rois_cap[b],
 * G[b])), num_rois_dev[b] -> the frame's count, read by the kernels (clamped to [0, cap]).  A training step then needs ONE
 * host round trip -- the counts of the candidate lists the caller draws from -- instead of two. */
int mv3d_proposal_target_stage1_batch_devn(int batch, const float *const *rois_bv_dev, const float *const *rois_3d_dev,
                                           const int *num_rois_cap, const int32_t *const *num_rois_dev,
                                           const float *const *gt_bv_dev, const float *const *gt_3d_dev, const int *G,
                                           const mv3d_proposal_target_params *params, int32_t *const *counts_dev,
                                           void *const *workspace, const size_t *workspace_bytes, void *stream);
int mv3d_proposal_target_stage2_batch_devn(int batch, const float *const *rois_bv_dev, const float *const *rois_3d_dev,
                                           const int *num_rois_cap, const int32_t *const *num_rois_dev,
                                           const float *const *gt_bv_dev, const float *const *gt_3d_dev,
                                           const float *const *gt_corners_dev, const int *G, const float *const *calib_dev,
                                           const mv3d_proposal_target_params *params, const int32_t *const *fg_pick_dev,
                                           const int *n_fg, const int32_t *const *bg_pick_dev, const int *n_bg,
                                           float *const *rois_bv_out, float *const *rois_img_out, int32_t *const *labels_out,
                                           float *const *bbox_targets_out, float *const *rois_3d_out, float *const *rois_fv_out,
                                           void *const *workspace, const size_t *workspace_bytes, void *stream);

/* ------------------------------------------------------------------ SURVEY §8(f) "next" rows
 * BEV rasteriser: replaces point_cloud_2_top (lib/utils/read_lidar.py:10-115, called with
 * res=0.1, zres=0.3, side_range=(-30,30), fwd_range=(0,60), height_range=(-2,0.4):
 * tools/read_lidar.py:121-133).  points_dev (P,4) f32 [x,y,z,reflectance], 16-byte aligned;
 * top_dev (601,601,9) f32: channels 0-7 = z + 2 of the last point of each height slice, channel
 * 8 = reflectance of the last point of the highest slice (numpy fancy-assignment order). */
int mv3d_point_cloud_2_top(const float *points_dev, int num_points, float *top_dev, void *stream);

/* The same function with its own parameters (lib/utils/read_lidar.py:10-16: res, zres, side_range, fwd_range, height_range).
 * mv3d_point_cloud_2_top_shape: dims[3] = (y_max + 1, x_max + 1, z_max + 1) of the map (read_lidar.py:49-52), host arithmetic only.
 * mv3d_point_cloud_2_top_ranges: top_dev of that shape; *status_dev (one int32 on the device, written by the call) becomes 1 when a
 * point's cell falls outside the map after numpy's wrap of negative indices -- where the reference's fancy assignment raises
 * IndexError (the point is skipped here; the host mirror raises).  MV3D_ERR_INVALID_ARG: empty or inverted ranges, more than
 * 32767 cells per axis or more than 31 height slices, P >= 2^27. */
int mv3d_point_cloud_2_top_shape(double res, double zres, double side_lo, double side_hi, double fwd_lo, double fwd_hi,
                                 double height_lo, double height_hi, int *dims);
int mv3d_point_cloud_2_top_ranges(const float *points_dev, int num_points, double res, double zres, double side_lo, double side_hi,
                                  double fwd_lo, double fwd_hi, double height_lo, double height_hi, float *top_dev,
                                  int *status_dev, void *stream);

/* ------------------------------------------------------------------ the target layers' random subsamplings (HOST code)
 * The reference subsamples with `npr.choice(inds, size=k, replace=False)` on the numpy GLOBAL legacy RandomState
 * (lib/rpn_msr/anchor_target_layer_tf.py:146-159,178-183; lib/rpn_msr/proposal_target_layer_tf.py:246-269) = 
 * `inds[permutation(len(inds))[:k]]`; the stage-2 entries above take those permutations as index lists.  These two host
 * functions draw them in C directly on the memory of numpy's generator (`mt19937_state` = the address numpy publishes as
 * `np.random.mtrand._rand._bit_generator.ctypes.state_address`: {uint32 key[624]; int32 pos}), restating numpy's
 * RandomState.permutation -> shuffle -> random_interval -> MT19937 genrand loop (numpy is a dependency of the reference that
 * /root/reference does not vendor; pinned against numpy itself in tests/test_legacy_rng.py).  The caller holds the
 * generator's lock (`bit_generator.lock`); no device work, no stream. */
/* out[0 .. n) = RandomState.permutation(n) */
int mv3d_legacy_permutation(void *mt19937_state, int32_t n, int32_t *out);
typedef struct {
    int32_t n_fg, n_bg, n_low;     /* anchor-target stage 1 counts: foreground / background / low-overlap candidates */
    int32_t pt_n_fg, pt_n_bg;      /* proposal-target stage 1 counts */
    int32_t reserved0;
    const uint8_t *fg_alive;       /* (n_fg) host bytes: foreground candidate i still positive after the relabel (:176) */
} mv3d_draw_frame;
typedef struct {
    int32_t rpn_batchsize;         /* cfg.TRAIN.RPN_BATCHSIZE */
    int32_t rpn_num_fg;            /* int(cfg.TRAIN.RPN_FG_FRACTION * RPN_BATCHSIZE) */
    int32_t rois_per_image;        /* cfg.TRAIN.BATCH_SIZE / images per batch (1) */
    int32_t roi_fg_max;            /* np.round(cfg.TRAIN.FG_FRACTION * rois_per_image) */
} mv3d_draw_params;
/* All draws of a batch, frame by frame, anchor targets before proposal targets (the order the numpy-contract layers draw in):
 * per frame five lists -- anchors to disable: surplus foreground, surplus background, surplus background after the relabel;
 * ROIs to keep: foreground picks, background picks -- appended to `lists` (host, e.g. pinned), sizes[5 * b + k] = their
 * lengths.  scratch: >= the largest candidate count, in int32 words.  MV3D_ERR_WORKSPACE if `lists` / `scratch` are too small. */
int mv3d_draw_training_subsamples(void *mt19937_state, int batch, const mv3d_draw_frame *frames, const mv3d_draw_params *par,
                                  int32_t *lists, size_t lists_cap, int32_t *sizes, int32_t *scratch, size_t scratch_cap);

/* ------------------------------------------------------------------ the training path on fresh frames, driven from C
 * One object runs what lib/networks/MV3D_train.py:83-112 wires behind the RPN heads for every training batch --
 *   proposal_layer_3d (TRAIN cfg)      lib/rpn_msr/proposal_layer_tf.py:25-202       = mv3d_proposal_3d
 *   anchor_target_layer                lib/rpn_msr/anchor_target_layer_tf.py:21-250  = mv3d_anchor_target_stage1/2_batch
 *   proposal_target_layer_3d           lib/rpn_msr/proposal_target_layer_tf.py:19-94 = mv3d_proposal_target_stage1/2_batch_devn
 * -- with the reference's host-side subsampling draws (mv3d_draw_training_subsamples on numpy's global generator) in between:
 *   submit()  enqueues every launch up to the candidate lists on `stream`, the copies of the counts to the slot's pinned host
 *             buffers and an event; returns at once.
 *   a helper thread owned by the object (created with helper_thread = 1) waits for that event, draws -- slots strictly in
 *             submission order, so the generator's stream is the reference's: frame by frame, anchor targets before proposal
 *             targets -- uploads the index lists and enqueues stage 2 on the same stream.
 *   finish()  waits until the slot's stage 2 is ENQUEUED (not executed) and reports the frames' row counts; the outputs are in the
 *             slot's buffers, ordered on the stream given to submit().  With helper_thread = 0 finish() does the helper's work itself.
 * A caller keeps up to `depth` batches in flight (submit batch i + 1 before finish of batch i): the host's draws of one batch then run
 * while the device works on its neighbours, and a batch costs its caller two calls.  Between submit() and finish() of a slot the
 * generator behind `mt19937_state` belongs to the object.  All buffers are the caller's (layouts: see the per-entry comments above;
 * B = batch <= 16, N = 4 H W anchors, cap = mv3d_proposal_3d_capacity): the object allocates only events and its thread. */
typedef struct {
    int32_t batch, H, W, num_classes;
    int32_t proposal_cap;          /* mv3d_proposal_3d_capacity(H, W, &proposal): rows per frame of the proposal blobs */
    int32_t anchor_cap;            /* rows per frame of anchors / anchors_3d */
    int32_t roi_cap;               /* rows per frame of the sampled-ROI outputs (cfg.TRAIN.BATCH_SIZE) */
    int32_t max_gt;                /* ground-truth boxes per frame the workspaces were sized for */
    mv3d_proposal_params proposal;
    mv3d_anchor_target_params anchor;
    mv3d_proposal_target_params target;      /* thresholds (frame_index is set per frame by the object) */
    mv3d_draw_params draw;
} mv3d_train_path_config;
typedef struct {
    /* device */
    float *blob_bv, *blob_img, *blob_3d;     /* proposals (B,cap,5) (B,cap,5) (B,cap,7) */
    int32_t *num_proposals;                  /* (2 B): the frames' proposal counts, then mv3d_proposal_3d's status words */
    void *proposal_ws; size_t proposal_ws_bytes;
    float *rpn_labels, *rpn_targets;         /* (B,N) (B,N,6) */
    float *anchors, *anchors_3d;             /* (B,anchor_cap,5) (B,anchor_cap,7) */
    int32_t *n_anchors;                      /* (B) */
    uint8_t *report; size_t report_row;      /* (B,report_row): per frame [counts 32 B | foreground flags N B], report_row % 256 == 0 */
    int32_t *pt_counts;                      /* (B,4) */
    void *anchor_ws; size_t anchor_ws_bytes; /* B workspaces back to back, each anchor_ws_bytes (a multiple of 256) */
    void *target_ws; size_t target_ws_bytes; /* the same for the proposal-target workspaces */
    float *rois_bev, *rois_rgb, *rois_fv;    /* (B roi_cap,5) each, frame b's rows behind frame b - 1's; rois_fv may be NULL */
    float *rois_3d; int32_t *labels; float *bbox_targets;    /* (B roi_cap,7) (B roi_cap) (B roi_cap, 24 num_classes) */
    int32_t *lists; size_t lists_cap;        /* the index lists of a batch, int32 words: >= B (3 N + 2 (cap + max_gt)) */
    /* host (pinned, except h_scratch) */
    uint8_t *h_report; size_t h_report_row;  /* (B,h_report_row): the first h_report_row <= report_row bytes of every report row */
    int32_t *h_pt_counts;                    /* (B,4) */
    int32_t *h_num_proposals;                /* (2 B) */
    int32_t *h_lists;                        /* lists_cap words */
    int32_t *h_scratch; size_t scratch_cap;  /* >= max(N, cap + max_gt) words: one permutation at its largest */
} mv3d_train_path_slot;
typedef struct mv3d_train_path mv3d_train_path;
int mv3d_train_path_create(const mv3d_train_path_config *config, int depth, const mv3d_train_path_slot *slots, int helper_thread,
                           mv3d_train_path **out);
/* replace the parameter structs (thresholds, NMS settings, draw sizes; the geometry must be unchanged); no slot may be in flight */
int mv3d_train_path_configure(mv3d_train_path *path, const mv3d_train_path_config *config);
/* prob (B,H,W,8), pred (B,H,W,24), im_info (B,3), calib (B,4,12): device f32; gt_*: host arrays of B device pointers -- gt_bv (G,5),
 * gt_3d (G,7), gt_corners (G,25) -- and G (B) with 1 <= G[b] <= max_gt.  The inputs must stay valid until finish() has returned. */
int mv3d_train_path_submit(mv3d_train_path *path, int slot, const float *prob_dev, const float *pred_dev, const float *im_info_dev,
                           const float *calib_dev, const float *const *gt_bv_dev, const float *const *gt_3d_dev,
                           const float *const *gt_corners_dev, const int *G, void *mt19937_state, void *stream);
/* rows_out (B): sampled ROIs per frame (frame b's rows start at the sum of the frames before it); num_proposals_out (B); sizes_out
 * (5 B, may be NULL): the lengths of the drawn lists as mv3d_draw_training_subsamples reports them.  Returns the first error of the
 * slot's host stage (MV3D_ERR_ZERO_DIVISION where the reference's NMS raises); the slot is free again either way. */
int mv3d_train_path_finish(mv3d_train_path *path, int slot, int32_t *rows_out, int32_t *num_proposals_out, int32_t *sizes_out);
/* host seconds the helper spent waiting for stage 1 / drawing since the last call (diagnostics); either pointer may be NULL */
int mv3d_train_path_host_seconds(mv3d_train_path *path, double *wait_s, double *draw_s);
void mv3d_train_path_destroy(mv3d_train_path *path);     /* waits for the slots in flight, joins the thread */

/* KITTI label rows -> ground-truth encodings (lib/datasets/kitti_mv3d.py:240-272 with computeCorners3D, camera_to_lidar_cnr,
 * lidar_cnr_to_3d, lidar_3d_to_bv of lib/utils/transform.py:441-465,502-524,172-187,113-142), one thread per labelled object:
 *   box_cam_dev (G,6) f32 [tx,ty,tz,l,w,h] (label columns 11-13, 10, 9, 8), cos_sin_dev (G,2) f64 = cos / sin of rotation_y,
 *   inv_rot_dev (9) f32 = inverse of the 3x3 rotation part of Tr_velo_to_cam, tr_velo_to_cam_dev (12) f32
 *   -> corners_cam_dev (G,24), corners_lidar_dev (G,24) [x0..7,y0..7,z0..7], boxes_3d_dev (G,6) LIDAR box, boxes_bv_dev (G,4)
 *   BEV pixel box, all f32 as the roidb stores them.  (cos / sin and the 3x3 inverse are numpy / LAPACK calls on a handful of
 *   host scalars in the reference and stay on the host next to the text parsing.) */
int mv3d_gt_encode(const float *box_cam_dev, const double *cos_sin_dev, int num_objects, const float *inv_rot_dev,
                   const float *tr_velo_to_cam_dev, float *corners_cam_dev, float *corners_lidar_dev,
                   float *boxes_3d_dev, float *boxes_bv_dev, void *stream);

/* Test-time tail of box_detect (lib/fast_rcnn/test_mv.py:240-261): rois_3d_dev (R,7) = rois[2],
 * bbox_pred_dev (R,24*nc) -> corners (R,24) [lidar_3d_to_corners], pred_cnr_r (R,24*nc)
 * [bbox_transform_inv_cnr, lib/fast_rcnn/bbox_transform.py:157-176], pred_bv / pred_bv_r (R,4*nc)
 * [corners_to_bv, lib/utils/transform.py:342-366, of the unregressed / regressed corners]. */
int mv3d_box_detect_tail(const float *rois_3d_dev, const float *bbox_pred_dev, int num_rois, int num_classes,
                         float *corners_dev, float *pred_cnr_r_dev, float *pred_bv_dev, float *pred_bv_r_dev,
                         void *stream);

/* Training losses and their gradients (lib/fast_rcnn/train_mv.py:74-136), fused: losses_dev[0] = mean softmax
 * cross-entropy, losses_dev[1] = mean over rows of sum smooth-L1 (sigma = 3 in the reference) of pred - target;
 * d_*_dev (may be NULL) receive d loss / d logits and d loss / d pred.  A mean over no rows is NaN, as in TF.
 * RPN (:92-113): rpn_cls_score (N,2) = rpn_cls_score_reshape, labels f32 in {-1,0,1} = rpn_data[0]; cross-entropy
 * over label != -1, box loss over label == 1, rpn_bbox_pred / targets (N,6).
 * RCNN (:115-127): cls_score (S,K), labels i32 = roi_data_3d[2], bbox_pred / targets (S, box_dim = 24 K); all rows. */
size_t mv3d_loss_workspace_bytes(int rows);
int mv3d_rpn_loss(const float *rpn_cls_score_dev, const float *rpn_labels_dev, const float *rpn_bbox_pred_dev,
                  const float *rpn_bbox_targets_dev, int num_anchors, float sigma, float *losses_dev,
                  float *d_cls_score_dev, float *d_bbox_pred_dev, void *workspace, size_t workspace_bytes, void *stream);
int mv3d_rcnn_loss(const float *cls_score_dev, const int32_t *labels_dev, const float *bbox_pred_dev,
                   const float *bbox_targets_dev, int num_rois, int num_classes, int box_dim, float sigma,
                   float *losses_dev, float *d_cls_score_dev, float *d_bbox_pred_dev, void *workspace,
                   size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------ the VGG16 trunk's contraction (north_star "MFMA only for the
 * VGG16 conv backbone"): Network.conv(3, 3, c_o, 1, 1) [3x3, stride 1, SAME, + bias, optional ReLU] and
 * Network.max_pool(2, 2, 2, 2, 'VALID') of lib/networks/network.py:109-133,182-189 for the layers of
 * lib/networks/MV3D_train.py:44-81, as an implicit GEMM on v_mfma_f32_32x32x16_f16 (f16 operands, f32 accumulate).
 * This is the SERVING trunk (BASELINE configs[4]: fp16), lower precision than the reference's fp32 graph and not part of the
 * bit-exact contract; tolerance vs an fp32 convolution of the same f16-rounded operands: 2e-3 relative to the map's max.
 *   x_framed  (batch, height + 2, width + 2, c_in)  f16 NHWC with a one-pixel ZERO frame (the SAME padding, materialised)
 *   w_packed  (c_out, 9 * c_in) f16, k = (ky * 3 + kx) * c_in + c   [TF's HWIO filter transposed to (O, H, W, I)]
 *   bias      (c_out) f32
 *   y         out_framed ? (batch, height + 2, width + 2, c_out) interior only (the frame is the owner's, zeroed once)
 *                        : (batch, height, width, c_out);  f32 if out_f32 (the map a RoiPool layer reads) else f16
 * c_in and c_out multiples of 64, or c_in == 16 = the input layer (conv1_1: 9 / 3 channels zero-padded to 16), whose weights are
 * packed (c_out, 192): k = 16 * tap + c for the 9 taps, zero for tap slots 9..11.  Every buffer < 2 GiB (32-bit buffer
 * offsets: split the batch above that) and 16-byte aligned; MV3D_ERR_INVALID_ARG otherwise, before any launch. */
int mv3d_conv3x3_f16(const void *x_framed, const void *w_packed, const float *bias, void *y, int batch, int height, int width,
                     int c_in, int c_out, int out_framed, int out_f32, int relu, void *stream);
/* The same convolution in the REFERENCE'S precision: f32 framed maps (c_in a multiple of 32; the input layer's 9 / 3 channels padded to
 * 32), f32 packed weights (c_out, 9 * c_in), f32 output (framed or not), exact f32 products and sums on v_mfma_f32_32x32x2_f32 -- only
 * the summation order differs from any other fp32 convolution.  MFMA-bound at the f32 matrix rate (157 TFLOP/s peak). */
int mv3d_conv3x3_f32(const void *x_framed, const void *w_packed, const float *bias, void *y, int batch, int height, int width, int c_in,
                     int c_out, int out_framed, int relu, void *stream);
int mv3d_maxpool2x2_f32(const void *x_framed, void *y_framed, int batch, int height, int width, int channels, void *stream);
int mv3d_frame_nhwc_f32(const float *x_nhwc, void *y_framed, int batch, int height, int width, int channels, int channels_out,
                        void *stream);
/* The same three entries on bfloat16 activations / weights (v_mfma_f32_32x32x16_bf16, f32 accumulate): the TRAINING trunk's
 * forward and data-gradient convolutions (mv3d_tf_amd/trunk_train.py) -- f16's 5-bit exponent would need loss scaling. */
int mv3d_conv3x3_bf16(const void *x_framed, const void *w_packed, const float *bias, void *y, int batch, int height, int width,
                      int c_in, int c_out, int out_framed, int out_f32, int relu, void *stream);
/* y = (conv3x3(x) + bias) where gate > 0, else 0: the data-gradient step of the training trunk in one launch (x = the framed gradient
 * of the layer above, w = that layer's filter flipped with its channel axes swapped, gate_framed = this layer's framed ReLU
 * output (batch, height + 2, width + 2, c_out) bf16); framed bf16 output, no ReLU of its own. */
int mv3d_conv3x3_gated_bf16(const void *x_framed, const void *w_packed, const float *bias, const void *gate_framed, void *y,
                            int batch, int height, int width, int c_in, int c_out, void *stream);
int mv3d_maxpool2x2_bf16(const void *x_framed, void *y_framed, int batch, int height, int width, int channels, void *stream);
int mv3d_frame_nhwc_bf16(const float *x_nhwc, void *y_framed, int batch, int height, int width, int channels, int channels_out,
                         void *stream);
/* Gradient of ReLU + max_pool(2, 2, 2, 2, 'VALID') for the training trunk: y_framed = the pre-pool map (a ReLU output),
 * g_pooled_framed = gradient w.r.t. the pooled map -> gy_framed = gradient w.r.t. y's pre-activation (first maximum of the
 * window gets the gradient if it is > 0).  All framed bf16; rows / columns the pool dropped are left untouched (zero). */
int mv3d_maxpool2x2_bwd_bf16(const void *y_framed, const void *g_pooled_framed, void *gy_framed, int batch, int height, int width,
                             int channels, void *stream);
int mv3d_maxpool2x2_bwd_f32(const void *y_framed, const void *g_pooled_framed, void *gy_framed, int batch, int height, int width,
                            int channels, void *stream);      /* the same on f32 maps (the fp32 training trunk) */
/* Several views of ONE layer shape behind one launch (the BEV / image / front-view trunks of lib/networks/MV3D_train.py:44-81 at one
 * VGG depth have the same channel counts and different map sizes): the grid is the concatenation of the views' tiles, so that
 * the small maps of a training batch fill the chip together, on ONE stream.  Same arithmetic per view as the single-view
 * entries above (which are these with num_views = 1).  gate_framed: optional (bf16 / f16, framed 16-bit output only; all views
 * or none) = the ReLU gate of mv3d_conv3x3_gated_bf16; the f32 entry takes an f32 gate (framed f32 output). */
#define MV3D_MAX_CONV_VIEWS 3
typedef struct {
    const void *x_framed;        /* (batch, height + 2, width + 2, c_in) */
    const void *w_packed;        /* (c_out, 9 * c_in), this view's filter */
    const float *bias;           /* (c_out) */
    const void *gate_framed;     /* optional, (batch, height + 2, width + 2, c_out) */
    void *y;                     /* framed (batch, height + 2, width + 2, c_out) or bare (batch, height, width, c_out) */
    int32_t batch, height, width, reserved0;
} mv3d_conv_view;
int mv3d_conv3x3_views_f16(int num_views, const mv3d_conv_view *views, int c_in, int c_out, int out_framed, int out_f32, int relu,
                           void *stream);
int mv3d_conv3x3_views_bf16(int num_views, const mv3d_conv_view *views, int c_in, int c_out, int out_framed, int out_f32, int relu,
                            void *stream);
int mv3d_conv3x3_views_f32(int num_views, const mv3d_conv_view *views, int c_in, int c_out, int out_framed, int relu, void *stream);
/* Convolution + bias + ReLU + the 2x2 / stride 2 VALID max pool that follows it (lib/networks/network.py:182-189) in ONE launch, for
 * layers whose full-size output only the pool reads (the serving graph's conv1_2 / conv2_2 / conv3_3): y = the POOLED framed map
 * (batch, height / 2 + 2, width / 2 + 2, c_out); the full-size map is never written.  Same values as the convolution entry followed
 * by mv3d_maxpool2x2_*. */
int mv3d_conv3x3_pool_views_f16(int num_views, const mv3d_conv_view *views, int c_in, int c_out, void *stream);
int mv3d_conv3x3_pool_views_bf16(int num_views, const mv3d_conv_view *views, int c_in, int c_out, void *stream);
/* The 2x2 pools of several views in one launch.  Forward: x_framed = the map, y_framed = the pooled map (g_pooled_framed unused).
 * Backward (mv3d_maxpool2x2_bwd_*): x_framed = the pre-pool map y, g_pooled_framed = the gradient w.r.t. the pooled map,
 * y_framed = the gradient w.r.t. y's pre-activation (written). */
typedef struct {
    const void *x_framed;
    const void *g_pooled_framed;
    void *y_framed;
    int32_t batch, height, width, reserved0;      /* size of the UNPOOLED map */
} mv3d_pool_view;
int mv3d_maxpool2x2_views_f16(int num_views, const mv3d_pool_view *views, int channels, void *stream);
int mv3d_maxpool2x2_views_bf16(int num_views, const mv3d_pool_view *views, int channels, void *stream);
int mv3d_maxpool2x2_views_f32(int num_views, const mv3d_pool_view *views, int channels, void *stream);
int mv3d_maxpool2x2_bwd_views_bf16(int num_views, const mv3d_pool_view *views, int channels, void *stream);
int mv3d_maxpool2x2_bwd_views_f32(int num_views, const mv3d_pool_view *views, int channels, void *stream);
/* Weight gradient of that convolution (csrc/conv3x3_wgrad.hip): dw (c_out, c_in_real, 3, 3) f32 [the OIHW filter layout; TF's
 * HWIO is its transpose(2, 3, 1, 0)], dw[co][ci][tap] = sum over the pixels of the batch of dy[pixel][co] * x[pixel + tap][ci],
 * for the first c_in_real <= c_in channels (the input layer's buffer is padded to 64 channels); db (c_out) f32 (may be NULL) =
 * the bias gradient, the sum of dy over the pixels, from the same launch; x_framed (batch, height + 2, width + 2, c_in) bf16 = the layer's input,
 * dy_framed (.., c_out) bf16 = the gradient w.r.t. its pre-activation with a ZERO frame; c_in, c_out multiples of 64.
 * workspace: >= mv3d_conv3x3_wgrad_workspace_bytes(...) bytes, 16-byte aligned (split-K partial sums, folded in a fixed order). */
size_t mv3d_conv3x3_wgrad_workspace_bytes(int batch, int height, int width, int c_in, int c_out);
int mv3d_conv3x3_wgrad_bf16(const void *x_framed, const void *dy_framed, float *dw, float *db, int batch, int height, int width,
                            int c_in, int c_in_real, int c_out, void *workspace, size_t workspace_bytes, void *stream);
/* The same on f32 maps (the fp32 training trunk; csrc/conv3x3_wgrad.hip, v_mfma_f32_32x32x2_f32: one ds_read_b32 per operand, no
 * transposing read): exact f32 products and sums, only the summation order differs from any other fp32 weight gradient. */
size_t mv3d_conv3x3_wgrad_f32_workspace_bytes(int batch, int height, int width, int c_in, int c_out);
int mv3d_conv3x3_wgrad_f32(const void *x_framed, const void *dy_framed, float *dw, float *db, int batch, int height, int width,
                           int c_in, int c_in_real, int c_out, void *workspace, size_t workspace_bytes, void *stream);
/* The weight gradients of several views (one filter SHAPE, every view its own maps and its own dw / db: the three trunks at one VGG
 * depth) behind one launch + one reduce launch.  db: NULL in every view or in none.  workspace: >=
 * mv3d_conv3x3_wgrad_views_workspace_bytes(...) bytes (f32_maps: 0 for the bf16 entry, 1 for the f32 one), 16-byte aligned. */
typedef struct {
    const void *x_framed;        /* (batch, height + 2, width + 2, c_in) */
    const void *dy_framed;       /* (batch, height + 2, width + 2, c_out), zero frame */
    float *dw;                   /* (c_out, c_in_real, 3, 3) */
    float *db;                   /* (c_out) or NULL */
    int32_t batch, height, width;
    int32_t c_in_real;           /* this view's real input channels (the input layers: 9 BEV, 3 image / front view); 0 = the call's */
} mv3d_wgrad_view;
size_t mv3d_conv3x3_wgrad_views_workspace_bytes(int num_views, const mv3d_wgrad_view *views, int c_in, int c_out, int f32_maps);
int mv3d_conv3x3_wgrad_views_bf16(int num_views, const mv3d_wgrad_view *views, int c_in, int c_in_real, int c_out, void *workspace,
                                  size_t workspace_bytes, void *stream);
int mv3d_conv3x3_wgrad_views_f32(int num_views, const mv3d_wgrad_view *views, int c_in, int c_in_real, int c_out, void *workspace,
                                 size_t workspace_bytes, void *stream);
/* fp32 OIHW filter (c_out, c_in, 3, 3) -> the packed bf16 forms of one training step in one launch: fwd_packed (c_out, 9 * c_in_pad)
 * [zero-initialised by the caller when c_in_pad > c_in] for mv3d_conv3x3_bf16, dgrad_packed (c_in, 9 * c_out) (may be NULL) for
 * the data-gradient convolution: the filter flipped by 180 degrees with its channel axes swapped. */
int mv3d_conv3x3_pack_bf16(const float *w_oihw, void *fwd_packed, void *dgrad_packed, int c_out, int c_in, int c_in_pad, void *stream);
/* the same for many filters in ONE launch (every 3x3 layer of the training graph at the top of a step) */
typedef struct {
    const float *w_oihw;
    void *fwd_packed, *dgrad_packed;               /* dgrad_packed may be NULL */
    int32_t c_out, c_in, c_in_pad, reserved0;
} mv3d_pack_item;
int mv3d_conv3x3_pack_many_bf16(int num_items, const mv3d_pack_item *items, void *stream);
int mv3d_conv3x3_pack_many_f32(int num_items, const mv3d_pack_item *items, void *stream);    /* f32 packings (c_out % 64 == 0 in both) */
/* framed f16 (batch, height + 2, width + 2, channels) -> framed (batch, height / 2 + 2, width / 2 + 2, channels); channels % 8 == 0 */
int mv3d_maxpool2x2_f16(const void *x_framed, void *y_framed, int batch, int height, int width, int channels, void *stream);
/* NHWC f32 (batch, height, width, channels) -> interior pixels, first `channels` channels of a framed f16 buffer
 * (batch, height + 2, width + 2, channels_out >= channels) */
int mv3d_frame_nhwc_f16(const float *x_nhwc, void *y_framed, int batch, int height, int width, int channels, int channels_out,
                        void *stream);

/* softmax over `rows` rows of `classes` f32 logits (row-major, 8-byte aligned): the RPN's reshape_layer(2) + softmax
 * (lib/networks/network.py:333-341, :399-403) and cls_prob (MV3D_train.py:178); forward only (the losses take the logits). */
int mv3d_softmax_rows(const float *logits_dev, float *prob_dev, long long rows, int classes, void *stream);

/* ------------------------------------------------------------------ the optimizer step (csrc/adam.hip)
 * Replaces tf.train.AdamOptimizer(lr).apply_gradients of lib/fast_rcnn/train_mv.py:138-146 (beta 0.9 / 0.999, eps 1e-8, no weight
 * decay) for ALL parameter tensors behind one launch: per element  m = m + (g - m) (1 - beta1);  v = beta2 v + (1 - beta2) g^2;
 * p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps), f32, in place.
 *   tensors_dev        device array of the tensors (each: f32 param / grad / both moments, numel elements)
 *   chunk_tensor_dev   device int32 [num_chunks]: chunk -> index into tensors_dev
 *   chunk_first_dev    device int32 [num_chunks]: chunk -> its first element / mv3d_adam_chunk_elements() inside that tensor
 *                      (the host cuts every tensor into chunks once; the tables change only with the parameter list)
 *   step               1-based count of this update (the bias corrections)
 *   lowp_dtype         the 16-bit type of the tensors' optional param_lowp copies (what the next step's GEMMs read the weights in) */
typedef struct {
    float *param;
    const float *grad;
    float *exp_avg, *exp_avg_sq;
    long long numel;
    void *param_lowp;            /* NULL, or `numel` 16-bit elements that receive the updated parameter rounded to lowp_dtype */
} mv3d_adam_tensor;
int mv3d_adam_chunk_elements(void);
int mv3d_adam_step(const mv3d_adam_tensor *tensors_dev, const int32_t *chunk_tensor_dev, const int32_t *chunk_first_dev, int num_chunks,
                   double lr, double beta1, double beta2, double eps, int step, int lowp_dtype /* of param_lowp: 0 none, 1 f16, 2 bf16 */,
                   void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MV3D_HIP_H */
