/*
 * mv3d_oracle.h -- CPU restatement of the MV3D RPN -> ROI-pool -> NMS hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under mv3d_tf_amd/ (the product) may
 * include, link or call this; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py do.  Plain C99, single-threaded, scalar.
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no fast-math).
 *
 * Parity pin: checked against golden vectors produced by importing the
 * reference's own Python/Cython in the build container
 * (tests/golden/make_golden.py -> tests/golden/*.npz).  The ROI-pool op of the
 * reference needs TensorFlow headers and cannot be built or run here, so for
 * the two ROI-pool functions parity is pinned by an independent numpy
 * restatement only (see DESIGN.md "parity pins").
 *
 * Every function cites the reference file:line (relative to the upstream
 * repository root) whose arithmetic it restates.
 */
#ifndef MV3D_ORACLE_H
#define MV3D_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* lib/rpn_msr/generate_anchors.py:37-51 -- 4 BEV base anchors (x1,y1,x2,y2). */
void mv3d_ref_generate_anchors_bv(int64_t out[16]);

/* lib/utils/bbox.pyx:15-55 -- (N,4) x (K,4) -> (N,K) IoU in f64, +1 pixel convention. */
void mv3d_ref_bbox_overlaps(const double *boxes, int n, const double *query, int k,
                            double *overlaps);

/* lib/nms/cpu_nms.pyx:17-68 == lib/utils/nms.pyx:17-68.
 * dets (n,5) f32 [x1,y1,x2,y2,score]; order = descending score, ties by
 * descending index (the reference leaves tie order to numpy's unstable sort).
 * presorted!=0: dets already in processing order, no internal sort.
 * Returns number kept (indices into dets, in processing order) or -1 for the
 * reference's ZeroDivisionError (union == 0). */
int mv3d_ref_cpu_nms(const float *dets, int n, double thresh, int presorted, int32_t *keep);
/* lib/nms/nms_kernel.cu:21-30,71,117-133: the CUDA path's rule (IoU > thresh, all f32, pre-sorted boxes).  Parity unpinned. */
int mv3d_ref_gpu_nms_rule(const float *dets, int n, float thresh, int32_t *keep);

/* Defined arithmetic shared (by specification, not by code) with the HIP kernels. */
float mv3d_ref_expf(float x);                 /* DESIGN.md "defined exp" */
double mv3d_ref_log(double x);                /* DESIGN.md "defined log" */
double mv3d_ref_floor_divide(double a, double b); /* numpy npy_floor_divide (f64) */

/* lib/rpn_msr/proposal_layer_tf.py:25-202 (proposal_layer_3d) with everything it
 * calls: lib/utils/transform.py:89-111,81-87 (bv_anchor_to_lidar),
 * lib/fast_rcnn/bbox_transform.py:108-155 (bbox_transform_inv_3d),
 * lib/utils/transform.py:113-142,13-20 (lidar_3d_to_bv), :290-315
 * (lidar_3d_to_corners), :483-500 + :369-386 (active lidar_cnr_to_img),
 * lib/fast_rcnn/bbox_transform.py:178-191 (clip_boxes),
 * lib/rpn_msr/proposal_layer_tf.py:336-352 (_filter_boxes, _filter_img_boxes),
 * :161-174 (sort / top-N / nms / top-N), :188-191 (blobs).
 *
 * prob (1,H,W,2A) f32, pred (1,H,W,6A) f32, im_info[3], calib (4,12) f32.
 * Outputs: blob_bv (R,5), blob_img (R,5), blob_3d (R,7) with R = *n_out <=
 * post_nms_topN (callers size for post_nms_topN rows; post_nms_topN<=0 means
 * "no cap", size for N).  Optional intermediates (NULL to skip), all length-N
 * (N = H*W*4) in anchor order (h,w,a):
 *   anchors3d f64 (N,6); props3d f32 (N,6); bv_raw f32 (N,4) pre-clip;
 *   corners f32 (N,24); img i32 (N,4); valid u8 (N) after both filters;
 *   order i32 (*n_order entries, anchor indices after sort + pre-NMS cut);
 *   nms_keep i32 (*n_out entries, positions into order).
 * Returns 0, or -1 if the NMS hit the reference's ZeroDivisionError. */
typedef struct {
    int feat_stride;      /* 8 */
    int pre_nms_topN;     /* cfg[key].RPN_PRE_NMS_TOP_N  */
    int post_nms_topN;    /* cfg[key].RPN_POST_NMS_TOP_N */
    double nms_thresh;    /* cfg[key].RPN_NMS_THRESH     */
    double min_size;      /* cfg[key].RPN_MIN_SIZE       */
} mv3d_ref_proposal_cfg;

int mv3d_ref_proposal_layer_3d(const float *prob, const float *pred, int H, int W,
                               const float *im_info, const float *calib,
                               const mv3d_ref_proposal_cfg *cfg,
                               float *blob_bv, float *blob_img, float *blob_3d, int *n_out,
                               double *anchors3d, float *props3d, float *bv_raw,
                               float *corners, int32_t *img, uint8_t *valid,
                               int32_t *order, int *n_order, int32_t *nms_keep);

/* lib/utils/transform.py:290-315 and :483-500 exposed separately (used by a17). */
void mv3d_ref_lidar_3d_to_corners(const float *boxes3d, int n, float *corners);
void mv3d_ref_lidar_cnr_to_img(const float *corners, int n, const float *calib, int32_t *img);

/* lib/rpn_msr/anchor_target_layer_tf.py:76-166 -- deterministic part of
 * anchor_target_layer (steps 1-5 and 7 of SURVEY Appendix A.1): inside filter,
 * IoU vs gt_boxes_bv, argmax / max, gt-argmax flood, labels before any random
 * subsampling, 6-d targets for every inside anchor.
 * gt_bv (G,5) f32, gt_3d (G,7) f32.  Outputs sized for N = H*W*4:
 *   inds_inside i32, argmax i32, max_ov f64, labels f32, targets f32 (n,6).
 * Returns n_inside. */
int mv3d_ref_anchor_target_stage1(int H, int W, int feat_stride, const float *im_info,
                                  const float *gt_bv, const float *gt_3d, int G,
                                  double neg_ov, double pos_ov, int clobber_positives,
                                  int32_t *inds_inside, int32_t *argmax, double *max_ov,
                                  float *labels, float *targets);
/* lib/utils/transform.py:89-111 for a list of BEV pixel boxes (int64 (n,4)) -> f64 (n,6). */
void mv3d_ref_bv_anchor_to_lidar(const int64_t *anchors, int n, double *out);

/* lib/fast_rcnn/bbox_transform.py:61-72 -- corner targets (gt - ex)/diag(gt), f32 in,
 * numpy promotes to f32 arithmetic with f32 norm; out f32 (n,24). */
void mv3d_ref_bbox_transform_cnr(const float *ex_cnr, const float *gt_cnr, int n, float *out);

/* lib/roi_pooling_layer/roi_pooling_op.cc:127-181 -- RoiPool forward, NHWC f32.
 * argmax may be NULL.  Returns 0, or -1 if a ROI's batch index is out of range. */
int mv3d_ref_roi_pool_forward(const float *data, int B, int H, int W, int C,
                              const float *rois, int R, int PH, int PW, float scale,
                              float *top, int32_t *argmax);
/* lib/roi_pooling_layer/roi_pooling_op.cc:373-443 -- RoiPoolGrad, gather form. */
void mv3d_ref_roi_pool_backward(const float *top_diff, const int32_t *argmax,
                                int B, int H, int W, int C,
                                const float *rois, int R, int PH, int PW, float scale,
                                float *bottom_diff);
/* ---- SURVEY §8(f) "next" rows ------------------------------------------------------------ */
float mv3d_ref_floor_dividef(float a, float b);   /* numpy npy_divmodf -> floor_divide (f32) */

/* Test-time tail of box_detect (lib/fast_rcnn/test_mv.py:240-261): rois_3d (R,7) [b,x,y,z,l,w,h],
 * deltas (R,24*nc) -> corners (R,24) f32 (lib/utils/transform.py:290-315), pred_cnr_r (R,24*nc) f32
 * (lib/fast_rcnn/bbox_transform.py:157-176), bv (R,4*nc) and bv_r (R,4*nc): corners_to_bv
 * (lib/utils/transform.py:342-366, f32 arithmetic incl. f32 floor-divide) of hstack(corners x nc)
 * and of pred_cnr_r. */
void mv3d_ref_box_tail(const float *rois_3d, const float *deltas, int R, int nc, float *corners,
                       float *pred_cnr_r, float *bv, float *bv_r);

/* BEV rasteriser (lib/utils/read_lidar.py:10-115 with the ranges of tools/read_lidar.py:121-123):
 * points (P,4) f32 -> top (601,601,9) f32; later point (then later height slice, for the
 * reflectance channel) wins a cell, as numpy fancy assignment does. */
void mv3d_ref_point_cloud_2_top(const float *points, int P, float *top);
/* ... with its own parameters (read_lidar.py:10-16); dims[3] = the map's shape (top may be NULL to ask for it), *err = 1 where numpy
 * raises IndexError (a cell outside the map after the wrap of negative indices) */
void mv3d_ref_point_cloud_2_top_ranges(const float *points, int P, double res, double zres, double side0, double side1, double fwd0,
                                       double fwd1, double h0, double h1, int *dims, float *top, int *err);

/* lib/datasets/kitti_mv3d.py:240-272 + lib/utils/transform.py:441-465,502-524,172-187,113-142: camera label boxes -> camera /
 * LIDAR corners, LIDAR box, BEV pixel box (f32 outputs as stored in the roidb).  cos_sin (G,2) f64 = np.cos / np.sin of the
 * yaw, inv_rot (9) f32 = np.linalg.inv(Tr[:, :3]) -- both taken on the host, like the text parsing. */
void mv3d_ref_gt_encode(const float *box_cam, const double *cos_sin, int G, const float *inv_rot, const float *Tr,
                        float *cnr_cam, float *cnr_lidar, float *box_lidar, float *boxes_bv);

/* Front-view ROI of a 3D proposal (PARITY UNPINNED: no reference code, lib/networks/network.py:293-315 is a TODO):
 * rois_3d (R,7) [b,x,y,z,l,w,h] -> rois_fv (R,5) [b,x1,y1,x2,y2] on the 64 x 512 cylindrical map. */
double mv3d_ref_fv_atan2(double y, double x);
void mv3d_ref_rois_3d_to_fv(const float *rois_3d, int R, float *rois_fv);

#ifdef __cplusplus
}
#endif
#endif
