// proposal_target_layer_3d for gfx950 (one frame): lib/rpn_msr/proposal_target_layer_tf.py:19-94
// with _sample_rois_3d (:227-298), lib/utils/bbox.pyx:15-55 (f64 IoU), lib/fast_rcnn/
// bbox_transform.py:61-72 (corner targets), _get_bbox_regression_labels_3d (:172-194) and the image
// projection (lib/utils/transform.py:483-500) fused in.
//
//  pt_overlap_kernel  per candidate ROI (proposals followed by the G ground-truth boxes, :38-44):
//                     f64 IoU against the GT staged in LDS, first-argmax / max.
//  pt_compact_kernel  one workgroup: ordered compaction of the fg (max_ov >= FG_THRESH) and bg
//                     (LO <= max_ov < HI) candidate lists + counts.
//  (host)             draws npr.permutation(n_fg) / (n_bg) from the numpy global RNG: draw-for-draw
//                     parity with `npr.choice(inds, size=k, replace=False)`.
//  pt_emit_kernel     eight lanes per sampled ROI (one per corner): gathers the ROI, its 8 corners (f32), corner
//                     targets (gt - roi) / ||gt_p0 - gt_p6|| in f32, class-slot expansion, image box and, for the
//                     batched entry, the third view's ROI (front_view.hip) of the same 3D box.
#include <math.h>
#include "geometry.h"
#include "front_view.h"

#define PT_MAX_GT 1024

struct PtDev {
    const float *rois_bv, *rois_3d;      // (R,5), (R,7)
    const float *gt_bv, *gt_3d;          // (G,5), (G,7)
    int R, G;
    const int32_t *R_dev;                // optional: the proposals' number lives on the device (<= R = the capacity)
    float gt_frame;                      // batch column of the appended ground-truth rows (0 in the reference)
    double fg_thresh, bg_hi, bg_lo;
    double *max_ov;                      // (R+G)
    int32_t *assign;                     // (R+G)
    int32_t *fg_list, *bg_list;          // (R+G) each
};

struct PtEmit {
    const int32_t *fg_pick, *bg_pick;    // positions in the fg / bg lists
    int n_fg, n_bg, num_classes;
    const float *gt_corners, *calib;     // (G,25), (4,12)
    float *rois_bv, *rois_img, *targets, *rois_3d;
    float *rois_fv;                      // third view's ROIs (front_view.hip), NULL = not wanted
    int32_t *labels;
};

// frames of a batch behind one launch of every kernel: blockIdx.y = frame
#define PT_MAX_BATCH 16
struct PtFrame { PtDev d; PtEmit e; int32_t *counts; };
struct PtBatch { PtFrame f[PT_MAX_BATCH]; };

// row r of the candidate set: proposals then ground truth with a 0 batch column (:38-44); a batched caller sets
// params.frame_index so that the ground-truth rows of frame b carry b like the frame's proposals do
__device__ __forceinline__ int pt_num_rois(const PtDev &d) { return d.R_dev ? min(max(*d.R_dev, 0), d.R) : d.R; }
__device__ __forceinline__ void cand_bv(const PtDev &d, const int R, int r, float o[5])
{
    if (r < R) { for (int j = 0; j < 5; ++j) o[j] = d.rois_bv[5 * r + j]; }
    else { o[0] = d.gt_frame; for (int j = 0; j < 4; ++j) o[1 + j] = d.gt_bv[5 * (r - R) + j]; }
}
__device__ __forceinline__ void cand_3d(const PtDev &d, const int R, int r, float o[7])
{
    if (r < R) { for (int j = 0; j < 7; ++j) o[j] = d.rois_3d[7 * r + j]; }
    else { o[0] = d.gt_frame; for (int j = 0; j < 6; ++j) o[1 + j] = d.gt_3d[7 * (r - R) + j]; }
}

__global__ __launch_bounds__(256) void pt_overlap_kernel(PtBatch bt)
{
    const PtDev &d = bt.f[blockIdx.y].d;
    __shared__ float s_gt[PT_MAX_GT * 4];
    for (int g = threadIdx.x; g < d.G; g += blockDim.x)
        for (int j = 0; j < 4; ++j) s_gt[4 * g + j] = d.gt_bv[5 * g + j];
    __syncthreads();
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int R = pt_num_rois(d);
    if (r >= R + d.G) return;
    float b[5];
    cand_bv(d, R, r, b);
    const double b0 = b[1], b1 = b[2], b2 = b[3], b3 = b[4];
    double mx = 0.0;
    int am = 0;
    // bbox.pyx:33-54; overlaps.argmax(axis=1) = first maximum, overlaps.max(axis=1) (:233-237)
    for (int g = 0; g < d.G; ++g) {
        const double q0 = s_gt[4 * g], q1 = s_gt[4 * g + 1], q2 = s_gt[4 * g + 2], q3 = s_gt[4 * g + 3];
        double o = 0.0;
        const double iw = fmin(b2, q2) - fmax(b0, q0) + 1;
        if (iw > 0) {
            const double ih = fmin(b3, q3) - fmax(b1, q1) + 1;
            if (ih > 0) {
                const double qarea = (q2 - q0 + 1) * (q3 - q1 + 1);
                const double ua = (b2 - b0 + 1) * (b3 - b1 + 1) + qarea - iw * ih;
                o = iw * ih / ua;
            }
        }
        if (g == 0 || o > mx) { mx = o; am = g; }
    }
    d.max_ov[r] = mx;
    d.assign[r] = am;
}

__global__ __launch_bounds__(1024) void pt_compact_kernel(PtBatch bt)
{
    const PtDev &d = bt.f[blockIdx.y].d;
    int32_t *counts = bt.f[blockIdx.y].counts;
    const int n = pt_num_rois(d) + d.G;
    int tot[2];
    mv3d_block_compact<2>(
        n,
        [&](int r, bool f[2]) {
            const double mx = d.max_ov[r];
            f[0] = (mx >= d.fg_thresh);                          // :246
            f[1] = (mx < d.bg_hi) && (mx >= d.bg_lo);            // :259-260
        },
        [&](int k, int pos, int r) { (k == 0 ? d.fg_list : d.bg_list)[pos] = r; },
        tot);
    if (threadIdx.x == 0) { counts[0] = n; counts[1] = tot[0]; counts[2] = tot[1]; counts[3] = 0; }
}


// Eight lanes per sampled ROI (lane k of the group = corner k): the f64 projections of the 8 corners (image box, and the
// front-view box when asked for) run side by side and are folded by the same first-extreme-wins scan as the one-thread
// loops of image_box() / rois_3d_to_fv_kernel; one wave per 8 ROIs instead of 64 puts the ~2 k dependent f64 instructions
// of a ROI on eight times as many SIMDs (the kernel was a single-wave latency chain: 10 us for 256 ROIs).
#define PT_EMIT_THREADS 128
__global__ __launch_bounds__(PT_EMIT_THREADS) void pt_emit_kernel(PtBatch bt)
{
    const PtDev &d = bt.f[blockIdx.y].d;
    const PtEmit &e = bt.f[blockIdx.y].e;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int t = gid >> 3, k = gid & 7;
    const int lane_base = (threadIdx.x & 63) & ~7;
    const int S = e.n_fg + e.n_bg;
    if (t >= S) return;
    // keep_inds = append(fg_inds, bg_inds) (:272)
    const int r = (t < e.n_fg) ? d.fg_list[e.fg_pick[t]] : d.bg_list[e.bg_pick[t - e.n_fg]];
    const int g = d.assign[r];
    // labels = gt_boxes_bv[gt_assignment, 4]; labels[fg_rois_per_this_image:] = 0 (:238, :276)
    const float lab = (t < e.n_fg) ? d.gt_bv[5 * g + 4] : 0.0f;
    float b[5], q[7];
    const int R = pt_num_rois(d);
    cand_bv(d, R, r, b);
    cand_3d(d, R, r, q);
    if (k == 0) {
        for (int j = 0; j < 5; ++j) e.rois_bv[5 * t + j] = b[j];
        for (int j = 0; j < 7; ++j) e.rois_3d[7 * t + j] = q[j];
        e.labels[t] = (int32_t)lab;
    }
    // lidar_3d_to_corners (transform.py:290-315), f32: this lane's corner
    const float *P = q + 1;
    float cx, cy, cz;
    box_corner(P, k, cx, cy, cz);
    // bbox_transform_cnr (bbox_transform.py:61-72): f32; diag = sqrt((d0^2 + d1^2) + d2^2)
    const float *gc = e.gt_corners + 25 * g;
    const float d0 = gc[0] - gc[6], d1 = gc[8] - gc[14], d2 = gc[16] - gc[22];
    float ss = __fmul_rn(d0, d0);
    ss = __fadd_rn(ss, __fmul_rn(d1, d1));
    ss = __fadd_rn(ss, __fmul_rn(d2, d2));
    const float diag = sqrtf(ss);   // IEEE (correctly rounded) sqrt; __fsqrt_rn is the native approximation in HIP
    // _get_bbox_regression_labels_3d (:172-194): class slot, zeros elsewhere; lane k owns columns k, 8 + k, 16 + k of a slot
    const int nc = e.num_classes;
    float *T = e.targets + (long long)t * 24 * nc;
    const int cls = (int)(unsigned short)lab;           // np.array(..., dtype=np.uint16)
    for (int c = 0; c < nc; ++c) {
        const bool own = (c == cls) && cls > 0;
        T[24 * c + k] = own ? __fdiv_rn(gc[k] - cx, diag) : 0.0f;
        T[24 * c + 8 + k] = own ? __fdiv_rn(gc[8 + k] - cy, diag) : 0.0f;
        T[24 * c + 16 + k] = own ? __fdiv_rn(gc[16 + k] - cz, diag) : 0.0f;
    }
    // lidar_cnr_to_img (transform.py:483-500); first column = the ROI's batch index (:76)
    float M[12];
    proj_matrix(e.calib, M);
    double px, py;
    image_point(M, cx, cy, cz, px, py);
    double xmin, xmax, ymin, ymax;
    bool nanx, nany;
    group8_minmax(px, lane_base, xmin, xmax, nanx);
    group8_minmax(py, lane_base, ymin, ymax, nany);
    if (nanx) xmin = xmax = NAN;
    if (nany) ymin = ymax = NAN;
    if (k == 0) {
        e.rois_img[5 * t] = b[0];
        e.rois_img[5 * t + 1] = (float)f64_to_i32(xmin); e.rois_img[5 * t + 2] = (float)f64_to_i32(ymin);
        e.rois_img[5 * t + 3] = (float)f64_to_i32(xmax); e.rois_img[5 * t + 4] = (float)f64_to_i32(ymax);
    }
    if (e.rois_fv) {      // (workgroup-uniform) the third view's ROI: front_view.hip, same arithmetic per corner
        double col, row;
        fv_point((double)cx, (double)cy, (double)cz, col, row);
        double cmin, cmax, rmin, rmax;
        bool nc_, nr_;
        group8_minmax(col, lane_base, cmin, cmax, nc_);
        group8_minmax(row, lane_base, rmin, rmax, nr_);
        if (nc_ || nr_) cmin = cmax = rmin = rmax = NAN;
        if (k == 0) {
            float *o = e.rois_fv + 5 * (long long)t;
            o[0] = q[0];
            o[1] = fv_clip(cmin, FV_W - 1.0); o[2] = fv_clip(rmin, FV_H - 1.0);
            o[3] = fv_clip(cmax, FV_W - 1.0); o[4] = fv_clip(rmax, FV_H - 1.0);
        }
    }
}

struct PtLayout { size_t o_maxov, o_assign, o_fg, o_bg, total; };
static bool pt_layout(int R, int G, PtLayout &L)
{
    if (R < 0 || G <= 0 || G > PT_MAX_GT || (long long)R + G > (1 << 22)) return false;
    const size_t n = (size_t)R + G;
    size_t o = 0;
    L.o_maxov = o; o += mv3d_align_up(n * 8);
    L.o_assign = o; o += mv3d_align_up(n * 4);
    L.o_fg = o; o += mv3d_align_up(n * 4);
    L.o_bg = o; o += mv3d_align_up(n * 4);
    L.total = o;
    return true;
}

static void pt_fill(PtDev &d, const PtLayout &L, char *ws, const float *rois_bv, const float *rois_3d, int R,
                    const float *gt_bv, const float *gt_3d, int G, const mv3d_proposal_target_params *p)
{
    d.rois_bv = rois_bv; d.rois_3d = rois_3d; d.gt_bv = gt_bv; d.gt_3d = gt_3d; d.R = R; d.G = G; d.R_dev = nullptr;
    d.fg_thresh = p->fg_thresh; d.bg_hi = p->bg_thresh_hi; d.bg_lo = p->bg_thresh_lo; d.gt_frame = (float)p->frame_index;
    d.max_ov = (double *)(ws + L.o_maxov); d.assign = (int32_t *)(ws + L.o_assign);
    d.fg_list = (int32_t *)(ws + L.o_fg); d.bg_list = (int32_t *)(ws + L.o_bg);
}

extern "C" size_t mv3d_proposal_target_workspace_bytes(int num_rois, int G)
{
    PtLayout L;
    return pt_layout(num_rois, G, L) ? L.total : 0;
}

static int pt_stage1_batch(int batch, const float *const *rois_bv_dev, const float *const *rois_3d_dev, const int *num_rois,
                           const int32_t *const *num_rois_dev, const float *const *gt_bv_dev, const float *const *gt_3d_dev,
                           const int *G, const mv3d_proposal_target_params *p, int32_t *const *counts_dev,
                           void *const *workspace, const size_t *workspace_bytes, void *stream)
{
    if (batch <= 0 || batch > PT_MAX_BATCH || !p || !rois_bv_dev || !rois_3d_dev || !num_rois || !gt_bv_dev || !gt_3d_dev || !G ||
        !counts_dev || !workspace || !workspace_bytes)
        return MV3D_ERR_INVALID_ARG;
    PtBatch bt = {};
    int most = 0;
    for (int b = 0; b < batch; ++b) {
        PtLayout L;
        if (!pt_layout(num_rois[b], G[b], L) || !gt_bv_dev[b] || !gt_3d_dev[b] || !counts_dev[b] ||
            (num_rois[b] > 0 && (!rois_bv_dev[b] || !rois_3d_dev[b])))
            return MV3D_ERR_INVALID_ARG;
        if (!workspace[b] || workspace_bytes[b] < L.total || ((uintptr_t)workspace[b] % MV3D_ALIGN)) return MV3D_ERR_WORKSPACE;
        pt_fill(bt.f[b].d, L, (char *)workspace[b], rois_bv_dev[b], rois_3d_dev[b], num_rois[b], gt_bv_dev[b], gt_3d_dev[b], G[b], &p[b]);
        bt.f[b].d.R_dev = num_rois_dev ? num_rois_dev[b] : nullptr;
        bt.f[b].counts = counts_dev[b];
        if (num_rois[b] + G[b] > most) most = num_rois[b] + G[b];
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(pt_overlap_kernel, dim3((most + 255) / 256, batch), dim3(256), 0, s, bt);
    hipLaunchKernelGGL(pt_compact_kernel, dim3(1, batch), dim3(1024), 0, s, bt);
    return mv3d_launch_status();
}

extern "C" int mv3d_proposal_target_stage1_batch(int batch, const float *const *rois_bv_dev, const float *const *rois_3d_dev,
                                                 const int *num_rois, const float *const *gt_bv_dev, const float *const *gt_3d_dev,
                                                 const int *G, const mv3d_proposal_target_params *p, int32_t *const *counts_dev,
                                                 void *const *workspace, const size_t *workspace_bytes, void *stream)
{
    return pt_stage1_batch(batch, rois_bv_dev, rois_3d_dev, num_rois, nullptr, gt_bv_dev, gt_3d_dev, G, p, counts_dev, workspace,
                           workspace_bytes, stream);
}

extern "C" int mv3d_proposal_target_stage1_batch_devn(int batch, const float *const *rois_bv_dev, const float *const *rois_3d_dev,
                                                      const int *num_rois_cap, const int32_t *const *num_rois_dev,
                                                      const float *const *gt_bv_dev, const float *const *gt_3d_dev, const int *G,
                                                      const mv3d_proposal_target_params *p, int32_t *const *counts_dev,
                                                      void *const *workspace, const size_t *workspace_bytes, void *stream)
{
    if (!num_rois_dev) return MV3D_ERR_INVALID_ARG;
    for (int b = 0; b < batch && b < PT_MAX_BATCH; ++b) if (!num_rois_dev[b]) return MV3D_ERR_INVALID_ARG;
    return pt_stage1_batch(batch, rois_bv_dev, rois_3d_dev, num_rois_cap, num_rois_dev, gt_bv_dev, gt_3d_dev, G, p, counts_dev,
                           workspace, workspace_bytes, stream);
}

extern "C" int mv3d_proposal_target_stage1(const float *rois_bv_dev, const float *rois_3d_dev, int num_rois,
                                           const float *gt_bv_dev, const float *gt_3d_dev, int G,
                                           const mv3d_proposal_target_params *p, int32_t *counts_dev, void *workspace,
                                           size_t workspace_bytes, void *stream)
{
    return mv3d_proposal_target_stage1_batch(1, &rois_bv_dev, &rois_3d_dev, &num_rois, &gt_bv_dev, &gt_3d_dev, &G, p, &counts_dev,
                                             &workspace, &workspace_bytes, stream);
}

static int pt_stage2_batch(int batch, const float *const *rois_bv_dev, const float *const *rois_3d_dev,
                                                 const int *num_rois, const int32_t *const *num_rois_dev, const float *const *gt_bv_dev, const float *const *gt_3d_dev,
                                                 const float *const *gt_corners_dev, const int *G, const float *const *calib_dev,
                                                 const mv3d_proposal_target_params *p, const int32_t *const *fg_pick_dev,
                                                 const int *n_fg, const int32_t *const *bg_pick_dev, const int *n_bg,
                                                 float *const *rois_bv_out, float *const *rois_img_out, int32_t *const *labels_out,
                                                 float *const *bbox_targets_out, float *const *rois_3d_out,
                                                 float *const *rois_fv_out, void *const *workspace,
                                                 const size_t *workspace_bytes, void *stream)
{
    if (batch <= 0 || batch > PT_MAX_BATCH || !p || !num_rois || !G || !n_fg || !n_bg || !workspace || !workspace_bytes)
        return MV3D_ERR_INVALID_ARG;
    PtBatch bt = {};
    int most = 0;
    for (int b = 0; b < batch; ++b) {
        PtLayout L;
        if (!pt_layout(num_rois[b], G[b], L) || n_fg[b] < 0 || n_bg[b] < 0 || p[b].num_classes <= 0) return MV3D_ERR_INVALID_ARG;
        PtEmit &e = bt.f[b].e;
        e.n_fg = n_fg[b]; e.n_bg = n_bg[b]; e.num_classes = p[b].num_classes;
        if (n_fg[b] + n_bg[b] == 0) continue;
        if (!gt_bv_dev[b] || !gt_3d_dev[b] || !gt_corners_dev[b] || !calib_dev[b] || (n_fg[b] && !fg_pick_dev[b]) ||
            (n_bg[b] && !bg_pick_dev[b]) || !rois_bv_out[b] || !rois_img_out[b] || !labels_out[b] || !bbox_targets_out[b] ||
            !rois_3d_out[b])
            return MV3D_ERR_INVALID_ARG;
        if (!workspace[b] || workspace_bytes[b] < L.total || ((uintptr_t)workspace[b] % MV3D_ALIGN)) return MV3D_ERR_WORKSPACE;
        pt_fill(bt.f[b].d, L, (char *)workspace[b], rois_bv_dev[b], rois_3d_dev[b], num_rois[b], gt_bv_dev[b], gt_3d_dev[b], G[b], &p[b]);
        bt.f[b].d.R_dev = num_rois_dev ? num_rois_dev[b] : nullptr;
        e.fg_pick = fg_pick_dev[b]; e.bg_pick = bg_pick_dev[b];
        e.gt_corners = gt_corners_dev[b]; e.calib = calib_dev[b];
        e.rois_bv = rois_bv_out[b]; e.rois_img = rois_img_out[b]; e.targets = bbox_targets_out[b]; e.rois_3d = rois_3d_out[b];
        e.labels = labels_out[b];
        e.rois_fv = rois_fv_out ? rois_fv_out[b] : nullptr;
        if (n_fg[b] + n_bg[b] > most) most = n_fg[b] + n_bg[b];
    }
    if (most == 0) return MV3D_OK;
    hipLaunchKernelGGL(pt_emit_kernel, dim3((most * 8 + PT_EMIT_THREADS - 1) / PT_EMIT_THREADS, batch), dim3(PT_EMIT_THREADS), 0,
                       (hipStream_t)stream, bt);
    return mv3d_launch_status();
}

#define PT_S2_ARGS gt_bv_dev, gt_3d_dev, gt_corners_dev, G, calib_dev, p, fg_pick_dev, n_fg, bg_pick_dev, n_bg, rois_bv_out, rois_img_out, \
                   labels_out, bbox_targets_out, rois_3d_out, rois_fv_out, workspace, workspace_bytes, stream
extern "C" int mv3d_proposal_target_stage2_batch(int batch, const float *const *rois_bv_dev, const float *const *rois_3d_dev,
                                                 const int *num_rois, const float *const *gt_bv_dev, const float *const *gt_3d_dev,
                                                 const float *const *gt_corners_dev, const int *G, const float *const *calib_dev,
                                                 const mv3d_proposal_target_params *p, const int32_t *const *fg_pick_dev,
                                                 const int *n_fg, const int32_t *const *bg_pick_dev, const int *n_bg,
                                                 float *const *rois_bv_out, float *const *rois_img_out, int32_t *const *labels_out,
                                                 float *const *bbox_targets_out, float *const *rois_3d_out,
                                                 float *const *rois_fv_out, void *const *workspace,
                                                 const size_t *workspace_bytes, void *stream)
{
    return pt_stage2_batch(batch, rois_bv_dev, rois_3d_dev, num_rois, nullptr, PT_S2_ARGS);
}

extern "C" int mv3d_proposal_target_stage2_batch_devn(int batch, const float *const *rois_bv_dev, const float *const *rois_3d_dev,
                                                      const int *num_rois_cap, const int32_t *const *num_rois_dev,
                                                      const float *const *gt_bv_dev, const float *const *gt_3d_dev,
                                                      const float *const *gt_corners_dev, const int *G, const float *const *calib_dev,
                                                      const mv3d_proposal_target_params *p, const int32_t *const *fg_pick_dev,
                                                      const int *n_fg, const int32_t *const *bg_pick_dev, const int *n_bg,
                                                      float *const *rois_bv_out, float *const *rois_img_out, int32_t *const *labels_out,
                                                      float *const *bbox_targets_out, float *const *rois_3d_out,
                                                      float *const *rois_fv_out, void *const *workspace,
                                                      const size_t *workspace_bytes, void *stream)
{
    if (!num_rois_dev) return MV3D_ERR_INVALID_ARG;
    return pt_stage2_batch(batch, rois_bv_dev, rois_3d_dev, num_rois_cap, num_rois_dev, PT_S2_ARGS);
}
#undef PT_S2_ARGS

extern "C" int mv3d_proposal_target_stage2(const float *rois_bv_dev, const float *rois_3d_dev, int num_rois,
                                           const float *gt_bv_dev, const float *gt_3d_dev,
                                           const float *gt_corners_dev, int G, const float *calib_dev,
                                           const mv3d_proposal_target_params *p, const int32_t *fg_pick_dev, int n_fg,
                                           const int32_t *bg_pick_dev, int n_bg, float *rois_bv_out, float *rois_img_out,
                                           int32_t *labels_out, float *bbox_targets_out, float *rois_3d_out,
                                           void *workspace, size_t workspace_bytes, void *stream)
{
    if (!p) return MV3D_ERR_INVALID_ARG;
    return mv3d_proposal_target_stage2_batch(1, &rois_bv_dev, &rois_3d_dev, &num_rois, &gt_bv_dev, &gt_3d_dev, &gt_corners_dev, &G,
                                             &calib_dev, p, &fg_pick_dev, &n_fg, &bg_pick_dev, &n_bg, &rois_bv_out, &rois_img_out,
                                             &labels_out, &bbox_targets_out, &rois_3d_out, nullptr, &workspace, &workspace_bytes, stream);
}
