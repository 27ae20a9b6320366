// anchor_target_layer for gfx950 (one frame): lib/rpn_msr/anchor_target_layer_tf.py:21-250
// with lib/utils/bbox.pyx:15-55 (IoU, f64) and lib/fast_rcnn/bbox_transform.py:32-58 fused in;
// the N x G overlap matrix is never materialised.
//
//  at_overlap_kernel   per anchor: inside test, f64 IoU against the G ground-truth boxes staged
//                      in LDS, first-argmax / max per anchor, per-GT column maximum of the workgroup's
//                      256 anchors (u64 atomicMax in LDS on the bit pattern of the non-negative f64),
//                      stored as one partial per (GT, workgroup): no global atomics, so nothing in the
//                      workspace has to be zero on entry (the look-back words of the two compactions
//                      are cleared by this launch as well) -- stage 1 is three launches and no memset.
//  at_label_kernel     labels before any random subsampling (incl. the "every anchor tying a
//                      GT's column max" flood) and the 6-d targets, written straight
//                      into the full-grid (unmapped) outputs: -1 / 0 fill for outside anchors.
//  at_compact_kernel   one workgroup: ordered (ascending anchor index) compaction of the three
//                      lists the reference subsamples from + their counts.
//  stage 2             applies the host-drawn permutations (numpy MT19937 draws stay on the host:
//                      draw-for-draw parity) and emits the debug anchor lists.
#include <math.h>
#include "common.h"

__constant__ int c_at_base[16] = {-19, -8, 20, 8, -5, -2, 5, 3, -8, -19, 8, 20, -2, -5, 3, 5};
#define AT_MAX_GT 1024

// Defined natural log (f64): x = m 2^k, m in [sqrt(1/2), sqrt(2)), s = (m-1)/(m+1),
// log m = 2 s sum_{d odd <= 39} s^(d-1)/d, Horner with explicit fma; same sequence of IEEE
// operations as oracle/mv3d_oracle.c:mv3d_ref_log.  Targets are rounded to f32 afterwards.
__device__ __forceinline__ double mv3d_log(double x)
{
    if (x != x || x < 0) return NAN;
    if (x == 0) return -INFINITY;
    if (isinf(x)) return x;
    int k;
    double m = frexp(x, &k);
    if (m < 0.70710678118654752440) { m *= 2.0; k -= 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    double p = 1.0 / 39;
#pragma unroll
    for (int d = 37; d >= 1; d -= 2) p = fma(p, z, 1.0 / d);
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double lm = 2.0 * s * p;
    return fma((double)k, LN2_HI, fma((double)k, LN2_LO, lm));
}

// bbox.pyx:33-54 for one (anchor, gt) pair
__device__ __forceinline__ double iou_f64(double b0, double b1, double b2, double b3, double q0, double q1, double q2,
                                          double q3)
{
    const double iw = fmin(b2, q2) - fmax(b0, q0) + 1;
    if (iw > 0) {
        const double ih = fmin(b3, q3) - fmax(b1, q1) + 1;
        if (ih > 0) {
            const double qarea = (q2 - q0 + 1) * (q3 - q1 + 1);
            const double ua = (b2 - b0 + 1) * (b3 - b1 + 1) + qarea - iw * ih;
            return iw * ih / ua;
        }
    }
    return 0.0;
}

struct AtDev {
    int H, W, N, G, stride, clobber;
    double neg_ov, pos_ov;
    const float *im_info, *gt_bv, *gt_3d;
    double *max_ov;                  // (N)  -1 for anchors outside the image
    int32_t *argmax;                 // (N)
    unsigned long long *gtpart;      // (G, nblk) bit pattern of the column max (>= 0) over the anchors of one workgroup
    int nblk;                        // workgroups of at_overlap_kernel per frame
    float *labels, *targets;
};

// Frames of a batch behind ONE launch of every kernel: blockIdx.y = frame (the reference's layer is single-frame; the
// per-frame entry points pass a batch of one).
#define AT_MAX_BATCH 16
struct AtLists {
    int32_t *fg, *bg, *low;          // (N) each: anchor indices, ascending
};
struct AtFrame {
    AtDev d;
    AtLists lists;
    int32_t *counts;                 // (8)
    uint8_t *fg_hi;                  // (N)
    int32_t *agg1, *agg2;            // look-back words of the two grid compactions
    // stage 2
    const int32_t *dis_fg, *dis_bg1, *dis_bg2;
    int n_dis_fg, n_dis_bg1, n_dis_bg2;
    float *anchors, *anchors_3d;
    int32_t *n_anchors;
};
struct AtBatch { AtFrame f[AT_MAX_BATCH]; int cap, n_agg1, n_agg2; };

__device__ __forceinline__ void anchor_coords(const AtDev &d, int n, int &x1, int &y1, int &x2, int &y2)
{
    const int a = n & 3, cell = n >> 2, w = cell % d.W, h = cell / d.W;
    const int sx = w * d.stride, sy = h * d.stride;
    x1 = c_at_base[4 * a] + sx; y1 = c_at_base[4 * a + 1] + sy;
    x2 = c_at_base[4 * a + 2] + sx; y2 = c_at_base[4 * a + 3] + sy;
}

__device__ __forceinline__ bool anchor_box(const AtDev &d, int n, int &x1, int &y1, int &x2, int &y2)
{
    anchor_coords(d, n, x1, y1, x2, y2);
    // anchor_target_layer_tf.py:93-98 (_allowed_border = 0)
    return x1 >= 0 && y1 >= 0 && (double)x2 < (double)d.im_info[1] && (double)y2 < (double)d.im_info[0];
}

__global__ __launch_bounds__(256) void at_overlap_kernel(AtBatch bt)
{
    const AtDev &d = bt.f[blockIdx.y].d;
    if (blockIdx.x == 0) {           // the look-back words of at_compact_kernel / at_emit_relabel_kernel (later launches)
        const AtFrame &F = bt.f[blockIdx.y];
        for (int t = threadIdx.x; t < bt.n_agg1; t += blockDim.x) F.agg1[t] = 0;
        for (int t = threadIdx.x; t < bt.n_agg2; t += blockDim.x) F.agg2[t] = 0;
    }
    __shared__ float s_gt[AT_MAX_GT * 4];
    __shared__ unsigned long long s_max[AT_MAX_GT];
    for (int g = threadIdx.x; g < d.G; g += blockDim.x) {
        s_gt[4 * g + 0] = d.gt_bv[5 * g + 0]; s_gt[4 * g + 1] = d.gt_bv[5 * g + 1];
        s_gt[4 * g + 2] = d.gt_bv[5 * g + 2]; s_gt[4 * g + 3] = d.gt_bv[5 * g + 3];
        s_max[g] = 0ull;
    }
    __syncthreads();
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < d.N) {
        int x1, y1, x2, y2;
        const bool inside = anchor_box(d, n, x1, y1, x2, y2);
        double mx = -1.0;
        int am = 0;
        if (inside) {
            for (int g = 0; g < d.G; ++g) {
                const double o = iou_f64(x1, y1, x2, y2, s_gt[4 * g], s_gt[4 * g + 1], s_gt[4 * g + 2], s_gt[4 * g + 3]);
                if (o > mx) { mx = o; am = g; }           // :118 argmax = first maximum
                if (o > 0.0) atomicMax(&s_max[g], (unsigned long long)__double_as_longlong(o));
            }
        }
        d.max_ov[n] = mx;
        d.argmax[n] = am;
    }
    __syncthreads();
    for (int g = threadIdx.x; g < d.G; g += blockDim.x) d.gtpart[(long long)g * d.nblk + blockIdx.x] = s_max[g];
}

__global__ __launch_bounds__(256) void at_label_kernel(AtBatch bt)
{
    const AtDev &d = bt.f[blockIdx.y].d;
    __shared__ float s_gt[AT_MAX_GT * 4];
    __shared__ unsigned long long s_maxu[AT_MAX_GT];
    for (int g = threadIdx.x; g < d.G; g += blockDim.x) {
        s_gt[4 * g + 0] = d.gt_bv[5 * g + 0]; s_gt[4 * g + 1] = d.gt_bv[5 * g + 1];
        s_gt[4 * g + 2] = d.gt_bv[5 * g + 2]; s_gt[4 * g + 3] = d.gt_bv[5 * g + 3];
        s_maxu[g] = 0ull;
    }
    __syncthreads();
    // column maxima = max over the overlap launch's per-workgroup partials (non-negative f64: the bit patterns order like
    // the values); coalesced over the (GT, workgroup) table, folded with LDS atomics
    for (int i = threadIdx.x; i < d.G * d.nblk; i += blockDim.x) {
        const unsigned long long v = d.gtpart[i];
        if (v) atomicMax(&s_maxu[i / d.nblk], v);
    }
    __syncthreads();
    const double *s_max = reinterpret_cast<const double *>(s_maxu);
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= d.N) return;
    int x1, y1, x2, y2;
    const bool inside = anchor_box(d, n, x1, y1, x2, y2);
    float lab = -1.0f;                                            // :110-111 / _unmap fill
    float T[6] = {0, 0, 0, 0, 0, 0};
    if (inside) {
        const double mx = d.max_ov[n];
        if (!d.clobber && 0 < mx && mx < d.neg_ov) lab = 0.0f;   // :125-130
        for (int g = 0; g < d.G; ++g) {                            // :123,:133 every anchor tying a column max
            const double o = iou_f64(x1, y1, x2, y2, s_gt[4 * g], s_gt[4 * g + 1], s_gt[4 * g + 2], s_gt[4 * g + 3]);
            if (o == s_max[g]) { lab = 1.0f; break; }
        }
        if (mx >= d.pos_ov) lab = 1.0f;                            // :139
        if (d.clobber && mx < d.neg_ov) lab = 0.0f;                // :141-143
        if (d.G > 0) {
            // transform.py:89-111 (f64) and bbox_transform.py:32-58 (f64) then astype(f32)
            const double ex_len = (double)(y2 - y1) * 0.1, ex_wid = (double)(x2 - x1) * 0.1;
            const double cx = (double)(x1 + x2) / 2.0, cy = (double)(y1 + y2) / 2.0;
            const double ey = 600 * 0.1 - (cx + 0.5) * 0.1 + (-30.0);
            const double ex = 600 * 0.1 - (cy + 0.5) * 0.1 + 0.0;
            const double ez = (double)(float)(-(1.73 - 1.56 / 2.0)), eh = (double)(float)1.56;
            const float *gt = d.gt_3d + 7 * d.argmax[n];
            T[0] = (float)(((double)gt[0] - ex) / ex_wid);         // dx / ex_widths  (sic)
            T[1] = (float)(((double)gt[1] - ey) / ex_len);         // dy / ex_lengths (sic)
            T[2] = (float)(((double)gt[2] - ez) / eh);
            T[3] = (float)mv3d_log((double)gt[3] / ex_len);
            T[4] = (float)mv3d_log((double)gt[4] / ex_wid);
            T[5] = (float)mv3d_log((double)gt[5] / eh);
        }
    }
    d.labels[n] = lab;
#pragma unroll
    for (int j = 0; j < 6; ++j) d.targets[6 * (long long)n + j] = T[j];
}

// ---- ordered compaction over a grid of workgroups (mv3d_grid_compact, common.h) ---------------
__global__ __launch_bounds__(256) void at_compact_kernel(AtBatch bt)
{
    const AtFrame &F = bt.f[blockIdx.y];
    const float *labels = F.d.labels;
    const double *max_ov = F.d.max_ov;
    const double neg_ov = F.d.neg_ov;
    const AtLists L = F.lists;
    uint8_t *fg_hi = F.fg_hi;
    int tot[4];
    mv3d_grid_compact<4>(
        F.d.N,
        [&](int n, bool f[4]) {
            const double mx = max_ov[n];
            const float lab = labels[n];
            const bool inside = (mx >= 0.0);                 // inside anchors carry max_ov >= 0
            f[0] = inside; f[1] = inside && (lab == 1.0f); f[2] = inside && (lab == 0.0f); f[3] = inside && (mx < neg_ov);
        },
        [&](int k, int pos, int n) {
            if (k == 1) { L.fg[pos] = n; fg_hi[pos] = (max_ov[n] >= neg_ov) ? 1 : 0; }
            else if (k == 2) L.bg[pos] = n;
            else if (k == 3) L.low[pos] = n;
        },
        F.agg1, tot);
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        int32_t *counts = F.counts;
        counts[0] = tot[0]; counts[1] = tot[1]; counts[2] = tot[2]; counts[3] = tot[3];
        counts[4] = counts[5] = counts[6] = counts[7] = 0;
    }
}

// Stage 2, first launch: the three host-drawn disable lists of anchor_target_layer_tf.py:146-159,178-183 in one pass.
// fg / bg1 positions are applied now (labels = -1); the second bg draw applies AFTER the relabel of :176, so its anchors
// are only marked -- in `argmax`, which stage 1 rewrites on every call and nothing reads afterwards -- and the second
// launch resolves them.
__global__ __launch_bounds__(256) void at_disable_all_kernel(AtBatch bt)
{
    const AtFrame &F = bt.f[blockIdx.y];
    const int n1 = F.n_dis_fg, n2 = F.n_dis_bg1, n3 = F.n_dis_bg2;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n1 + n2 + n3; t += gridDim.x * blockDim.x) {
        if (t < n1) F.d.labels[F.lists.fg[F.dis_fg[t]]] = -1.0f;
        else if (t < n1 + n2) F.d.labels[F.lists.bg[F.dis_bg1[t - n1]]] = -1.0f;
        else F.d.argmax[F.lists.low[F.dis_bg2[t - n1 - n2]]] = -1;
    }
}

// Stage 2, second launch: the debug rows of anchor_target_layer_tf.py:170-174 ((0, anchor) and (0, anchor_3d) for
// labels != -1, an ordered compaction) and then, by the same thread for the same anchors, the final labels:
// labels[max_overlaps < RPN_NEGATIVE_OVERLAP] = 0 (:176) unless the second bg draw disabled the anchor (:178-183).
__global__ __launch_bounds__(256) void at_emit_relabel_kernel(AtBatch bt)
{
    const AtFrame &F = bt.f[blockIdx.y];
    const AtDev &d = F.d;
    float *anchors = F.anchors, *anchors_3d = F.anchors_3d;
    const int cap = bt.cap;
    int tot[1];
    mv3d_grid_compact<1>(
        d.N,
        [&](int n, bool f[1]) { f[0] = anchors && (d.max_ov[n] >= 0.0) && (d.labels[n] != -1.0f); },
        [&](int, int r, int n) {
            if (r >= cap) return;
            int x1, y1, x2, y2;
            anchor_coords(d, n, x1, y1, x2, y2);
            float *A = anchors + 5 * (long long)r, *B = anchors_3d + 7 * (long long)r;
            A[0] = 0.0f; A[1] = (float)x1; A[2] = (float)y1; A[3] = (float)x2; A[4] = (float)y2;
            const double ex_len = (double)(y2 - y1) * 0.1, ex_wid = (double)(x2 - x1) * 0.1;
            const double cx = (double)(x1 + x2) / 2.0, cy = (double)(y1 + y2) / 2.0;
            B[0] = 0.0f;
            B[1] = (float)(600 * 0.1 - (cy + 0.5) * 0.1 + 0.0);
            B[2] = (float)(600 * 0.1 - (cx + 0.5) * 0.1 + (-30.0));
            B[3] = (float)(-(1.73 - 1.56 / 2.0));
            B[4] = (float)ex_len; B[5] = (float)ex_wid; B[6] = (float)1.56;
        },
        F.agg2, tot);
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0 && F.n_anchors) F.n_anchors[0] = tot[0];
    // the items this thread classified above (mv3d_grid_compact: wave w of workgroup b owns [1024 b + 256 w, +256))
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int beg = blockIdx.x * MV3D_GC_ITEMS + wave * 256;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int n = beg + 64 * u + lane;
        if (n < d.N) {
            const double mx = d.max_ov[n];
            if (mx >= 0.0 && mx < d.neg_ov) d.labels[n] = (d.argmax[n] < 0) ? -1.0f : 0.0f;
        }
    }
}

// ------------------------------------------------------------------------ workspace / C-ABI
struct AtLayout { size_t o_maxov, o_argmax, o_gtpart, o_agg1, o_agg2, o_fg, o_bg, o_low, total; int N, ncwg, nblk; };

static bool at_layout(int H, int W, int G, AtLayout &L)
{
    if (H <= 0 || W <= 0 || G < 0 || G > AT_MAX_GT) return false;
    const long long N = (long long)H * W * 4;
    if (N > (1 << 22)) return false;
    L.N = (int)N;
    size_t o = 0;
    L.o_maxov = o; o += mv3d_align_up((size_t)N * 8);
    L.o_argmax = o; o += mv3d_align_up((size_t)N * 4);
    L.nblk = (int)((N + 255) / 256);
    // look-back words of the two grid compactions (cleared by at_overlap_kernel)
    L.ncwg = (int)((N + MV3D_GC_ITEMS - 1) / MV3D_GC_ITEMS);
    L.o_agg1 = o; o += mv3d_align_up((size_t)(L.ncwg * 4 + 1) * 4);
    L.o_agg2 = o; o += mv3d_align_up((size_t)(L.ncwg * 1 + 1) * 4);
    L.o_fg = o; o += mv3d_align_up((size_t)N * 4);
    L.o_bg = o; o += mv3d_align_up((size_t)N * 4);
    L.o_low = o; o += mv3d_align_up((size_t)N * 4);
    // last, so that every other offset is the same whatever G is (stage 2 lays the workspace out without knowing G)
    L.o_gtpart = o; o += mv3d_align_up((size_t)(G > 0 ? G : 1) * L.nblk * 8);
    L.total = o;
    return true;
}

extern "C" size_t mv3d_anchor_target_workspace_bytes(int H, int W, int G)
{
    AtLayout L;
    return at_layout(H, W, G, L) ? L.total : 0;
}

static void at_fill(AtDev &d, const AtLayout &L, int H, int W, int G, const mv3d_anchor_target_params *p, char *ws)
{
    d.H = H; d.W = W; d.N = L.N; d.G = G; d.stride = p->feat_stride; d.clobber = p->clobber_positives;
    d.neg_ov = p->negative_overlap; d.pos_ov = p->positive_overlap;
    d.max_ov = (double *)(ws + L.o_maxov); d.argmax = (int32_t *)(ws + L.o_argmax);
    d.gtpart = (unsigned long long *)(ws + L.o_gtpart); d.nblk = L.nblk;
}

static void at_frame(AtFrame &F, const AtLayout &L, int H, int W, int G, const mv3d_anchor_target_params *p, char *ws)
{
    at_fill(F.d, L, H, W, G, p, ws);
    F.lists.fg = (int32_t *)(ws + L.o_fg); F.lists.bg = (int32_t *)(ws + L.o_bg); F.lists.low = (int32_t *)(ws + L.o_low);
    F.agg1 = (int32_t *)(ws + L.o_agg1); F.agg2 = (int32_t *)(ws + L.o_agg2);
}

extern "C" int mv3d_anchor_target_stage1_batch(int batch, int H, int W, const float *im_info_dev, const float *const *gt_bv_dev,
                                               const float *const *gt_3d_dev, const int *G, const mv3d_anchor_target_params *p,
                                               float *labels_dev, float *targets_dev, int32_t *const *counts_dev,
                                               uint8_t *const *fg_hi_dev, void *const *workspace, size_t workspace_bytes,
                                               void *stream)
{
    if (batch <= 0 || batch > AT_MAX_BATCH || !p || p->feat_stride <= 0 || !im_info_dev || !labels_dev || !targets_dev ||
        !gt_bv_dev || !gt_3d_dev || !G || !counts_dev || !fg_hi_dev || !workspace)
        return MV3D_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    AtBatch bt = {};
    AtLayout L = {};
    for (int b = 0; b < batch; ++b) {
        if (!at_layout(H, W, G[b], L)) return MV3D_ERR_INVALID_ARG;
        if (!counts_dev[b] || !fg_hi_dev[b] || (G[b] > 0 && (!gt_bv_dev[b] || !gt_3d_dev[b]))) return MV3D_ERR_INVALID_ARG;
        if (!workspace[b] || workspace_bytes < L.total || ((uintptr_t)workspace[b] % MV3D_ALIGN)) return MV3D_ERR_WORKSPACE;
        AtFrame &F = bt.f[b];
        at_frame(F, L, H, W, G[b], p, (char *)workspace[b]);
        F.d.im_info = im_info_dev + 3 * b; F.d.gt_bv = gt_bv_dev[b]; F.d.gt_3d = gt_3d_dev[b];
        F.d.labels = labels_dev + (size_t)b * L.N; F.d.targets = targets_dev + (size_t)b * L.N * 6;
        F.counts = counts_dev[b]; F.fg_hi = fg_hi_dev[b];
    }
    bt.n_agg1 = L.ncwg * 4 + 1; bt.n_agg2 = L.ncwg * 1 + 1;
    const int blocks = L.nblk;
    hipLaunchKernelGGL(at_overlap_kernel, dim3(blocks, batch), dim3(256), 0, s, bt);
    hipLaunchKernelGGL(at_label_kernel, dim3(blocks, batch), dim3(256), 0, s, bt);
    hipLaunchKernelGGL(at_compact_kernel, dim3(L.ncwg, batch), dim3(256), 0, s, bt);
    return mv3d_launch_status();
}

extern "C" int mv3d_anchor_target_stage1(int H, int W, const float *im_info_dev, const float *gt_bv_dev,
                                         const float *gt_3d_dev, int G, const mv3d_anchor_target_params *p,
                                         float *labels_dev, float *targets_dev, int32_t *counts_dev, uint8_t *fg_hi_dev,
                                         void *workspace, size_t workspace_bytes, void *stream)
{
    return mv3d_anchor_target_stage1_batch(1, H, W, im_info_dev, &gt_bv_dev, &gt_3d_dev, &G, p, labels_dev, targets_dev,
                                           &counts_dev, &fg_hi_dev, &workspace, workspace_bytes, stream);
}

extern "C" int mv3d_anchor_target_stage2_batch(int batch, int H, int W, const mv3d_anchor_target_params *p,
                                               const int32_t *const *disable_fg_dev, const int *n_dis_fg,
                                               const int32_t *const *disable_bg1_dev, const int *n_dis_bg1,
                                               const int32_t *const *disable_bg2_dev, const int *n_dis_bg2, float *labels_dev,
                                               float *anchors_dev, float *anchors_3d_dev, int32_t *n_anchors_dev,
                                               int anchors_cap, void *const *workspace, size_t workspace_bytes, void *stream)
{
    if (batch <= 0 || batch > AT_MAX_BATCH || !p || !labels_dev || !workspace || !n_dis_fg || !n_dis_bg1 || !n_dis_bg2 ||
        !disable_fg_dev || !disable_bg1_dev || !disable_bg2_dev)
        return MV3D_ERR_INVALID_ARG;
    AtLayout L;
    if (!at_layout(H, W, 0, L)) return MV3D_ERR_INVALID_ARG;
    const bool emit = anchors_dev && anchors_3d_dev && n_anchors_dev;
    AtBatch bt = {};
    bt.cap = anchors_cap;
    int most = 0;
    for (int b = 0; b < batch; ++b) {
        if (n_dis_fg[b] < 0 || n_dis_bg1[b] < 0 || n_dis_bg2[b] < 0) return MV3D_ERR_INVALID_ARG;
        if ((n_dis_fg[b] && !disable_fg_dev[b]) || (n_dis_bg1[b] && !disable_bg1_dev[b]) || (n_dis_bg2[b] && !disable_bg2_dev[b]))
            return MV3D_ERR_INVALID_ARG;
        if (!workspace[b] || workspace_bytes < L.total || ((uintptr_t)workspace[b] % MV3D_ALIGN)) return MV3D_ERR_WORKSPACE;
        AtFrame &F = bt.f[b];
        at_frame(F, L, H, W, 0, p, (char *)workspace[b]);
        F.d.im_info = nullptr; F.d.gt_bv = nullptr; F.d.gt_3d = nullptr; F.d.targets = nullptr;
        F.d.labels = labels_dev + (size_t)b * L.N;
        F.dis_fg = disable_fg_dev[b]; F.dis_bg1 = disable_bg1_dev[b]; F.dis_bg2 = disable_bg2_dev[b];
        F.n_dis_fg = n_dis_fg[b]; F.n_dis_bg1 = n_dis_bg1[b]; F.n_dis_bg2 = n_dis_bg2[b];
        F.anchors = emit ? anchors_dev + (size_t)b * anchors_cap * 5 : nullptr;
        F.anchors_3d = emit ? anchors_3d_dev + (size_t)b * anchors_cap * 7 : nullptr;
        F.n_anchors = emit ? n_anchors_dev + b : nullptr;
        const int n = n_dis_fg[b] + n_dis_bg1[b] + n_dis_bg2[b];
        if (n > most) most = n;
    }
    hipStream_t s = (hipStream_t)stream;
    if (most > 0) hipLaunchKernelGGL(at_disable_all_kernel, dim3((most + 255) / 256, batch), dim3(256), 0, s, bt);
    hipLaunchKernelGGL(at_emit_relabel_kernel, dim3(L.ncwg, batch), dim3(256), 0, s, bt);
    return mv3d_launch_status();
}

extern "C" int mv3d_anchor_target_stage2(int H, int W, const mv3d_anchor_target_params *p,
                                         const int32_t *disable_fg_dev, int n_dis_fg, const int32_t *disable_bg1_dev,
                                         int n_dis_bg1, const int32_t *disable_bg2_dev, int n_dis_bg2, float *labels_dev,
                                         float *anchors_dev, float *anchors_3d_dev, int32_t *n_anchors_dev,
                                         int anchors_cap, void *workspace, size_t workspace_bytes, void *stream)
{
    return mv3d_anchor_target_stage2_batch(1, H, W, p, &disable_fg_dev, &n_dis_fg, &disable_bg1_dev, &n_dis_bg1, &disable_bg2_dev,
                                           &n_dis_bg2, labels_dev, anchors_dev, anchors_3d_dev, n_anchors_dev, anchors_cap,
                                           &workspace, workspace_bytes, stream);
}
