// the four training losses of lib/fast_rcnn/train_mv.py:74-136 and their gradients, fused.
//
//   cross-entropy   mean over the selected rows of  logsumexp(z) - z[label]
//                   (tf.nn.sparse_softmax_cross_entropy_with_logits + tf.reduce_mean, :101, :119)
//   box loss        mean over the selected rows of  sum_d smoothL1_sigma(pred - target)          (:74-90, :111-113, :125-127)
//                   smoothL1(x) = 0.5 (sigma x)^2 if |x| < 1/sigma^2, |x| - 0.5/sigma^2 otherwise
//
// Row selection: RPN (labels f32 in {-1, 0, 1}): cross-entropy over label != -1, box loss over label == 1 (:95-97);
// RCNN (labels i32): every row.  A mean over no rows is NaN, as tf.reduce_mean of an empty tensor.
//
// Two launches, no atomics, deterministic: (1) every workgroup writes the losses' partial sums of its 256 rows and the
// UNSCALED gradients; (2) every workgroup adds the partials up in the same fixed order, scales its rows' gradients
// by 1/count, workgroup 0 writes the two scalars.  All f32, sums in f64 (the reduction order of TF's own kernels
// is not part of the contract; parity is to 1e-5 relative against the numpy restatement and a torch reference).
#include <math.h>
#include "common.h"

struct LossDev {
    const float *cls;            // (R, K) logits
    const float *labels_f;       // (R) f32 labels (RPN) or NULL
    const int32_t *labels_i;     // (R) i32 labels (RCNN) or NULL
    const float *pred, *tgt;     // (R, D)
    int R, K, D;
    float sigma2;
    double *partial;             // (blocks, 4): n_ce, n_box, sum_ce, sum_box
    float *losses;               // [2]
    float *d_cls, *d_pred;       // (R, K), (R, D); may be NULL
};

__global__ __launch_bounds__(256) void loss_partial_kernel(LossDev d)
{
    __shared__ double s_red[4][4];
    const int r = blockIdx.x * 256 + threadIdx.x;
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    if (r < d.R) {
        int label;
        bool use_ce, use_box;
        if (d.labels_f) { const float lf = d.labels_f[r]; label = (int)lf; use_ce = (lf != -1.0f); use_box = (lf == 1.0f); }
        else { label = d.labels_i[r]; use_ce = use_box = true; }
        const float *z = d.cls + (long long)r * d.K;
        float *dz = d.d_cls ? d.d_cls + (long long)r * d.K : nullptr;
        if (use_ce) {
            float m = z[0];
            for (int k = 1; k < d.K; ++k) m = fmaxf(m, z[k]);
            float se = 0.0f;
            for (int k = 0; k < d.K; ++k) se += expf(z[k] - m);
            const bool ok = (label >= 0 && label < d.K);
            const float zl = ok ? z[label] : NAN;                      // an out-of-range label poisons the loss (TF: NaN / error)
            v[0] = 1.0; v[2] = (double)(logf(se) - (zl - m));
            if (dz) for (int k = 0; k < d.K; ++k) dz[k] = expf(z[k] - m) / se - ((k == label) ? 1.0f : 0.0f);
        } else if (dz) {
            for (int k = 0; k < d.K; ++k) dz[k] = 0.0f;
        }
        const float *p = d.pred + (long long)r * d.D, *t = d.tgt + (long long)r * d.D;
        float *dp = d.d_pred ? d.d_pred + (long long)r * d.D : nullptr;
        if (use_box) {
            const float thr = 1.0f / d.sigma2;
            float acc = 0.0f;
            for (int j = 0; j < d.D; ++j) {
                const float x = p[j] - t[j], ax = fabsf(x);
                const bool quad = ax < thr;                              // train_mv.py:83
                acc += quad ? (x * x) * (0.5f * d.sigma2) : (ax - 0.5f / d.sigma2);
                if (dp) dp[j] = quad ? d.sigma2 * x : (x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f));
            }
            v[1] = 1.0; v[3] = (double)acc;
        } else if (dp) {
            for (int j = 0; j < d.D; ++j) dp[j] = 0.0f;
        }
    }
    // block sums in a fixed order: wave shuffles, then the four wave totals
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double x = v[q];
        for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][q] = x;
    }
    __syncthreads();
    if (threadIdx.x < 4)
        d.partial[(long long)blockIdx.x * 4 + threadIdx.x] =
            ((s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + s_red[2][threadIdx.x]) + s_red[3][threadIdx.x];
}

__global__ __launch_bounds__(256) void loss_final_kernel(LossDev d)
{
    __shared__ double s_tot[4];
    if (threadIdx.x < 4) {
        double t = 0.0;
        for (int b = 0; b < (int)gridDim.x; ++b) t += d.partial[(long long)b * 4 + threadIdx.x];   // same order everywhere
        s_tot[threadIdx.x] = t;
    }
    __syncthreads();
    const double n_ce = s_tot[0], n_box = s_tot[1];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        d.losses[0] = (float)(s_tot[2] / n_ce);                          // 0/0 = NaN: mean of an empty selection
        d.losses[1] = (float)(s_tot[3] / n_box);
    }
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= d.R) return;
    // an empty selection gives a NaN loss VALUE but ZERO gradients (TF: reduce_mean over an empty gather back-propagates
    // nothing), so the scale of the zero-filled gradient rows must not become 1/0
    const float ice = n_ce > 0.0 ? (float)(1.0 / n_ce) : 0.0f, ibox = n_box > 0.0 ? (float)(1.0 / n_box) : 0.0f;
    if (d.d_cls) { float *dz = d.d_cls + (long long)r * d.K; for (int k = 0; k < d.K; ++k) dz[k] *= ice; }
    if (d.d_pred) { float *dp = d.d_pred + (long long)r * d.D; for (int j = 0; j < d.D; ++j) dp[j] *= ibox; }
}

extern "C" size_t mv3d_loss_workspace_bytes(int rows)
{
    return rows < 0 ? 0 : mv3d_align_up((size_t)((rows + 255) / 256 + 1) * 4 * sizeof(double));
}

static int launch_loss(LossDev d, void *workspace, size_t workspace_bytes, void *stream)
{
    if (d.R < 0 || d.K <= 0 || d.D <= 0 || !d.losses || !(d.sigma2 > 0.0f)) return MV3D_ERR_INVALID_ARG;
    if (d.R > 0 && (!d.cls || !d.pred || !d.tgt || (!d.labels_f && !d.labels_i))) return MV3D_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < mv3d_loss_workspace_bytes(d.R) || ((uintptr_t)workspace % MV3D_ALIGN)) return MV3D_ERR_WORKSPACE;
    d.partial = (double *)workspace;
    const int blocks = d.R > 0 ? (d.R + 255) / 256 : 1;
    hipLaunchKernelGGL(loss_partial_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d);
    hipLaunchKernelGGL(loss_final_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d);
    return mv3d_launch_status();
}

extern "C" int mv3d_rpn_loss(const float *rpn_cls_score_dev, const float *rpn_labels_dev, const float *rpn_bbox_pred_dev,
                             const float *rpn_bbox_targets_dev, int num_anchors, float sigma, float *losses_dev,
                             float *d_cls_score_dev, float *d_bbox_pred_dev, void *workspace, size_t workspace_bytes,
                             void *stream)
{
    LossDev d = {};
    d.cls = rpn_cls_score_dev; d.labels_f = rpn_labels_dev; d.pred = rpn_bbox_pred_dev; d.tgt = rpn_bbox_targets_dev;
    d.R = num_anchors; d.K = 2; d.D = 6; d.sigma2 = sigma * sigma; d.losses = losses_dev;
    d.d_cls = d_cls_score_dev; d.d_pred = d_bbox_pred_dev;
    return launch_loss(d, workspace, workspace_bytes, stream);
}

extern "C" int mv3d_rcnn_loss(const float *cls_score_dev, const int32_t *labels_dev, const float *bbox_pred_dev,
                              const float *bbox_targets_dev, int num_rois, int num_classes, int box_dim, float sigma,
                              float *losses_dev, float *d_cls_score_dev, float *d_bbox_pred_dev, void *workspace,
                              size_t workspace_bytes, void *stream)
{
    LossDev d = {};
    d.cls = cls_score_dev; d.labels_i = labels_dev; d.pred = bbox_pred_dev; d.tgt = bbox_targets_dev;
    d.R = num_rois; d.K = num_classes; d.D = box_dim; d.sigma2 = sigma * sigma; d.losses = losses_dev;
    d.d_cls = d_cls_score_dev; d.d_pred = d_bbox_pred_dev;
    return launch_loss(d, workspace, workspace_bytes, stream);
}

// ---- softmax over rows of K logits (K = 2: the RPN's reshape_layer(2) + softmax, lib/networks/MV3D_train.py:90-93 / network.py:399-403, and
// cls_prob of the two-class head): max, exp of the differences, sum in class order, divide -- an elementwise pass over 92 k pairs per
// frame, one thread per row.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float *__restrict__ x, float *__restrict__ y, long long rows, int K)
{
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float *xr = x + r * K;
    float *yr = y + r * K;
    if (K == 2) {
        const float2 v = *reinterpret_cast<const float2 *>(xr);
        const float m = fmaxf(v.x, v.y);
        const float ea = expf(v.x - m), eb = expf(v.y - m), s = ea + eb;
        *reinterpret_cast<float2 *>(yr) = make_float2(ea / s, eb / s);
        return;
    }
    float m = xr[0];
    for (int k = 1; k < K; ++k) m = fmaxf(m, xr[k]);
    float s = 0.0f;
    for (int k = 0; k < K; ++k) { const float e = expf(xr[k] - m); yr[k] = e; s += e; }
    for (int k = 0; k < K; ++k) yr[k] = yr[k] / s;
}

extern "C" int mv3d_softmax_rows(const float *logits_dev, float *prob_dev, long long rows, int classes, void *stream)
{
    if (!logits_dev || !prob_dev || rows < 0 || classes < 1 || classes > 1024) return MV3D_ERR_INVALID_ARG;
    if (((uintptr_t)logits_dev | (uintptr_t)prob_dev) & 7) return MV3D_ERR_INVALID_ARG;
    if (rows == 0) return MV3D_OK;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, logits_dev, prob_dev, rows, classes);
    return mv3d_launch_status();
}
