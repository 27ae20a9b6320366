// proposal_layer_3d for gfx950: lib/rpn_msr/proposal_layer_tf.py:25-202 behind one C-ABI call.
//
//   proposal_decode_kernel  one thread per anchor (h,w,a): anchor -> 3D anchor (f64,
//        lib/utils/transform.py:89-111) -> decoded 3D box (f32, numpy's f32 exp restated;
//        lib/fast_rcnn/bbox_transform.py:108-155) -> BEV pixel box (f64 floor-divide,
//        transform.py:113-142) -> 8 corners (transform.py:290-315) -> image box (f32 matrix,
//        f64 product, trunc to i32; transform.py:483-500,369-386) -> clip + both filters
//        (bbox_transform.py:178-191, proposal_layer_tf.py:336-352).  Nothing but the final
//        candidate record (BEV box, image box, 3D box, score key) is written: the ~25 numpy
//        temporaries of the reference never exist.  Reads are the two NHWC head tensors,
//        consecutive anchors = consecutive addresses.
//   rank_kernel (rank.hip)   argsort()[::-1][:pre_nms_topN]
//   nms_mask/reduce (nms.hip) greedy NMS on the sorted candidates, capped at post_nms_topN
//   (ROI blobs)              gathered by the tail of the NMS reduce kernel, unused rows zero-filled.
//
// All shapes are static given (H, W, params); data-dependent counts stay on the device, so
// the whole call is capturable in a hipGraph.
#include <math.h>
#include "geometry.h"
#include "kernels.h"

struct ProposalDev {
    const float *prob, *pred, *im_info, *calib;
    int H, W, N, key_stride;
    int feat_stride, img_h, img_w, img_pad;
    float min_size;
    // candidate records, (batch, N)
    float4 *bv;
    int4 *img;
    float *p3;          // (batch, N, 6)
    uint32_t *key;
    int32_t *blockcnt;  // (batch, gridDim.x) candidates per workgroup (summed by rank_scatter_kernel)
    int32_t *status;    // (batch) flag word, zeroed here (may be NULL)
};

// grid (key_stride / DEC_THREADS, batch): 128 threads per workgroup = 184 workgroups for a KITTI grid, so the f64-heavy
// decode spreads over most of the 256 CUs
#define DEC_THREADS 128
__global__ __launch_bounds__(DEC_THREADS) void proposal_decode_kernel(ProposalDev d)
{
    const int f = blockIdx.y;
    const int n = blockIdx.x * DEC_THREADS + threadIdx.x;
    bool ok = false;
    if (n < d.N) {
        const int a = n & 3, cell = n >> 2, w = cell % d.W, h = cell / d.W;
        const float *info = d.im_info + 3 * f;
        float M[12];
        proj_matrix(d.calib + 48 * f, M);
        const int sx = w * d.feat_stride, sy = h * d.feat_stride;                 // proposal_layer_tf.py:79-95
        float A[6];
        anchor_to_lidar(c_base_anchors[4 * a] + sx, c_base_anchors[4 * a + 1] + sy,
                        c_base_anchors[4 * a + 2] + sx, c_base_anchors[4 * a + 3] + sy, A);
        const float *dl = d.pred + ((long long)f * d.H * d.W + cell) * 24 + 6 * a;   // :105
        float P[6];
        P[0] = __fadd_rn(__fmul_rn(dl[0], A[3]), A[0]);                            // bbox_transform.py:130-135
        P[1] = __fadd_rn(__fmul_rn(dl[1], A[4]), A[1]);
        P[2] = __fadd_rn(__fmul_rn(dl[2], A[5]), A[2]);
        P[3] = __fmul_rn(np_expf(dl[3]), A[3]);
        P[4] = __fmul_rn(np_expf(dl[4]), A[4]);
        P[5] = __fmul_rn(np_expf(dl[5]), A[5]);
        // transform.py:131-137
        const double r0 = (double)__fadd_rn(P[0], __fmul_rn(P[3], 0.5f));
        const double r1 = (double)__fadd_rn(P[1], __fmul_rn(P[4], 0.5f));
        const double r2 = (double)__fsub_rn(P[0], __fmul_rn(P[3], 0.5f));
        const double r3 = (double)__fsub_rn(P[1], __fmul_rn(P[4], 0.5f));
        float B0 = (float)(BV_YN - np_floor_divide(r1 - TOP_Y_MIN_D, BV_RES));
        float B1 = (float)(BV_XN - np_floor_divide(r0 - TOP_X_MIN_D, BV_RES));
        float B2 = (float)(BV_YN - np_floor_divide(r3 - TOP_Y_MIN_D, BV_RES));
        float B3 = (float)(BV_XN - np_floor_divide(r2 - TOP_X_MIN_D, BV_RES));
        int32_t I[4];
        image_box(M, P, I);
        const float xmaxc = info[1] - 1.0f, ymaxc = info[0] - 1.0f;                // bbox_transform.py:184-190
        B0 = np_max32(np_min32(B0, xmaxc), 0.0f);
        B1 = np_max32(np_min32(B1, ymaxc), 0.0f);
        B2 = np_max32(np_min32(B2, xmaxc), 0.0f);
        B3 = np_max32(np_min32(B3, ymaxc), 0.0f);
        const float ws = (B2 - B0) + 1.0f, hs = (B3 - B1) + 1.0f;                 // :336-341
        const float min_size = d.min_size * info[2];
        ok = (ws >= min_size) && (hs >= min_size);
        ok = ok && (-d.img_pad <= I[0]) && (I[2] <= d.img_w + d.img_pad) &&       // :343-352
             (-d.img_pad <= I[1]) && (I[3] <= d.img_h + d.img_pad);
        const long long o = (long long)f * d.N + n;
        d.bv[o] = make_float4(B0, B1, B2, B3);
        d.img[o] = make_int4(I[0], I[1], I[2], I[3]);
        float *q = d.p3 + o * 6;
#pragma unroll
        for (int j = 0; j < 6; ++j) q[j] = P[j];
        const float score = d.prob[((long long)f * d.H * d.W + cell) * 8 + 2 * a + 1];   // :63
        d.key[(long long)f * d.key_stride + n] = ok ? mv3d_score_key(score) : 0u;
    } else if (n < d.key_stride) {
        d.key[(long long)f * d.key_stride + n] = 0u;          // padding the rank kernel relies on
    }
    __shared__ int s_cnt[DEC_THREADS / 64];
    const unsigned long long bal = __ballot(ok);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        int c = 0;
#pragma unroll
        for (int w = 0; w < DEC_THREADS / 64; ++w) c += s_cnt[w];
        d.blockcnt[(long long)f * gridDim.x + blockIdx.x] = c;
        if (blockIdx.x == 0 && d.status) d.status[f] = 0;
    }
}

// ------------------------------------------------------------------------ workspace
struct ProposalLayout {
    int N, order_cap, cap;
    int n_blocks, key_stride;
    size_t o_bv, o_img, o_p3, o_key, o_order, o_sbox, o_keep, o_cnt, o_rank, o_nms, total;
};

static bool proposal_layout(int batch, int H, int W, const mv3d_proposal_params *p, ProposalLayout &L)
{
    if (batch <= 0 || H <= 0 || W <= 0 || !p) return false;
    const long long N = (long long)H * W * 4;
    if (N > 16384 * 4) return false;
    L.N = (int)N;
    L.order_cap = (p->pre_nms_topN > 0 && p->pre_nms_topN < L.N) ? p->pre_nms_topN : L.N;
    if ((L.order_cap + 63) / 64 > 512) return false;        // NMS limit (32768 boxes per frame)
    L.cap = (p->post_nms_topN > 0 && p->post_nms_topN < L.order_cap) ? p->post_nms_topN : L.order_cap;
    size_t o = 0;
    const size_t b = (size_t)batch;
    L.o_bv = o; o += mv3d_align_up(b * L.N * 16);
    L.o_img = o; o += mv3d_align_up(b * L.N * 16);
    L.o_p3 = o; o += mv3d_align_up(b * L.N * 24);
    L.key_stride = mv3d_rank_key_stride(L.N);
    L.o_key = o; o += mv3d_align_up(b * L.key_stride * 4);
    L.o_order = o; o += mv3d_align_up(b * L.order_cap * 4);
    L.o_sbox = o; o += mv3d_align_up(b * L.order_cap * 16);       // BEV boxes in score order (NMS input)
    L.o_keep = o; o += mv3d_align_up(b * L.order_cap * 4);
    L.n_blocks = L.key_stride / DEC_THREADS;
    L.o_cnt = o; o += mv3d_align_up(b * (2 + L.n_blocks) * 4);   // nvalid[batch], num_keep[batch], blockcnt[batch][n_blocks]
    L.o_rank = o; o += mv3d_rank_ws_bytes(L.N, batch);
    L.o_nms = o; o += mv3d_nms_ws_bytes(L.order_cap, batch);
    L.total = o;
    return true;
}

extern "C" int mv3d_proposal_3d_capacity(int H, int W, const mv3d_proposal_params *p)
{
    ProposalLayout L;
    return proposal_layout(1, H, W, p, L) ? L.cap : -1;
}

extern "C" size_t mv3d_proposal_3d_workspace_bytes(int batch, int H, int W, const mv3d_proposal_params *p)
{
    ProposalLayout L;
    return proposal_layout(batch, H, W, p, L) ? L.total : 0;
}

extern "C" int mv3d_proposal_3d(const float *prob_dev, const float *pred_dev, int batch, int H, int W,
                                const float *im_info_dev, const float *calib_dev, const mv3d_proposal_params *p,
                                float *blob_bv_dev, float *blob_img_dev, float *blob_3d_dev, int32_t *num_out_dev,
                                int32_t *status_dev, void *workspace, size_t workspace_bytes, void *stream)
{
    ProposalLayout L;
    if (!proposal_layout(batch, H, W, p, L)) return MV3D_ERR_INVALID_ARG;
    if (!prob_dev || !pred_dev || !im_info_dev || !calib_dev || !blob_bv_dev || !blob_img_dev || !blob_3d_dev ||
        !num_out_dev || p->feat_stride <= 0)
        return MV3D_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < L.total || ((uintptr_t)workspace % MV3D_ALIGN)) return MV3D_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)workspace;
    int32_t *cnt = (int32_t *)(ws + L.o_cnt);          // no memset: every word is written before it is read

    ProposalDev d;
    d.prob = prob_dev; d.pred = pred_dev; d.im_info = im_info_dev; d.calib = calib_dev;
    d.H = H; d.W = W; d.N = L.N; d.key_stride = L.key_stride; d.feat_stride = p->feat_stride;
    d.img_h = p->img_height; d.img_w = p->img_width; d.img_pad = p->img_padding;
    d.min_size = (float)p->min_size;
    d.bv = (float4 *)(ws + L.o_bv); d.img = (int4 *)(ws + L.o_img); d.p3 = (float *)(ws + L.o_p3);
    d.key = (uint32_t *)(ws + L.o_key); d.blockcnt = cnt + 2 * batch; d.status = status_dev;
    hipLaunchKernelGGL(proposal_decode_kernel, dim3(L.n_blocks, batch), dim3(DEC_THREADS), 0, s, d);

    int32_t *order = (int32_t *)(ws + L.o_order), *keep = (int32_t *)(ws + L.o_keep);
    float4 *sbox = (float4 *)(ws + L.o_sbox);
    int rc = mv3d_launch_rank(d.key, L.N, L.key_stride, batch, order, L.order_cap, d.blockcnt, L.n_blocks, cnt, ws + L.o_rank, s,
                              d.bv, sbox);
    if (rc != MV3D_OK) return rc;

    NmsLaunch nl = {};
    nl.boxes = (const float *)sbox; nl.box_stride = 4; nl.boxes_frame_stride = (long long)L.order_cap * 4;
    nl.idx = nullptr; nl.idx_frame_stride = 0;
    nl.n_dev = cnt; nl.n_cap = L.order_cap; nl.batch = batch;
    // nms_wrapper.py:13-21: the cpu_nms rule compares (double)IoU >= thresh, the gpu_nms rule IoU > (float)thresh
    nl.strict_gt = p->nms_strict_gt ? 1 : 0;
    nl.thresh_f32 = nl.strict_gt ? (float)p->nms_thresh : mv3d_ceil_f32(p->nms_thresh);
    nl.max_keep = L.cap;
    nl.keep = keep; nl.keep_frame_stride = L.order_cap; nl.num_keep = cnt + batch; nl.status = status_dev;
    nl.workspace = ws + L.o_nms;
    nl.emit.enabled = 1; nl.emit.N = L.N; nl.emit.order_cap = L.order_cap; nl.emit.cap = L.cap;
    nl.emit.bv = d.bv; nl.emit.img = d.img; nl.emit.p3 = d.p3; nl.emit.order = order;
    nl.emit.blob_bv = blob_bv_dev; nl.emit.blob_img = blob_img_dev; nl.emit.blob_3d = blob_3d_dev;
    nl.emit.num_out = num_out_dev;
    return mv3d_launch_nms(nl, s);
}
