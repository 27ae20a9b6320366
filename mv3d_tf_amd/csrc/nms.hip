// Greedy IoU NMS for gfx950 (wave64).
//
// Replaces lib/nms/cpu_nms.pyx:17-68 (== lib/utils/nms.pyx:17-68) and the CUDA path
// lib/nms/nms_kernel.cu:34-144.  Two kernels:
//
//  nms_mask_kernel    all CUs.  One wavefront <-> one 64x64 tile of the upper triangle of
//                     the suppression matrix: lane = row box (registers), 64 column boxes
//                     staged in LDS and read as broadcasts; one u64 mask word per lane.
//                     For diagonal tiles the wave ballot of each column's predicate is the
//                     TRANSPOSED word (who suppresses column j), which is what the greedy
//                     pass wants, so only that form is stored for them.
//  nms_reduce_kernel  one workgroup per frame: the greedy dependency chain.  `removed`
//                     bitmap in LDS; per 64-box block wave 0 resolves the diagonal tile by
//                     fixed-point iteration on ballots (converges in chain-depth steps to
//                     the unique greedy solution), then all 16 waves OR the rows of the
//                     kept boxes into `removed` (only kept rows are ever read).  Stops as
//                     soon as max_keep boxes are kept (= the reference's keep[:post_nms_topN]).
//
// Arithmetic is the reference's, operation for operation (see pair_suppresses()).
#include <math.h>
#include "kernels.h"

#define NMS_MAX_WORDS 256   // up to 16384 boxes per frame

// lib/nms/cpu_nms.pyx:55-65 for one (kept box i, later box j) pair.  f32, separate IEEE
// ops.  Cython emits ((xx2 - xx1) + 1.0) with a double literal and narrows to f32; for
// f32 operands that equals the f32 add (the f64 sum is exact or rounds identically), so
// the f32 form below is bit-identical.  `tf` is ceil_f32(thresh): (double)ovr >= thresh
// <=> ovr >= tf.
__device__ __forceinline__ bool pair_suppresses(float ix1, float iy1, float ix2, float iy2, float iarea,
                                                float jx1, float jy1, float jx2, float jy2, float jarea,
                                                float tf, int strict_gt, bool &zero_den)
{
    const float xx1 = cy_max(ix1, jx1);
    const float yy1 = cy_max(iy1, jy1);
    const float xx2 = cy_min(ix2, jx2);
    const float yy2 = cy_min(iy2, jy2);
    const float w = cy_max(0.0f, (xx2 - xx1) + 1.0f);
    const float h = cy_max(0.0f, (yy2 - yy1) + 1.0f);
    const float inter = w * h;
    const float den = (iarea + jarea) - inter;
    zero_den = (den == 0.0f);
    const float ovr = inter / den;
    return strict_gt ? (ovr > tf) : (ovr >= tf);
}

struct NmsDev {
    const float *boxes;
    int box_stride;
    long long boxes_frame_stride;
    const int32_t *idx;
    long long idx_frame_stride;
    const int32_t *n_dev;
    int n_cap;
    int nbw;                 // mask words per row = ceil(n_cap/64)
    float tf;
    int strict_gt;
    int max_keep;
    unsigned long long *mask;    // (batch, nbw*64, nbw)
    unsigned long long *diagT;   // (batch, nbw*64)
    int32_t *keep;
    long long keep_frame_stride;
    int32_t *num_keep;
    int32_t *status;
};

__device__ __forceinline__ int frame_n(const NmsDev &d, int f)
{
    int n = d.n_cap;
    if (d.n_dev) { int v = d.n_dev[f]; n = v < n ? v : n; }
    return n < 0 ? 0 : n;
}

__device__ __forceinline__ float4 load_box(const NmsDev &d, int f, int p)
{
    long long q = d.idx ? (long long)d.idx[(long long)f * d.idx_frame_stride + p] : (long long)p;
    const float *b = d.boxes + (long long)f * d.boxes_frame_stride + q * d.box_stride;
    return make_float4(b[0], b[1], b[2], b[3]);
}

// grid: (nbw, ceil(nbw/4), batch); block 256 = 4 waves; wave w owns row block 4*blockIdx.y+w.
__global__ __launch_bounds__(256) void nms_mask_kernel(NmsDev d)
{
    __shared__ float4 s_box[64];
    __shared__ float s_area[64];
    const int f = blockIdx.z;
    const int n = frame_n(d, f);
    const int cb = blockIdx.x;
    if (cb * 64 >= n) return;                       // whole block: nothing in this column block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rb = blockIdx.y * 4 + wave;
    if (threadIdx.x < 64) {
        const int c = cb * 64 + threadIdx.x;
        float4 b = make_float4(NAN, NAN, NAN, NAN); // NaN box: every predicate false
        if (c < n) b = load_box(d, f, c);
        s_box[threadIdx.x] = b;
        s_area[threadIdx.x] = ((b.z - b.x) + 1.0f) * ((b.w - b.y) + 1.0f);   // cpu_nms.pyx:24
    }
    __syncthreads();
    if (rb > cb) return;                            // lower triangle
    const int r = rb * 64 + lane;
    float4 rbx = make_float4(NAN, NAN, NAN, NAN);
    if (r < n) rbx = load_box(d, f, r);
    const float rarea = ((rbx.z - rbx.x) + 1.0f) * ((rbx.w - rbx.y) + 1.0f);
    const bool diag = (rb == cb);
    unsigned long long bits = 0, mycol = 0;
    bool any_zero = false;
#pragma unroll 8
    for (int j = 0; j < 64; ++j) {
        const float4 cbx = s_box[j];
        bool zd;
        bool p = pair_suppresses(rbx.x, rbx.y, rbx.z, rbx.w, rarea, cbx.x, cbx.y, cbx.z, cbx.w, s_area[j],
                                 d.tf, d.strict_gt, zd);
        const bool live = diag ? (lane < j) : true;   // within a tile only earlier rows count
        p = p && live;
        any_zero |= (zd && live && r < n && (cb * 64 + j) < n);
        bits |= (unsigned long long)p << j;
        if (diag) {
            const unsigned long long bal = __ballot(p);   // rows that suppress column j
            if (lane == j) mycol = bal;
        }
    }
    unsigned long long *mask = d.mask + (long long)f * d.nbw * 64 * d.nbw;
    if (diag) d.diagT[(long long)f * d.nbw * 64 + cb * 64 + lane] = mycol;
    else if (r < n) mask[(long long)r * d.nbw + cb] = bits;
    if (d.status && __any(any_zero) && lane == 0) atomicOr(&d.status[f], MV3D_FLAG_ZERO_DIVISION);
}

// grid: (batch); block 1024 = 16 waves.
__global__ __launch_bounds__(1024) void nms_reduce_kernel(NmsDev d)
{
    __shared__ unsigned long long s_removed[NMS_MAX_WORDS];
    __shared__ unsigned long long s_kept;
    const int f = blockIdx.x;
    const int n = frame_n(d, f);
    const int nb = (n + 63) >> 6;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long *mask = d.mask + (long long)f * d.nbw * 64 * d.nbw;
    const unsigned long long *diagT = d.diagT + (long long)f * d.nbw * 64;
    int32_t *keep = d.keep + (long long)f * d.keep_frame_stride;
    for (int w = threadIdx.x; w < NMS_MAX_WORDS; w += blockDim.x) s_removed[w] = 0;
    __syncthreads();
    int total = 0;
    for (int b = 0; b < nb; ++b) {
        if (wave == 0) {
            const int p = b * 64 + lane;
            const bool alive = (p < n) && !((s_removed[b] >> lane) & 1ull);
            const unsigned long long col = diagT[p];        // earlier rows of this block that suppress me
            unsigned long long K = __ballot(alive);
            // K_{t+1} = { alive j : no i in K_t suppresses j }.  Box b*64 has no predecessor in
            // the block, so index k is final after k+1 steps; the fixed point is the greedy set.
            for (;;) {
                const unsigned long long K2 = __ballot(alive && !(col & K));
                if (K2 == K) break;
                K = K2;
            }
            const bool kept = (K >> lane) & 1ull;
            const int pos = total + __popcll(K & ((1ull << lane) - 1ull));
            if (kept && (d.max_keep <= 0 || pos < d.max_keep)) keep[pos] = p;
            if (lane == 0) s_kept = K;
        }
        __syncthreads();
        const unsigned long long K = s_kept;
        total += __popcll(K);
        if (d.max_keep > 0 && total >= d.max_keep) break;
        if (b + 1 < nb && K) {
            // OR the rows of kept boxes into removed[b+1 .. nb): lane <-> word, wave <-> every 16th kept row
            unsigned long long acc[NMS_MAX_WORDS / 64] = {0, 0, 0, 0};
            unsigned long long Kw = K;
            int ord = 0;
            while (Kw) {
                const int i = __builtin_ctzll(Kw);
                Kw &= Kw - 1;
                if ((ord++ & 15) != wave) continue;
                const unsigned long long *row = mask + (long long)(b * 64 + i) * d.nbw;
#pragma unroll
                for (int ps = 0; ps < NMS_MAX_WORDS / 64; ++ps) {
                    const int w = b + 1 + ps * 64 + lane;
                    if (w < nb) acc[ps] |= row[w];
                }
            }
#pragma unroll
            for (int ps = 0; ps < NMS_MAX_WORDS / 64; ++ps) {
                const int w = b + 1 + ps * 64 + lane;
                if (w < nb && acc[ps]) atomicOr(&s_removed[w], acc[ps]);
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int nk = total;
        if (d.max_keep > 0 && nk > d.max_keep) nk = d.max_keep;
        d.num_keep[f] = nk;
    }
}

size_t mv3d_nms_ws_bytes(int n_cap, int batch)
{
    const size_t nbw = (size_t)(n_cap + 63) / 64;
    const size_t rows = nbw * 64;
    return (size_t)batch * (mv3d_align_up(rows * nbw * 8) + mv3d_align_up(rows * 8));
}

int mv3d_launch_nms(const NmsLaunch &L, hipStream_t stream)
{
    if (L.n_cap < 0 || L.batch <= 0) return MV3D_ERR_INVALID_ARG;
    const int nbw = (L.n_cap + 63) / 64;
    if (nbw > NMS_MAX_WORDS) return MV3D_ERR_INVALID_ARG;
    NmsDev d;
    d.boxes = L.boxes; d.box_stride = L.box_stride; d.boxes_frame_stride = L.boxes_frame_stride;
    d.idx = L.idx; d.idx_frame_stride = L.idx_frame_stride; d.n_dev = L.n_dev; d.n_cap = L.n_cap;
    d.nbw = nbw; d.tf = L.thresh_f32; d.strict_gt = L.strict_gt; d.max_keep = L.max_keep;
    const size_t rows = (size_t)nbw * 64;
    d.mask = (unsigned long long *)L.workspace;
    d.diagT = (unsigned long long *)((char *)L.workspace + (size_t)L.batch * mv3d_align_up(rows * nbw * 8));
    d.keep = L.keep; d.keep_frame_stride = L.keep_frame_stride; d.num_keep = L.num_keep; d.status = L.status;
    if (nbw > 0) {
        dim3 grid(nbw, (nbw + 3) / 4, L.batch);
        hipLaunchKernelGGL(nms_mask_kernel, grid, dim3(256), 0, stream, d);
    }
    hipLaunchKernelGGL(nms_reduce_kernel, dim3(L.batch), dim3(1024), 0, stream, d);
    return mv3d_launch_status();
}

// ------------------------------------------------------------------------ C-ABI
extern "C" int mv3d_version(void) { return 100; }

extern "C" const char *mv3d_status_string(int s)
{
    switch (s) {
    case MV3D_OK: return "ok";
    case MV3D_ERR_INVALID_ARG: return "invalid argument";
    case MV3D_ERR_WORKSPACE: return "workspace too small or misaligned";
    case MV3D_ERR_HIP: return "HIP runtime error";
    case MV3D_ERR_ZERO_DIVISION: return "float division";
    default: return "unknown status";
    }
}

extern "C" size_t mv3d_nms_workspace_bytes(int max_boxes)
{
    if (max_boxes < 0 || (max_boxes + 63) / 64 > NMS_MAX_WORDS) return 0;
    return mv3d_nms_ws_bytes(max_boxes, 1);
}

extern "C" int mv3d_nms_device(const float *dets_dev, int n, double thresh, int max_keep, int32_t *keep_dev,
                               int32_t *num_keep_dev, int32_t *status_dev, void *workspace,
                               size_t workspace_bytes, void *stream)
{
    if (n < 0 || !keep_dev || !num_keep_dev || (n > 0 && !dets_dev)) return MV3D_ERR_INVALID_ARG;
    if ((n + 63) / 64 > NMS_MAX_WORDS) return MV3D_ERR_INVALID_ARG;
    if (workspace_bytes < mv3d_nms_ws_bytes(n, 1) || (n > 0 && !workspace) || ((uintptr_t)workspace % MV3D_ALIGN))
        return MV3D_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (status_dev) MV3D_HIP_TRY(hipMemsetAsync(status_dev, 0, sizeof(int32_t), s));
    NmsLaunch L = {};
    L.boxes = dets_dev; L.box_stride = 5; L.n_cap = n; L.batch = 1;
    L.thresh_f32 = mv3d_ceil_f32(thresh); L.strict_gt = 0; L.max_keep = max_keep;
    L.keep = keep_dev; L.keep_frame_stride = n; L.num_keep = num_keep_dev; L.status = status_dev;
    L.workspace = workspace;
    return mv3d_launch_nms(L, s);
}

// keys of a (n,5) dets array for the device sort of mv3d_nms_host
__global__ void nms_score_keys_kernel(const float *dets, int n, uint32_t *keys)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = mv3d_score_key(dets[5 * i + 4]);
}
__global__ void nms_map_keep_kernel(const int32_t *order, const int32_t *keep, const int32_t *num_keep, int32_t *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < num_keep[0]) out[i] = order[keep[i]];
}

static int nms_host_impl(int32_t *keep_out, int32_t *num_out, const float *dets_host, int n, int presorted,
                         float tf, int strict_gt, int device_id)
{
    *num_out = 0;
    if (n == 0) return MV3D_OK;                      // nms_wrapper.py:16-17
    if (n < 0 || !dets_host || !keep_out || (n + 63) / 64 > NMS_MAX_WORDS) return MV3D_ERR_INVALID_ARG;
    MV3D_HIP_TRY(hipSetDevice(device_id));
    const size_t ws_bytes = mv3d_nms_ws_bytes(n, 1);
    char *buf = nullptr;
    const size_t o_dets = 0, o_keys = mv3d_align_up((size_t)n * 20), o_order = o_keys + mv3d_align_up((size_t)n * 4),
                 o_keep = o_order + mv3d_align_up((size_t)n * 4), o_out = o_keep + mv3d_align_up((size_t)n * 4),
                 o_cnt = o_out + mv3d_align_up((size_t)n * 4), o_ws = o_cnt + MV3D_ALIGN;
    MV3D_HIP_TRY(hipMalloc(&buf, o_ws + ws_bytes));
    int rc = MV3D_OK;
    hipStream_t s = nullptr;
    float *dets = (float *)(buf + o_dets);
    uint32_t *keys = (uint32_t *)(buf + o_keys);
    int32_t *order = (int32_t *)(buf + o_order), *keep = (int32_t *)(buf + o_keep), *out = (int32_t *)(buf + o_out);
    int32_t *cnt = (int32_t *)(buf + o_cnt);   // [0] num_keep, [1] status
    int32_t host_cnt[2] = {0, 0};
    do {
        if (hipMemcpyAsync(dets, dets_host, (size_t)n * 20, hipMemcpyHostToDevice, s) != hipSuccess ||
            hipMemsetAsync(cnt, 0, 8, s) != hipSuccess) { rc = MV3D_ERR_HIP; break; }
        NmsLaunch L = {};
        L.boxes = dets; L.box_stride = 5; L.n_cap = n; L.batch = 1; L.thresh_f32 = tf; L.strict_gt = strict_gt;
        L.max_keep = 0; L.keep = keep; L.keep_frame_stride = n; L.num_keep = cnt; L.status = cnt + 1;
        L.workspace = buf + o_ws;
        if (!presorted) {
            hipLaunchKernelGGL(nms_score_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dets, n, keys);
            if ((rc = mv3d_launch_rank(keys, n, 1, order, n, s)) != MV3D_OK) break;
            L.idx = order; L.idx_frame_stride = n;
        }
        if ((rc = mv3d_launch_nms(L, s)) != MV3D_OK) break;
        const int32_t *res = keep;
        if (!presorted) {
            hipLaunchKernelGGL(nms_map_keep_kernel, dim3((n + 255) / 256), dim3(256), 0, s, order, keep, cnt, out);
            res = out;
        }
        if (hipMemcpyAsync(host_cnt, cnt, 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) { rc = MV3D_ERR_HIP; break; }
        if (host_cnt[0] > 0 &&
            hipMemcpy(keep_out, res, (size_t)host_cnt[0] * 4, hipMemcpyDeviceToHost) != hipSuccess) { rc = MV3D_ERR_HIP; break; }
        *num_out = host_cnt[0];
        if (host_cnt[1] & MV3D_FLAG_ZERO_DIVISION) rc = MV3D_ERR_ZERO_DIVISION;
    } while (0);
    (void)hipFree(buf);
    return rc;
}

extern "C" int mv3d_nms_host(int32_t *keep_out, int32_t *num_out, const float *dets_host, int n, double thresh,
                             int device_id)
{
    if (!num_out) return MV3D_ERR_INVALID_ARG;
    return nms_host_impl(keep_out, num_out, dets_host, n, 0, mv3d_ceil_f32(thresh), 0, device_id);
}

extern "C" void _nms(int *keep_out, int *num_out, const float *boxes_host, int boxes_num, int boxes_dim,
                     float nms_overlap_thresh, int device_id)
{
    // lib/nms/nms_kernel.cu:91-144: errors are only printed there; here num_out = 0 on failure.
    int32_t n_out = 0;
    if (boxes_dim == 5) (void)nms_host_impl(keep_out, &n_out, boxes_host, boxes_num, 1, nms_overlap_thresh, 1, device_id);
    if (num_out) *num_out = n_out;
}
