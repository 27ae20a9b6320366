// Greedy IoU NMS for gfx950 (wave64).
//
// Replaces lib/nms/cpu_nms.pyx:17-68 (== lib/utils/nms.pyx:17-68) and the CUDA path
// lib/nms/nms_kernel.cu:34-144.  Two kernels:
//
//  nms_mask_kernel    all CUs, one single-wave workgroup per 64x64 tile of the UPPER TRIANGLE of
//                     the suppression matrix (tiles enumerated column-block-major so the launch is
//                     balanced and the tiles past the live box count fall off the end): lane = row
//                     box (registers), the 64 column boxes staged in LDS and read back as
//                     broadcasts.  Tiles within NMS_BAND blocks of the diagonal are stored in COLUMN
//                     form (lane j gets the ballot of column j's predicate = "which rows of the
//                     row block suppress box j"); far tiles in ROW form (one u64 per row).
//  nms_reduce_kernel  one workgroup per frame = the greedy dependency chain, software-pipelined by
//                     wave specialisation:
//                       wave 0 (chain)   per 64-box block: removed bits -> fixed-point iteration on
//                                        ballots over the diagonal tile (converges to the unique
//                                        greedy set) -> kept mask K_b; then the near-band column
//                                        tiles turn K_b into removed bits of the next NMS_BAND-1
//                                        blocks with one AND + one wave compare each.  All its
//                                        global addresses are static, so its loads run 4 blocks
//                                        ahead: the per-block critical path is ALU only.
//                       waves 1-15       trail the chain: OR the ROW-form words of kept rows into
//                                        the LDS bitmap for blocks >= NMS_BAND ahead (only kept rows
//                                        are ever read).  LDS flags (chain position / per-worker
//                                        progress) order the two roles; no global synchronisation.
//                     Stops as soon as max_keep boxes are kept (= the reference's
//                     keep[:post_nms_topN]) and, for proposal_layer_3d, gathers the ROI blobs.
//
// Arithmetic is the reference's, operation for operation (see pair_suppresses()).
#include <math.h>
#include "kernels.h"

#define NMS_MAX_WORDS 256   // up to 16384 boxes per frame
#define NMS_BAND 8          // diagonal + 7 following column blocks are kept in column form
#define NMS_WORKERS 15
#define NMS_GROUPS 3            // worker groups; group g owns the blocks b = g (mod NMS_GROUPS)
#define NMS_GW 5                // workers per group
#define NMS_ROWS_PER_WORKER 13  // ceil(64 / NMS_GW)
#define NMS_PF 4            // chain-wave prefetch depth (blocks)

// lib/nms/cpu_nms.pyx:55-65 for one (kept box i, later box j) pair.  f32, separate IEEE
// ops.  Cython emits ((xx2 - xx1) + 1.0) with a double literal and narrows to f32; for
// f32 operands that equals the f32 add (the f64 sum is exact or rounds identically), so
// the f32 form below is bit-identical.  `tf` is ceil_f32(thresh): (double)ovr >= thresh
// <=> ovr >= tf.  The CUDA rule `ovr > thresh` (nms_kernel.cu:71) is served by the same
// compare with tf = nextafter(thresh, +inf).
__device__ __forceinline__ bool pair_suppresses(float ix1, float iy1, float ix2, float iy2, float iarea,
                                                float jx1, float jy1, float jx2, float jy2, float jarea,
                                                float tf, bool &zero_den)
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
    return ovr >= tf;
}

struct NmsDev {
    const float *boxes;
    int box_stride;
    long long boxes_frame_stride;
    const int32_t *idx;
    long long idx_frame_stride;
    const int32_t *n_dev;
    int n_cap;
    int nbw;                     // blocks of 64 boxes per frame (capacity)
    int nbs;                     // row-form words per row = nbw rounded up to even (16-byte rows)
    float tf;
    float neg_h;                 // -(half the gap below tf), see tile_fast()
    int fast_ok;                 // tf is a positive normal f32 in a range where tile_fast() is valid
    int max_keep;
    unsigned long long *mask;    // (batch, nbw*64, nbw)        row form, far tiles
    unsigned long long *band;    // (batch, nbw, NMS_BAND, 64)  column form, near tiles
    int32_t *keep;
    long long keep_frame_stride;
    int32_t *num_keep;
    int32_t *status;
    EmitDev emit;
    long long *trace;            // diagnostics: 4 x i64 per block (frame 0 only), may be NULL
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

// Single-instruction max / min for the NaN-free fast path (fmaxf/fminf would add input
// canonicalisation: two extra v_max per call under IEEE mode).
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
typedef float f2 __attribute__((ext_vector_type(2)));

// "tame": finite and small enough (|coordinate| < 2^18) that no intermediate of
// pair_suppresses() can overflow, be NaN or lose the exactness arguments below.  For tame
// boxes cy_max/cy_min (a>=b?a:b) and v_max/v_min agree except for the sign of a zero, which
// cannot change any later result (it only feeds x - y + 1 and w * h).
__device__ __forceinline__ bool tame(float4 b)
{
    const float L = 0x1p18f;
    return fabsf(b.x) < L && fabsf(b.y) < L && fabsf(b.z) < L && fabsf(b.w) < L;
}

// Division-free exact form of  RN(inter / den) >= tf  for den > 0, inter >= 0 and a positive
// normal tf:  let h = half the distance from tf to its f32 predecessor (a power of two) and
// mid = tf - h.  RN(q) >= tf  <=>  q > mid, or q == mid and the tie rounds to tf.  With
// e = fma(-tf, den, inter) (ONE rounding of inter - tf*den) and nc = -h*den (exact):
//     e > nc  =>  inter - tf*den > -h*den  =>  q > mid            => suppressed
//     e < nc  =>  q < mid                                         => not suppressed
//     e == nc =>  undecided (q within a rounding of mid)          => the tile is redone exactly
// (RN is monotone, so e > nc cannot come from an exact value <= nc).  Everything is f32 and
// packed two columns per instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32).
//
// One 64x64 tile, all 128 boxes valid and tame.  `lb` is the lane's own box, `s_box/s_area` the
// 64 boxes of the other side (LDS broadcasts).  Bit j of the result = predicate(lane box, box j).
// DIAG_SWAPPED: the lane box is the COLUMN (later) box of a diagonal tile: only rows j < lane count.
template <bool DIAG_SWAPPED>
__device__ __forceinline__ unsigned long long tile_fast(const float4 lb, const float larea, const float4 *s_box,
                                                        const float *s_area, const float tf, const float neg_h,
                                                        const int lane, float &min_den, bool &undecided)
{
    unsigned lo = 0, hi = 0;
    bool amb = false;
    const f2 ntf = {-tf, -tf}, nh = {neg_h, neg_h}, one = {1.0f, 1.0f}, la = {larea, larea};
#pragma unroll
    for (int j = 0; j < 64; j += 2) {
        const float4 A = s_box[j], B = s_box[j + 1];
        const f2 ca = *reinterpret_cast<const f2 *>(s_area + j);
        const f2 xx1 = {vmax(lb.x, A.x), vmax(lb.x, B.x)}, yy1 = {vmax(lb.y, A.y), vmax(lb.y, B.y)};
        const f2 xx2 = {vmin(lb.z, A.z), vmin(lb.z, B.z)}, yy2 = {vmin(lb.w, A.w), vmin(lb.w, B.w)};
        const f2 dx = (xx2 - xx1) + one, dy = (yy2 - yy1) + one;
        const f2 w = {vmax(0.0f, dx.x), vmax(0.0f, dx.y)}, h = {vmax(0.0f, dy.x), vmax(0.0f, dy.y)};
        const f2 inter = w * h;
        const f2 den = (la + ca) - inter;
        min_den = vmin3(min_den, den.x, den.y);
        const f2 e = __builtin_elementwise_fma(ntf, den, inter);
        const f2 nc = den * nh;
        bool p0 = e.x > nc.x, p1 = e.y > nc.y;
        amb |= (e.x == nc.x) | (e.y == nc.y);
        if (DIAG_SWAPPED) { p0 = p0 && (j < lane); p1 = p1 && (j + 1 < lane); }
        if (j < 32) { lo |= p0 ? (1u << j) : 0u; lo |= p1 ? (2u << j) : 0u; }
        else { hi |= p0 ? (1u << (j - 32)) : 0u; hi |= p1 ? (2u << (j - 32)) : 0u; }
    }
    undecided = amb;
    return ((unsigned long long)hi << 32) | lo;
}

// grid: (nbw*(nbw+1)/2 tiles, 1, batch); block 64 = one wave per tile.
__global__ __launch_bounds__(64) void nms_mask_kernel(NmsDev d)
{
    __shared__ float4 s_box[64];
    __shared__ float s_area[64];
    const int f = blockIdx.z;
    const int n = frame_n(d, f);
    // tile t -> (cb, rb <= cb), column-block-major: t = cb(cb+1)/2 + rb
    const int t = blockIdx.x;
    int cb = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while ((cb + 1) * (cb + 2) / 2 <= t) ++cb;
    while (cb * (cb + 1) / 2 > t) --cb;
    const int rb = t - cb * (cb + 1) / 2;
    if (cb * 64 >= n) return;
    const int lane = threadIdx.x;
    const int dist = cb - rb;
    const bool diag = (dist == 0), colform = (dist < NMS_BAND);
    // Row form: lane = row box, LDS = column boxes, bit j = column j.  Column form (near band):
    // the roles are swapped -- the predicate is symmetric in the two boxes (max/min/+/* commute) --
    // so the lane's word is directly "which rows of block rb suppress my column".
    const int lds_idx = (colform ? rb : cb) * 64 + lane, own_idx = (colform ? cb : rb) * 64 + lane;
    float4 sb = make_float4(NAN, NAN, NAN, NAN);    // NaN box: every predicate false
    if (lds_idx < n) sb = load_box(d, f, lds_idx);
    s_box[lane] = sb;
    s_area[lane] = ((sb.z - sb.x) + 1.0f) * ((sb.w - sb.y) + 1.0f);          // cpu_nms.pyx:24
    const bool lds_tame = __all(tame(sb));          // false if the block is ragged (NaN padding)
    __syncthreads();
    float4 lb = make_float4(NAN, NAN, NAN, NAN);
    if (own_idx < n) lb = load_box(d, f, own_idx);
    const float larea = ((lb.z - lb.x) + 1.0f) * ((lb.w - lb.y) + 1.0f);
    unsigned long long word = 0;
    bool any_zero = false, done = false;
    if (d.fast_ok && lds_tame && __all(tame(lb))) {
        float min_den = 1.0f;
        bool und;
        if (diag) word = tile_fast<true>(lb, larea, s_box, s_area, d.tf, d.neg_h, lane, min_den, und);
        else word = tile_fast<false>(lb, larea, s_box, s_area, d.tf, d.neg_h, lane, min_den, und);
        // a denominator that is not safely positive, or an undecided compare: redo the tile exactly
        done = !__any(und || !(min_den >= 0x1p-20f));
    }
    if (!done) {
        // exact path: ragged blocks, NaN / huge coordinates, zero / negative unions, undecided compares
        word = 0;
        for (int j = 0; j < 64; ++j) {
            const float4 q = s_box[j];
            bool zd;
            // arguments in (kept box i, later box j) order of cpu_nms.pyx: i = row side
            bool p = colform ? pair_suppresses(q.x, q.y, q.z, q.w, s_area[j], lb.x, lb.y, lb.z, lb.w, larea, d.tf, zd)
                             : pair_suppresses(lb.x, lb.y, lb.z, lb.w, larea, q.x, q.y, q.z, q.w, s_area[j], d.tf, zd);
            const bool live = diag ? (j < lane) : true;                        // rows before my column
            p = p && live;
            any_zero |= (zd && live && own_idx < n && ((colform ? rb : cb) * 64 + j) < n);
            word |= (unsigned long long)p << j;
        }
    }
    if (colform) d.band[(((long long)f * d.nbw + rb) * NMS_BAND + dist) * 64 + lane] = word;
    else if (own_idx < n) d.mask[((long long)f * d.nbw * 64 + own_idx) * d.nbs + cb] = word;
    if (d.status && __any(any_zero) && lane == 0) atomicOr(&d.status[f], MV3D_FLAG_ZERO_DIVISION);
}

__device__ __forceinline__ int lds_load_i32(const int *p)
{
    const int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    return v;
}
__device__ __forceinline__ void lds_store_i32(int *p, int v)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// grid: (batch); block 1024 = 16 waves: wave 0 = chain, waves 1..15 = bulk workers.
__global__ __launch_bounds__(1024) void nms_reduce_kernel(NmsDev d)
{
    __shared__ unsigned long long s_removed[NMS_MAX_WORDS];   // far-band contributions (workers, ds_or)
    __shared__ unsigned long long s_kept[NMS_MAX_WORDS];      // K_b, valid once s_chain_pos > b
    __shared__ int s_chain_pos, s_done, s_total;
    __shared__ int s_wprog[NMS_WORKERS + 1];
    const int f = blockIdx.x;
    const int n = frame_n(d, f);
    const int nb = (n + 63) >> 6;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long *mask = d.mask + (long long)f * d.nbw * 64 * d.nbs;
    const unsigned long long *band = d.band + (long long)f * d.nbw * NMS_BAND * 64;
    int32_t *keep = d.keep + (long long)f * d.keep_frame_stride;
    for (int w = threadIdx.x; w < NMS_MAX_WORDS; w += blockDim.x) s_removed[w] = 0;
    if (threadIdx.x < NMS_WORKERS + 1) s_wprog[threadIdx.x] = 0;
    if (threadIdx.x == 0) { s_chain_pos = 0; s_done = 0; s_total = 0; }
    __syncthreads();

    if (wave == 0) {
        // ------------------------------------------------------------------ chain wave
        __builtin_amdgcn_s_setprio(3);
        unsigned long long ring[NMS_PF][NMS_BAND];
#pragma unroll
        for (int s = 0; s < NMS_PF; ++s)
#pragma unroll
            for (int k = 0; k < NMS_BAND; ++k)
                ring[s][k] = band[((long long)min(s, d.nbw - 1) * NMS_BAND + k) * 64 + lane];
        unsigned long long acc[NMS_BAND];             // acc[k]: near-band removed bits for block b+k
#pragma unroll
        for (int k = 0; k < NMS_BAND; ++k) acc[k] = 0ull;
        int total = 0;
        bool stop = false;
        for (int b0 = 0; b0 < nb && !stop; b0 += NMS_PF) {
#pragma unroll
            for (int s = 0; s < NMS_PF; ++s) {
                const int b = b0 + s;
                if (b >= nb || stop) break;
                const long long t_begin = d.trace ? (long long)__builtin_readcyclecounter() : 0;
                // far-band words for this block come from rows of blocks <= b - NMS_BAND
                if (b >= NMS_BAND) {
                    const int need = b - NMS_BAND + 1;
                    for (;;) {
                        const int pr = (lane < NMS_WORKERS) ? lds_load_i32(&s_wprog[lane]) : need;
                        if (__all(pr >= need)) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                const long long t_wait = d.trace ? (long long)__builtin_readcyclecounter() : 0;
                int iters = 0;
                const unsigned long long rem = s_removed[b] | acc[0];
                const int p = b * 64 + lane;
                const bool alive = (p < n) && !((rem >> lane) & 1ull);
                const unsigned long long col = ring[s][0];
                unsigned long long K = __ballot(alive);
                // K_{t+1} = { alive j : no i in K_t suppresses j }.  Box b*64 has no predecessor in
                // the block, so index k is final after k+1 steps; the fixed point is the greedy set.
                for (;;) {
                    const unsigned long long K2 = __ballot(alive && !(col & K));
                    ++iters;
                    if (K2 == K) break;
                    K = K2;
                }
                if (d.trace && f == 0 && lane == 0) {
                    long long *tr = d.trace + 4 * b;
                    tr[0] = t_begin; tr[1] = t_wait; tr[2] = (long long)__builtin_readcyclecounter();
                    tr[3] = ((long long)iters << 32) | (unsigned)__popcll(K);
                }
                const bool kept = (K >> lane) & 1ull;
                const int pos = total + __popcll(K & ((1ull << lane) - 1ull));
                if (kept && (d.max_keep <= 0 || pos < d.max_keep)) keep[pos] = p;
                total += __popcll(K);
                if (lane == 0) s_kept[b] = K;
                lds_store_i32(&s_chain_pos, b + 1);
                if (d.max_keep > 0 && total >= d.max_keep) { stop = true; break; }
                // near band: K_b -> removed bits of blocks b+1 .. b+NMS_BAND-1
#pragma unroll
                for (int k = 1; k < NMS_BAND; ++k) acc[k] |= __ballot((ring[s][k] & K) != 0ull);
#pragma unroll
                for (int k = 0; k < NMS_BAND - 1; ++k) acc[k] = acc[k + 1];
                acc[NMS_BAND - 1] = 0ull;
                // refill this ring slot for block b + NMS_PF (static addresses: runs ahead of the chain)
                const int bn = b + NMS_PF;
                if (bn < nb) {
#pragma unroll
                    for (int k = 0; k < NMS_BAND; ++k) ring[s][k] = band[((long long)bn * NMS_BAND + k) * 64 + lane];
                }
            }
        }
        if (lane == 0) s_total = total;
        lds_store_i32(&s_done, 1);
    } else {
        // ------------------------------------------------------------------ bulk workers
        // Group g trails the chain on the blocks b = g (mod NMS_GROUPS); its NMS_GW waves split the
        // 64 rows.  All loads of a (block, pass) are in flight together: 16 bytes = 2 words per lane,
        // 128 words per pass.  s_wprog[me] = first block this wave has NOT finished its share of
        // (blocks of other groups count as finished).
        const int me = wave - 1, g = me / NMS_GW, lw = me % NMS_GW;
        lds_store_i32(&s_wprog[me], g);
        int b = g;
        for (; b + NMS_BAND < nb; b += NMS_GROUPS) {
            bool quit = false;
            for (;;) {
                if (lds_load_i32(&s_chain_pos) > b) break;
                if (lds_load_i32(&s_done)) { quit = true; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            if (quit) break;
            const unsigned long long K = s_kept[b];
            const int w0 = b + NMS_BAND;               // first far word of this block's rows
            for (int wb = (w0 & ~1); wb < nb; wb += 128) {
                const int w = wb + 2 * lane;            // this lane's two words: w, w + 1
                ulonglong2 a[NMS_ROWS_PER_WORKER];
#pragma unroll
                for (int r = 0; r < NMS_ROWS_PER_WORKER; ++r) {
                    const int i = lw + NMS_GW * r;
                    const bool on = (i < 64) && ((K >> (i & 63)) & 1ull) && (w < nb);
                    const ulonglong2 *row = reinterpret_cast<const ulonglong2 *>(mask + (long long)(b * 64 + (i & 63)) * d.nbs + w);
                    a[r] = on ? *row : make_ulonglong2(0ull, 0ull);
                }
                ulonglong2 v = make_ulonglong2(0ull, 0ull);
#pragma unroll
                for (int r = 0; r < NMS_ROWS_PER_WORKER; ++r) { v.x |= a[r].x; v.y |= a[r].y; }
                if (w >= w0 && w < nb && v.x) atomicOr(&s_removed[w], v.x);        // w0 - 1 is a band slot: skip
                if (w + 1 < nb && v.y) atomicOr(&s_removed[w + 1], v.y);
            }
            lds_store_i32(&s_wprog[me], b + NMS_GROUPS);
        }
        lds_store_i32(&s_wprog[me], NMS_MAX_WORDS * 2);   // nothing left that the chain could wait for
    }
    __syncthreads();
    int nk = s_total;
    if (d.max_keep > 0 && nk > d.max_keep) nk = d.max_keep;
    if (threadIdx.x == 0) d.num_keep[f] = nk;
    if (d.emit.enabled) {
        // proposal_layer_tf.py:188-191: the three ROI blobs, batch column = frame index
        const EmitDev &e = d.emit;
        if (threadIdx.x == 0) e.num_out[f] = nk;
        for (int r = threadIdx.x; r < e.cap; r += blockDim.x) {
            float *obv = e.blob_bv + ((long long)f * e.cap + r) * 5;
            float *oim = e.blob_img + ((long long)f * e.cap + r) * 5;
            float *o3 = e.blob_3d + ((long long)f * e.cap + r) * 7;
            if (r < nk) {
                const int c = e.order[(long long)f * e.order_cap + keep[r]];
                const long long o = (long long)f * e.N + c;
                const float4 bx = e.bv[o];
                const int4 im = e.img[o];
                const float bi = (float)f;
                obv[0] = bi; obv[1] = bx.x; obv[2] = bx.y; obv[3] = bx.z; obv[4] = bx.w;
                oim[0] = bi; oim[1] = (float)im.x; oim[2] = (float)im.y; oim[3] = (float)im.z; oim[4] = (float)im.w;
                o3[0] = bi;
#pragma unroll
                for (int j = 0; j < 6; ++j) o3[1 + j] = e.p3[o * 6 + j];
            } else {
#pragma unroll
                for (int j = 0; j < 5; ++j) { obv[j] = 0.0f; oim[j] = 0.0f; }
#pragma unroll
                for (int j = 0; j < 7; ++j) o3[j] = 0.0f;
            }
        }
    }
}

size_t mv3d_nms_ws_bytes(int n_cap, int batch)
{
    const size_t nbw = (size_t)(n_cap + 63) / 64, nbs = (nbw + 1) & ~(size_t)1;
    const size_t rows = nbw * 64;
    return (size_t)batch * (mv3d_align_up(rows * nbs * 8) + mv3d_align_up(nbw * NMS_BAND * 64 * 8)) + MV3D_ALIGN;
}

int mv3d_launch_nms(const NmsLaunch &L, hipStream_t stream)
{
    if (L.n_cap < 0 || L.batch <= 0) return MV3D_ERR_INVALID_ARG;
    const int nbw = (L.n_cap + 63) / 64;
    if (nbw > NMS_MAX_WORDS) return MV3D_ERR_INVALID_ARG;
    NmsDev d;
    d.boxes = L.boxes; d.box_stride = L.box_stride; d.boxes_frame_stride = L.boxes_frame_stride;
    d.idx = L.idx; d.idx_frame_stride = L.idx_frame_stride; d.n_dev = L.n_dev; d.n_cap = L.n_cap;
    d.nbw = nbw; d.nbs = (nbw + 1) & ~1; d.tf = L.strict_gt ? nextafterf(L.thresh_f32, INFINITY) : L.thresh_f32; d.max_keep = L.max_keep;
    d.fast_ok = (d.tf >= 0x1p-10f && d.tf <= 0x1p10f) ? 1 : 0;
    d.neg_h = d.fast_ok ? -0.5f * (d.tf - nextafterf(d.tf, 0.0f)) : 0.0f;
    const size_t rows = (size_t)nbw * 64;
    d.mask = (unsigned long long *)L.workspace;
    d.band = (unsigned long long *)((char *)L.workspace + (size_t)L.batch * mv3d_align_up(rows * (size_t)d.nbs * 8));
    d.keep = L.keep; d.keep_frame_stride = L.keep_frame_stride; d.num_keep = L.num_keep; d.status = L.status;
    d.emit = L.emit;
    d.trace = L.trace;
    if (nbw > 0) {
        dim3 grid(nbw * (nbw + 1) / 2, 1, L.batch);
        hipLaunchKernelGGL(nms_mask_kernel, grid, dim3(64), 0, stream, d);
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

static int nms_device_impl(const float *dets_dev, int n, double thresh, int max_keep, int32_t *keep_dev,
                           int32_t *num_keep_dev, int32_t *status_dev, void *workspace, size_t workspace_bytes,
                           void *stream, long long *trace)
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
    L.trace = trace;
    return mv3d_launch_nms(L, s);
}

extern "C" int mv3d_nms_device(const float *dets_dev, int n, double thresh, int max_keep, int32_t *keep_dev,
                               int32_t *num_keep_dev, int32_t *status_dev, void *workspace,
                               size_t workspace_bytes, void *stream)
{
    return nms_device_impl(dets_dev, n, thresh, max_keep, keep_dev, num_keep_dev, status_dev, workspace,
                           workspace_bytes, stream, nullptr);
}

extern "C" int mv3d_nms_device_trace(const float *dets_dev, int n, double thresh, int max_keep, int32_t *keep_dev,
                                     int32_t *num_keep_dev, int32_t *status_dev, void *workspace,
                                     size_t workspace_bytes, void *stream, int64_t *trace_dev)
{
    return nms_device_impl(dets_dev, n, thresh, max_keep, keep_dev, num_keep_dev, status_dev, workspace,
                           workspace_bytes, stream, (long long *)trace_dev);
}

// keys of a (n,5) dets array for the device sort of mv3d_nms_host
__global__ void nms_score_keys_kernel(const float *dets, int n, int key_stride, uint32_t *keys)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < key_stride) keys[i] = (i < n) ? mv3d_score_key(dets[5 * i + 4]) : 0u;
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
    const int kstride = mv3d_rank_key_stride(n);
    const size_t o_dets = 0, o_keys = mv3d_align_up((size_t)n * 20), o_order = o_keys + mv3d_align_up((size_t)kstride * 4),
                 o_keep = o_order + mv3d_align_up((size_t)n * 4), o_out = o_keep + mv3d_align_up((size_t)n * 4),
                 o_cnt = o_out + mv3d_align_up((size_t)n * 4), o_rank = o_cnt + MV3D_ALIGN,
                 o_ws = o_rank + (presorted ? 0 : mv3d_rank_ws_bytes(n, 1));
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
            hipLaunchKernelGGL(nms_score_keys_kernel, dim3(kstride / 256), dim3(256), 0, s, dets, n, kstride, keys);
            if ((rc = mv3d_launch_rank(keys, n, kstride, 1, order, n, nullptr, 0, nullptr, buf + o_rank, s)) != MV3D_OK) break;
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
