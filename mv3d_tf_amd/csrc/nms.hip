// Greedy IoU NMS for gfx950 (wave64).
//
// Replaces lib/nms/cpu_nms.pyx:17-68 (== lib/utils/nms.pyx:17-68) and the CUDA path
// lib/nms/nms_kernel.cu:34-144.  The suppression matrix is cut into 64x64 tiles of its upper
// triangle, enumerated column-block-major (tile (rb, cb) at cb(cb+1)/2 + rb), every tile stored in
// COLUMN form: lane j of the tile holds the u64 "which rows of row block rb suppress box cb*64+j".
// The work proceeds in ROUNDS over ranges of column blocks ([0,32), [32,nbw), or narrower ones for a large max_keep):
// the greedy pass almost always reaches post_nms_topN inside the first range, and a round whose
// frame is already finished returns at once, so the tiles nobody will read are never computed
// (a TEST-config frame needs ~15 of its 94 column blocks: 528 tiles instead of 4465).
//
//  nms_tiles_kernel     round 1, all CUs, one workgroup per tile (four waves, a quarter of the rows each).  lane = column box
//        (registers), the 64 row boxes staged in LDS and read back as broadcasts; packed-f32 math and
//        a division-free exact compare (tile_fast()); the lane's word is a plain accumulation of bits.
//  nms_chain_lds_kernel round 1, one workgroup per frame: the greedy dependency.  Loader waves stream the
//        round's tiles (one contiguous run in consumption order) into LDS, helper waves fold the older
//        row blocks as their K words appear, one wave runs the serial part (see the kernel).
//  nms_round_kernel     a later round in one launch: tile phase on all CUs (tiles of row blocks that
//        earlier rounds finished are reduced on the spot to "removed" bits, the round's own tiles are
//        stored), then the workgroup that finishes last runs the round's chain (chain1_round()).
// State (K_b, kept count, done flag) persists in the workspace between rounds.  The round that finishes
// the frame (max_keep kept boxes = the reference's keep[:post_nms_topN], or the last block) also
// gathers the ROI blobs of proposal_layer_3d.
//
// Arithmetic is the reference's, operation for operation (see pair_suppresses()).
#include <math.h>
#include "kernels.h"

#define NMS_MAX_WORDS 512   // up to 32768 boxes per frame (a full 76x76x4 grid is 23104)
#define LDS_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local")
#define LDS_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local")

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
    int ntiles;                  // nbw (nbw + 1) / 2
    int b0, b1;                  // column-block range of this round
    int first_round;
    float tf;
    float neg_h;                 // -(half the gap below tf), see tile_fast()
    int fast_ok;                 // tf is a positive normal f32 in a range where tile_fast() is valid
    int max_keep;
    unsigned long long *tiles;   // (batch, ntiles, 64) column form
    unsigned long long *kstate;  // (batch, nbw) kept masks of finished blocks
    unsigned long long *rem;     // (batch, nbw) boxes removed by blocks of EARLIER rounds (later rounds only)
    int32_t *cstate;             // (batch, 4): [0] kept so far, [1] done
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
                                                        const int lane, float &min_den, bool &undecided,
                                                        const int c_begin = 0, const int c_end = 64)
{
    unsigned long long word = 0;
    bool amb = false;
    const f2 ntf = {-tf, -tf}, nh = {neg_h, neg_h}, one = {1.0f, 1.0f}, la = {larea, larea};
    // 4 chunks of 16 columns: the chunk body is straight-line (constant shifts, loads hoisted), the
    // chunk loop is kept rolled so that the live set stays well inside 128 VGPRs
#pragma unroll 1
    for (int c = c_begin; c < c_end; c += 16) {
        unsigned bits = 0;
#pragma unroll
        for (int u = 0; u < 16; u += 2) {
            const int j = c + u;
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
            bits |= p0 ? (1u << u) : 0u;
            bits |= p1 ? (2u << u) : 0u;
        }
        word |= (unsigned long long)bits << c;
    }
    undecided = amb;
    return word;
}

// One tile by one wave.  `s_box/s_area` are the wave's own 64-entry LDS staging arrays.  AGENT_STORE: the
// word is written through at agent scope (read later in the SAME kernel by another workgroup).
// PARTS = 2 / 4: the wave computes only rows [64/PARTS * part, +64/PARTS) of the tile and stores that 32- / 16-bit
// piece of every word (several waves per tile: the round-1 tile kernel is bound by the ~1600 VALU instructions of
// a tile; 6.4 us with one wave per tile, ~3.5 us with four).
template <bool AGENT_STORE, int PARTS = 1>
__device__ __forceinline__ void nms_one_tile(const NmsDev &d, const int f, const int t, const int lane, float4 *s_box,
                                             float *s_area, const int part = 0)
{
    const int r_begin = (64 / PARTS) * part, r_end = r_begin + 64 / PARTS;
    const int n = frame_n(d, f);
    // tile t -> (cb, rb <= cb): t = cb(cb+1)/2 + rb
    int cb = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while ((cb + 1) * (cb + 2) / 2 <= t) ++cb;
    while (cb * (cb + 1) / 2 > t) --cb;
    const int rb = t - cb * (cb + 1) / 2;
    if (cb * 64 >= n) return;
    // LDS side = the 64 ROW boxes of block rb; lane = COLUMN box cb*64 + lane.  The predicate is
    // symmetric in the two boxes (max/min/+/* commute), so this is the transposed tile.
    const int ridx = rb * 64 + lane, cidx = cb * 64 + lane;
    float4 sb = make_float4(NAN, NAN, NAN, NAN);            // NaN box: every predicate false
    if (ridx < n) sb = load_box(d, f, ridx);
    s_box[lane] = sb;
    s_area[lane] = ((sb.z - sb.x) + 1.0f) * ((sb.w - sb.y) + 1.0f);         // cpu_nms.pyx:24
    const bool lds_tame = __all(tame(sb));                  // false if the block is ragged (NaN padding)
    LDS_RELEASE();                                          // one wave: its LDS stores are in order
    float4 lb = make_float4(NAN, NAN, NAN, NAN);
    if (cidx < n) lb = load_box(d, f, cidx);
    const float larea = ((lb.z - lb.x) + 1.0f) * ((lb.w - lb.y) + 1.0f);
    const bool diag = (rb == cb);
    unsigned long long word = 0;
    bool any_zero = false, done = false;
    if (d.fast_ok && lds_tame && __all(tame(lb))) {
        float min_den = 1.0f;
        bool und;
        if (diag) word = tile_fast<true>(lb, larea, s_box, s_area, d.tf, d.neg_h, lane, min_den, und, r_begin, r_end);
        else word = tile_fast<false>(lb, larea, s_box, s_area, d.tf, d.neg_h, lane, min_den, und, r_begin, r_end);
        // a denominator that is not safely positive, or an undecided compare: redo the tile exactly
        done = !__any(und || !(min_den >= 0x1p-20f));
    }
    if (!done) {
        // exact path: ragged blocks, NaN / huge coordinates, zero / negative unions, undecided compares
        word = 0;
        for (int j = r_begin; j < r_end; ++j) {
            const float4 q = s_box[j];
            bool zd;
            // (kept box i = row j of the LDS side, later box = my column), the argument order of cpu_nms.pyx
            bool p = pair_suppresses(q.x, q.y, q.z, q.w, s_area[j], lb.x, lb.y, lb.z, lb.w, larea, d.tf, zd);
            const bool live = diag ? (j < lane) : true;                       // rows before my column
            p = p && live;
            any_zero |= (zd && live && cidx < n && (rb * 64 + j) < n);
            word |= (unsigned long long)p << j;
        }
    }
    unsigned long long *dst = &d.tiles[((long long)f * d.ntiles + t) * 64 + lane];
    if (AGENT_STORE && rb < d.b0) {
        // the row block was finished in an earlier round: its K word is final, so the tile reduces on the
        // spot to "which boxes of column block cb does it remove" and never travels to the chain
        const unsigned long long K = d.kstate[(long long)f * d.nbw + rb];
        const unsigned long long m = __ballot((word & K) != 0ull);
        if (m && lane == 0) __hip_atomic_fetch_or(&d.rem[(long long)f * d.nbw + cb], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (AGENT_STORE) __hip_atomic_store(dst, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (PARTS == 2) reinterpret_cast<unsigned *>(dst)[part] = (unsigned)(word >> r_begin);
    else if (PARTS == 4) reinterpret_cast<unsigned short *>(dst)[part] = (unsigned short)(word >> r_begin);
    else *dst = word;
    if (d.status && __any(any_zero) && lane == 0) atomicOr(&d.status[f], MV3D_FLAG_ZERO_DIVISION);
}

// grid: (tiles of the round, 1, batch); block 256 = four waves per tile, each with its own staging arrays and its
// own quarter of the rows (no barrier between them).
__global__ __launch_bounds__(256) void nms_tiles_kernel(NmsDev d)
{
    __shared__ float4 s_box[4][64];
    __shared__ float s_area[4][64];
    const int f = blockIdx.z;
    if (!d.first_round && d.cstate[4 * f + 1]) return;       // frame already finished in an earlier round
    const int h = threadIdx.x >> 6;
    nms_one_tile<false, 4>(d, f, d.b0 * (d.b0 + 1) / 2 + blockIdx.x, threadIdx.x & 63, s_box[h], s_area[h], h);
}

// The chain of a LATER round (columns [b0, b1), b1 - b0 <= KMAX), run by one workgroup: the greedy
// dependency in "pull" form.  removed(j) of block b = remv[b] (everything the blocks of earlier rounds
// remove, reduced by the tile phase) | OR over the round's own row blocks rb in [b0, b) of
// (tile(rb,b)[j] & K_rb) != 0 -- the wave pulls those tiles itself (registers, refilled one column ahead,
// static addresses) -- then it resolves the diagonal tile by fixed-point iteration on ballots (converges
// to the unique greedy set).  State (K_b, kept count, done flag) persists in the workspace between
// rounds; the round that finishes the frame (max_keep kept boxes = the reference's
// keep[:post_nms_topN], or the last block) also gathers the ROI blobs of proposal_layer_3d.
template <int KMAX, int THREADS>
__device__ __forceinline__ void chain1_round(const NmsDev &d, const int f)
{
    __shared__ unsigned long long s_K[NMS_MAX_WORDS];
    __shared__ int s_fin, s_nk, s_over, s_sink;
    int32_t *cstate = d.cstate + 4 * f;
    const int n = frame_n(d, f);
    const int nb = (n + 63) >> 6;
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0) s_over = 0;
    __syncthreads();
    if (threadIdx.x < 64) {
    const unsigned long long *tiles = d.tiles + (long long)f * d.ntiles * 64;
    unsigned long long *kstate = d.kstate + (long long)f * d.nbw;
    const unsigned long long *remv = d.rem + (long long)f * d.nbw;
    int32_t *keep = d.keep + (long long)f * d.keep_frame_stride;
    const int b0 = d.b0, b1 = min(d.b1, nb);
    for (int w = lane; w < NMS_MAX_WORDS; w += 64) s_K[w] = 0ull;          // K of the round's blocks, by absolute index
    int total = cstate[0];
    bool stop = false;
    // cur[q] = tile(b0 + q, b): the rows of THIS round only (earlier rounds arrive reduced in remv)
    unsigned long long cur[KMAX], dcur = 0ull, rcur = 0ull;
#pragma unroll
    for (int q = 0; q < KMAX; ++q) cur[q] = 0ull;
    if (b0 < b1) {
        dcur = tiles[((long long)b0 * (b0 + 1) / 2 + b0) * 64 + lane];
        rcur = __hip_atomic_load(&remv[b0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");   // s_K visible to every lane's reads
    for (int b = b0; b < b1; ++b) {
        const long long t_begin = d.trace ? (long long)__builtin_readcyclecounter() : 0;
        unsigned long long rem = rcur;
        const int bu = __builtin_amdgcn_readfirstlane(b);     // scalar: whole chunks of 8 tiles are branched over
#pragma unroll
        for (int c = 0; c < KMAX / 8; ++c) {
            if (8 * c < bu - b0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int q = 8 * c + u;                  // cur[q] == 0 for b0 + q >= b (never refilled)
                    rem |= __ballot((cur[q] & s_K[b0 + q]) != 0ull);
                }
            }
        }
        const unsigned long long dg = dcur;
        if (bu + 1 < b1) {                                    // refill with column b + 1 (static addresses)
            const long long base = (long long)(bu + 1) * (bu + 2) / 2;
            dcur = tiles[(base + bu + 1) * 64 + lane];
            rcur = __hip_atomic_load(&remv[bu + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int c = 0; c < KMAX / 8; ++c) {
                if (8 * c < bu + 1 - b0) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int q = 8 * c + u;
                        cur[q] = (q < bu + 1 - b0) ? tiles[(base + b0 + q) * 64 + lane] : 0ull;
                    }
                }
            }
        }
        const long long t0 = d.trace ? (long long)__builtin_readcyclecounter() : 0;
        const int p = b * 64 + lane;
        const bool alive = (p < n) && !((rem >> lane) & 1ull);
        unsigned long long K = __ballot(alive);
        int iters = 0;
        // K_{t+1} = { alive j : no i in K_t suppresses j }.  Box b*64 has no predecessor in the block, so index k
        // is final after k+1 steps; the fixed point is the greedy set.
        for (;;) {
            const unsigned long long K2 = __ballot(alive && !(dg & K));
            ++iters;
            if (K2 == K) break;
            K = K2;
        }
        const bool kept = (K >> lane) & 1ull;
        const int pos = total + __popcll(K & ((1ull << lane) - 1ull));
        if (kept && (d.max_keep <= 0 || pos < d.max_keep)) keep[pos] = p;
        total += __popcll(K);
        if (lane == 0) { s_K[b] = K; kstate[b] = K; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        if (d.trace && f == 0 && lane == 0) {
            long long *tr = d.trace + 4 * b;
            tr[0] = t_begin; tr[1] = t0; tr[2] = (long long)__builtin_readcyclecounter();
            tr[3] = ((long long)iters << 32) | (unsigned)__popcll(K);
        }
        if (d.max_keep > 0 && total >= d.max_keep) { stop = true; break; }
    }
    const bool finished = stop || (b1 >= nb);
    int nk = total;
    if (d.max_keep > 0 && nk > d.max_keep) nk = d.max_keep;
    if (lane == 0) {
        cstate[0] = total; cstate[1] = finished ? 1 : 0;
        s_fin = finished ? 1 : 0; s_nk = nk;
        if (finished) d.num_keep[f] = nk;
        __hip_atomic_store(&s_over, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    } else {
        // the other waves run ahead of wave 0 and touch the round's own tiles, column by column, one load
        // per 128-B line: the words were written through by other XCDs, a first touch costs a full memory
        // latency (~2 us) that wave 0's one-column-ahead refill cannot hide; an L2 hit costs a fraction
        const int nb2 = (frame_n(d, f) + 63) >> 6;
        const int b0 = d.b0, b1 = min(d.b1, nb2);
        const char *tl = reinterpret_cast<const char *>(d.tiles + (long long)f * d.ntiles * 64);
        const int lt = threadIdx.x - 64, nt = blockDim.x - 64;
        unsigned sink = 0;
        for (int b = b0; b < b1; ++b) {
            const long long colbase = ((long long)b * (b + 1) / 2 + b0) * 512;   // tile (b0, b)
            const int lines = (b - b0 + 1) * 4;
            for (int l = lt; l < lines; l += nt) sink ^= *reinterpret_cast<const unsigned *>(tl + colbase + (long long)l * 128);
            if (__hip_atomic_load(&s_over, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;   // wave 0 is through
        }
        if (sink == 0x9e3779b9u) __hip_atomic_store(&s_sink, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // keeps the loads alive
    }
    __syncthreads();                                          // also orders wave 0's keep[] stores for the readers
    if (!s_fin) return;
    const int nk = s_nk;
    const int32_t *keep = d.keep + (long long)f * d.keep_frame_stride;
    if (d.emit.enabled) {
        // proposal_layer_tf.py:188-191: the three ROI blobs, batch column = frame index
        const EmitDev &e = d.emit;
        if (threadIdx.x == 0) e.num_out[f] = nk;
        // one workgroup gathers up to cap (TRAIN: 2000) rows: eight rows per thread at a time, every level
        // of the dependent chain keep -> order -> record issued for all eight before the next level
        constexpr int EM = 8;
        for (int r0 = 0; r0 < e.cap; r0 += EM * blockDim.x) {
            int kp[EM], c[EM];
#pragma unroll
            for (int u = 0; u < EM; ++u) { const int r = r0 + u * blockDim.x + threadIdx.x; kp[u] = keep[min(r, max(nk - 1, 0))]; }
#pragma unroll
            for (int u = 0; u < EM; ++u) c[u] = e.order[(long long)f * e.order_cap + (nk > 0 ? kp[u] : 0)];
            float4 bx[EM];
            int4 im[EM];
            float2 pa[EM], pb[EM], pc[EM];
#pragma unroll
            for (int u = 0; u < EM; ++u) {
                const long long o = (long long)f * e.N + c[u];
                bx[u] = e.bv[o];
                im[u] = e.img[o];
                pa[u] = *reinterpret_cast<const float2 *>(e.p3 + o * 6);
                pb[u] = *reinterpret_cast<const float2 *>(e.p3 + o * 6 + 2);
                pc[u] = *reinterpret_cast<const float2 *>(e.p3 + o * 6 + 4);
            }
#pragma unroll
            for (int u = 0; u < EM; ++u) {
                const int r = r0 + u * blockDim.x + threadIdx.x;
                if (r >= e.cap) continue;
                float *obv = e.blob_bv + ((long long)f * e.cap + r) * 5;
                float *oim = e.blob_img + ((long long)f * e.cap + r) * 5;
                float *o3 = e.blob_3d + ((long long)f * e.cap + r) * 7;
                if (r < nk) {
                    const float bi = (float)f;
                    obv[0] = bi; obv[1] = bx[u].x; obv[2] = bx[u].y; obv[3] = bx[u].z; obv[4] = bx[u].w;
                    oim[0] = bi; oim[1] = (float)im[u].x; oim[2] = (float)im[u].y; oim[3] = (float)im[u].z; oim[4] = (float)im[u].w;
                    o3[0] = bi; o3[1] = pa[u].x; o3[2] = pa[u].y; o3[3] = pb[u].x; o3[4] = pb[u].y; o3[5] = pc[u].x; o3[6] = pc[u].y;
                } else {
#pragma unroll
                    for (int q = 0; q < 5; ++q) { obv[q] = 0.0f; oim[q] = 0.0f; }
#pragma unroll
                    for (int q = 0; q < 7; ++q) o3[q] = 0.0f;
                }
            }
        }
    }
}

// A later round in ONE launch (grid: (<= 256, 1, batch); block 256 = four tiles at a time, grid-stride):
// a frame that is already finished costs one flag load per workgroup; otherwise the workgroup that
// finishes last (ticket in cstate[2]) runs the round's chain.  The tile words are written through at
// agent scope and complete (vmcnt 0) before the ticket is taken; the chain side starts with an
// agent-scope acquire, so it reads what the other XCDs wrote.
template <int KMAX>
__global__ __launch_bounds__(256) void nms_round_kernel(NmsDev d)
{
    __shared__ float4 s_box[4][64];
    __shared__ float s_area[4][64];
    __shared__ int s_last;
    const int f = blockIdx.z;
    int32_t *cstate = d.cstate + 4 * f;
    if (cstate[1]) return;                                   // finished in an earlier round
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ntr = d.b1 * (d.b1 + 1) / 2 - d.b0 * (d.b0 + 1) / 2;
    for (int tl = blockIdx.x * 4 + wave; tl < ntr; tl += gridDim.x * 4) {
        nms_one_tile<true>(d, f, d.b0 * (d.b0 + 1) / 2 + tl, lane, s_box[wave], s_area[wave]);
        LDS_RELEASE();                                        // the staging arrays are reused by the next tile
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's tile words have reached memory
    __syncthreads();
    if (threadIdx.x == 0)
        s_last = (__hip_atomic_fetch_add(&cstate[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1);
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) cstate[2] = 0;                      // ready for the next launch on this workspace
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    chain1_round<KMAX, 256>(d, f);
}

// Round-1 form of the chain (column blocks [0, b1 <= 32)).  A single wave executes ~1 instruction per
// 8-10 cycles, so the serial part is cut to the bone and everything else runs beside it, in one
// workgroup of 8 waves with three roles that talk through LDS only:
//   loaders (waves 5..7)  the tiles of the round are ONE contiguous stream in consumption order
//        (column-block-major): stream it into LDS, 16-B loads, two batches (48 KB) in flight.
//   helpers (waves 1..4)  column c (c mod 4 = helper) : acc_c[j] = OR over row blocks rb <= c-3 of
//        tile(rb,c)[j] & K_rb, consumed as the K_rb are published; parked in an 8-slot LDS ring.
//   wave 0                the greedy dependency proper: acc_c | the two youngest tiles & K, one compare,
//        the diagonal fixed point, publish K_c.  ~45 instructions per 64-box block.
// The stream is cut into epochs that fit the LDS arena ([0,23) and [23,32)); kept positions are
// expanded from the K words by all waves afterwards, and the sorted `order` of the round's boxes is
// staged in LDS so that the ROI-blob gather at the end is one dependent load deep.
#define CHL_THREADS 512
#define CHL_HELPERS 4                                   // waves 1..4
#define CHL_LOADER0 5                                   // waves 5..7
#define CHL_LOADERS 3
#define CHL_ARENA 231                                   // tiles (columns 0..20): 138 KB of LDS in all, so that the workgroup
                                                        // fits on a CU next to the small-LDS workgroups of another stream
#define CHL_PER (CHL_LOADERS * 64)                      // 16-B units per load instruction of the loader group
#define CHL_LU 8                                        // loads per loader thread and batch
#define CHL_BATCH (CHL_LU * CHL_PER)                    // units per batch (48 tiles, 24 KB)
#define CHL_BOXES 2048
#define CHL_POST ((1 + CHL_HELPERS) * 64)                // threads of the post-processing (waves 0..4)

__device__ __forceinline__ int lds_ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// barrier among the waves that outlive the loaders: arrive on an LDS counter, wait for `target` arrivals
__device__ __forceinline__ void chl_sync(int *ctr, const int target, const int lane)
{
    LDS_RELEASE();
    if (lane == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (lds_ld(ctr) < target) {}
    LDS_ACQUIRE();
}

// units of the stream known to be in LDS (min over the loader waves), refreshed until >= need
__device__ __forceinline__ bool chl_wait_units(int &have, const int need, const int *s_prog, const int *s_halt, const int lane,
                                               const bool may_halt)
{
    while (have < need) {
        int v = (lane < CHL_LOADERS) ? lds_ld(&s_prog[CHL_LOADER0 + lane]) : 0x7fffffff;
        v = min(v, __shfl_xor(v, 1));
        v = min(v, __shfl_xor(v, 2));
        have = __builtin_amdgcn_readfirstlane(v);
        if (have < need) {
            if (may_halt && lds_ld(s_halt)) return false;
            __builtin_amdgcn_s_sleep(1);
        }
    }
    LDS_ACQUIRE();
    return true;
}

// epoch = columns [c0, c1): as many whole columns as fit the arena
__device__ __forceinline__ int chl_epoch_end(const int c0, const int cols)
{
    const int T0 = c0 * (c0 + 1) / 2;
    int c1 = c0;
    while (c1 < cols && (c1 + 1) * (c1 + 2) / 2 - T0 <= CHL_ARENA) ++c1;
    return c1;
}

#define CHL_REP(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define CHL_DECL(q) uint4 ra##q, rb##q;
#define CHL_LDA0(q) ra##q = src[min(lt + q * CHL_PER, last)];
#define CHL_LDB(q) rb##q = src[min(v + q * CHL_PER, last)];
#define CHL_LDA(q) ra##q = src[min(v + q * CHL_PER, last)];
#define CHL_STA(q) if (u + q * CHL_PER < units) tiles_lds[u + q * CHL_PER] = ra##q;
#define CHL_STB(q) if (u + q * CHL_PER < units) tiles_lds[u + q * CHL_PER] = rb##q;

__global__ __launch_bounds__(CHL_THREADS) void nms_chain_lds_kernel(NmsDev d)
{
    __shared__ uint4 s_tiles[(CHL_ARENA + 3) * 32];      // 512 B per tile
    __shared__ unsigned long long s_K[32];
    __shared__ unsigned long long s_acc[8 * 64];          // helper results, ring over columns
    __shared__ unsigned long long s_junk[64];             // where lanes 1..63 of a "lane 0 only" store go
    __shared__ int s_accflag[32];
    __shared__ int s_order[CHL_BOXES];
    __shared__ int s_prog[8];
    __shared__ int s_ready, s_halt, s_stop, s_sync;
    const int f = blockIdx.x;
    const long long t_start = d.trace ? (long long)__builtin_readcyclecounter() : 0;
    long long t_first = 0;
    int32_t *cstate = d.cstate + 4 * f;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint4 *gsrc = reinterpret_cast<const uint4 *>(d.tiles + (long long)f * d.ntiles * 64);
    int32_t *keep = d.keep + (long long)f * d.keep_frame_stride;
    // loaders and helpers work from the static capacity (tiles beyond the frame's boxes are never consumed)
    const int cols = min(d.b1, d.nbw);
    const bool loader = wave >= CHL_LOADER0;
    const int lt = threadIdx.x - CHL_LOADER0 * 64;
    int c0 = 0, c1 = chl_epoch_end(0, cols);
    // the first batch of the first epoch (and the sorted order of the round's boxes) is requested before
    // anything else happens in the workgroup
    CHL_REP(CHL_DECL)
    int o0 = 0, o1 = 0, o2 = 0, o3 = 0, o4 = 0, o5 = 0, o6 = 0, o7 = 0, o8 = 0, o9 = 0, o10 = 0;
    const bool want_order = d.emit.enabled;
    if (loader && cols > 0) {
        const uint4 *src = gsrc;
        const int last = (c1 * (c1 + 1) / 2) * 32 - 1;
        CHL_REP(CHL_LDA0)
        if (want_order) {
            const int32_t *ord = d.emit.order + (long long)f * d.emit.order_cap;
            const int m = min(d.n_cap, CHL_BOXES) - 1;
            o0 = ord[min(lt, m)]; o1 = ord[min(lt + CHL_PER, m)]; o2 = ord[min(lt + 2 * CHL_PER, m)];
            o3 = ord[min(lt + 3 * CHL_PER, m)]; o4 = ord[min(lt + 4 * CHL_PER, m)]; o5 = ord[min(lt + 5 * CHL_PER, m)];
            o6 = ord[min(lt + 6 * CHL_PER, m)]; o7 = ord[min(lt + 7 * CHL_PER, m)]; o8 = ord[min(lt + 8 * CHL_PER, m)];
            o9 = ord[min(lt + 9 * CHL_PER, m)]; o10 = ord[min(lt + 10 * CHL_PER, m)];
        }
    }
    if (threadIdx.x < 32) { s_K[threadIdx.x] = 0ull; s_accflag[threadIdx.x] = 0; }
    if (threadIdx.x >= 32 && threadIdx.x < 40) s_prog[threadIdx.x - 32] = 0;
    if (threadIdx.x == 40) { s_stop = 0; s_ready = 0; s_halt = 0; s_sync = 0; }
    __syncthreads();
    // (a scalar load: every wave may ask, only wave 0 uses it)
    const int n = __builtin_amdgcn_readfirstlane(frame_n(d, f)), nb = (n + 63) >> 6, b1 = min(d.b1, nb);
    // three tiles of front padding: the "three youngest row blocks" of columns 0..2 read (and mask out) what lies before
    uint4 *const tiles_lds = s_tiles + 3 * 32;
    const unsigned long long *arena = reinterpret_cast<const unsigned long long *>(tiles_lds);
    int total = 0;                                         // wave 0
    unsigned long long Kp1 = 0ull, Kp2 = 0ull, Kp3 = 0ull; // K of blocks b-1, b-2, b-3 (wave 0)
    bool stop = false, frame_done = false;
    while (c0 < cols) {
        const int T0 = c0 * (c0 + 1) / 2;
        const int units = ((c1 * (c1 + 1)) / 2 - T0) * 32;
        if (loader) {
            // ---- loaders: two batches of eight 16-B loads in flight per thread; addresses are clamped so
            // that every load is unconditional (no divergent control flow around the loads)
            const uint4 *src = gsrc + (long long)T0 * 32;
            const int last = units - 1;
            const int nbatch = (units + CHL_BATCH - 1) / CHL_BATCH;
            for (int k = 0; k < nbatch; k += 2) {
                {   // batch k from A while batch k+1 flies into B
                    const int u = k * CHL_BATCH + lt, v = u + CHL_BATCH;
                    CHL_REP(CHL_LDB)
                    CHL_REP(CHL_STA)
                    // the sorted order of the round's boxes is staged BEFORE the first progress word is published: the
                    // release below then covers it, and every reader of s_order (the emit gather of waves 0..4) is
                    // ordered after it through wave 0's acquire of s_prog and the s_halt hand-off
                    if (k == 0 && c0 == 0 && want_order) {
                        s_order[lt] = o0; s_order[lt + CHL_PER] = o1; s_order[lt + 2 * CHL_PER] = o2; s_order[lt + 3 * CHL_PER] = o3;
                        s_order[lt + 4 * CHL_PER] = o4; s_order[lt + 5 * CHL_PER] = o5; s_order[lt + 6 * CHL_PER] = o6;
                        s_order[lt + 7 * CHL_PER] = o7; s_order[lt + 8 * CHL_PER] = o8; s_order[lt + 9 * CHL_PER] = o9;
                        if (lt + 10 * CHL_PER < CHL_BOXES) s_order[lt + 10 * CHL_PER] = o10;
                    }
                    LDS_RELEASE();
                    if (lane == 0) lds_st(&s_prog[wave], (k + 1) * CHL_BATCH);
                }
                if (k + 1 >= nbatch || lds_ld(&s_halt)) break;
                {   // batch k+1 from B while batch k+2 flies into A
                    const int u = (k + 1) * CHL_BATCH + lt, v = u + CHL_BATCH;
                    CHL_REP(CHL_LDA)
                    CHL_REP(CHL_STB)
                    LDS_RELEASE();
                    if (lane == 0) lds_st(&s_prog[wave], (k + 2) * CHL_BATCH);
                }
                if (lds_ld(&s_halt)) break;
            }
        } else if (wave >= 1) {
            // ---- helpers: rows 0 .. c-4 of every fourth column, as the K words appear
            int have = 0;
            for (int c = max(c0, 4) + ((wave - 1 - max(c0, 4)) & 3); c < c1; c += CHL_HELPERS) {
                const int tb = c * (c + 1) / 2 - T0;
                if (!chl_wait_units(have, (tb + c - 3) * 32, s_prog, &s_halt, lane, true)) break;
                const unsigned long long *col = arena + (long long)tb * 64 + lane;
                const int rows = c - 3;
                unsigned long long acc = 0ull;
                int done = 0;
                bool halted = false;
                while (done < rows) {
                    const int r = min(lds_ld(&s_ready), rows);
                    if (r > done) {
                        LDS_ACQUIRE();
#pragma unroll 4
                        for (int rb = done; rb < r; ++rb) acc |= col[rb * 64] & s_K[rb];
                        done = r;
                    } else if (lds_ld(&s_halt)) { halted = true; break; }
                }
                if (halted) break;
                s_acc[(c & 7) * 64 + lane] = acc;
                LDS_RELEASE();
                if (lane == 0) lds_st(&s_accflag[c], 1);
            }
        } else {
            // ---- wave 0: the greedy dependency, software-pipelined: the LDS reads of block b+1 are issued
            // before the bookkeeping of block b.  "lane 0 only" stores are made branch-free by sending the
            // other lanes' copies to s_junk.
            __builtin_amdgcn_s_setprio(3);
            int have = 0;
            const int ce = min(c1, b1);
            unsigned long long tA = 0ull, tB = 0ull, tC = 0ull, dg = 0ull, av = 0ull;
            int fl = 0;
            unsigned long long *const junk = &s_junk[lane];
// `pd` = this lane's word of the diagonal tile of block bb; the three youngest row blocks sit right before it
#define CHL_FETCH(bb)                                                                                      \
    {                                                                                                      \
        if (__builtin_expect(have < need_u, 0)) chl_wait_units(have, need_u, s_prog, &s_halt, lane, false); \
        fl = lds_ld(&s_accflag[(bb)]);                   /* flag first, then the value (LDS is in order) */ \
        av = __hip_atomic_load(&s_acc[((bb) & 7) * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
        tA = pd[-192]; tB = pd[-128]; tC = pd[-64]; dg = pd[0];                                            \
    }
            const unsigned long long *pd = arena + ((long long)(c0 * (c0 + 1) / 2 - T0 + c0)) * 64 + lane;
            int need_u = (c0 * (c0 + 1) / 2 - T0 + c0 + 1) * 32;   // units up to and including the diagonal tile
            if (c0 < ce) CHL_FETCH(c0)
            if (d.trace && c0 == 0) t_first = (long long)__builtin_readcyclecounter();
            for (int b = c0; b < ce; ++b) {
                const long long t_begin = d.trace ? (long long)__builtin_readcyclecounter() : 0;
                // the three youngest row blocks (K of a missing block is 0) + the helpers' part
                unsigned long long acc = (tA & Kp3) | (tB & Kp2) | (tC & Kp1);
                if (b >= 4) {
                    while (__builtin_expect(!fl, 0)) {
                        fl = lds_ld(&s_accflag[b]);
                        av = __hip_atomic_load(&s_acc[(b & 7) * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    acc |= av;
                }
                const unsigned long long A = __ballot(acc == 0ull && (b * 64 + lane) < n);
                const unsigned long long dgc = dg;
                unsigned long long K = A;
                int iters = 0;
                for (;;) {                                        // fixed point = greedy set (see chain1_round)
                    const unsigned long long K2 = A & __ballot((dgc & K) == 0ull);
                    ++iters;
                    if (K2 == K) break;
                    K = K2;
                }
                // publish: K word, then the count of finished blocks (two LDS stores of one wave stay in order)
                __hip_atomic_store(lane == 0 ? &s_K[b] : junk, K, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(lane == 0 ? &s_ready : (int *)junk, b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                Kp3 = Kp2; Kp2 = Kp1; Kp1 = K;
                total += __popcll(K);
                const bool last_one = (d.max_keep > 0 && total >= d.max_keep);
                pd += (b + 2) * 64; need_u += (b + 2) * 32;        // diagonal tile of block b + 1
                if (b + 1 < ce && !last_one) CHL_FETCH(b + 1)
                if (d.trace && f == 0 && lane == 0) {
                    long long *tr = d.trace + 4 * b;
                    tr[0] = t_begin; tr[1] = t_begin; tr[2] = t_begin;   // one stamp per block (a second costs ~200 cycles)
                    tr[3] = ((long long)iters << 32) | (unsigned)__popcll(K);
                }
                if (last_one) { stop = true; break; }
            }
            frame_done = stop || ce >= b1;
            LDS_RELEASE();                                 // everything wave 0 acquired (tiles, s_order) before the halt word
            if (lane == 0) {
                if (frame_done) lds_st(&s_stop, 1);        // stop first, then halt (LDS keeps the order)
                lds_st(&s_halt, 1);
            }
        }
        if (wave != 0) {
            while (!lds_ld(&s_halt)) __builtin_amdgcn_s_sleep(1);
            LDS_ACQUIRE();                                 // s_order / s_K reads below are ordered after the halt word
            frame_done = __builtin_amdgcn_readfirstlane(lds_ld(&s_stop)) != 0;
        }
        // a finished frame needs no more tiles: its loaders leave at once (their last batch may still be in
        // flight) and the other five waves go on without a workgroup barrier
        if (frame_done) break;
        __syncthreads();
        c0 = c1;
        if (c0 >= cols) break;
        // next epoch: reset the progress words, request its first batch
        c1 = chl_epoch_end(c0, cols);
        if (threadIdx.x < 8) s_prog[threadIdx.x] = 0;
        if (threadIdx.x == 8) s_halt = 0;
        if (loader) {
            const int T1 = c0 * (c0 + 1) / 2;
            const uint4 *src = gsrc + (long long)T1 * 32;
            const int last = ((c1 * (c1 + 1)) / 2 - T1) * 32 - 1;
            CHL_REP(CHL_LDA0)
        }
        __syncthreads();
    }
    if (loader) return;
    // ---- waves 0..4 (CHL_POST threads): kept positions from the K words (cpu_nms.pyx:45 keep.append(i), in
    // processing order) and, for a finished frame, the ROI blobs (proposal_layer_tf.py:188-191) -- every
    // wave derives the prefix of the kept counts itself, so nothing here waits for another wave
    if (d.trace && f == 0 && threadIdx.x == 0) {
        long long *ph = d.trace + 4 * (long long)d.nbw;       // phase stamps after the per-block records
        ph[0] = t_start; ph[1] = t_first; ph[2] = (long long)__builtin_readcyclecounter();
    }
    const unsigned long long Kl = (lane < 32) ? s_K[lane] : 0ull;   // blocks not run have K = 0
    const int cnt = __popcll(Kl);
    int inc = cnt;
#pragma unroll
    for (int m = 1; m < 32; m <<= 1) { const int t = __shfl_up(inc, m); if (lane >= m) inc += t; }
    const int pref = inc - cnt;                                 // lane l < 32: kept before block l
    const int ktotal = __shfl(inc, 31);
    const bool finished = (d.max_keep > 0 && ktotal >= d.max_keep) || (b1 >= nb);
    const int nk = (d.max_keep > 0 && ktotal > d.max_keep) ? d.max_keep : ktotal;
    if (threadIdx.x < min(32, d.nbw)) d.kstate[(long long)f * d.nbw + threadIdx.x] = Kl;   // a frame's slice has nbw words
    for (int w = threadIdx.x; w < d.nbw; w += CHL_POST) d.rem[(long long)f * d.nbw + w] = 0ull;   // accumulated by the later rounds' tile phases
    if (threadIdx.x == 0) {
        cstate[0] = ktotal; cstate[1] = finished ? 1 : 0; cstate[2] = 0;
        if (finished) d.num_keep[f] = nk;
    }
    const bool emit = finished && d.emit.enabled;
    const EmitDev &e = d.emit;
    const int nrun = __builtin_amdgcn_readfirstlane(lds_ld(&s_ready));   // blocks the chain went through
    for (int p0 = 0; p0 < nrun * 64; p0 += CHL_POST) {
        const int p = p0 + threadIdx.x;                            // one block (64 boxes) per wave and pass
        const int w = p >> 6;                                      // wave-uniform
        const unsigned long long K = __shfl(Kl, w & 31);
        const int base = __shfl(pref, w & 31);
        if (w < nrun && ((K >> lane) & 1ull)) {
            const int pos = base + __popcll(K & ((1ull << lane) - 1ull));
            if (d.max_keep <= 0 || pos < d.max_keep) {
                keep[pos] = p;
                if (emit && pos < e.cap) {
                    const int c = s_order[p];
                    const long long o = (long long)f * e.N + c;
                    const float4 bx = e.bv[o];
                    const int4 im = e.img[o];
                    const float2 pa = *reinterpret_cast<const float2 *>(e.p3 + o * 6);
                    const float2 pb = *reinterpret_cast<const float2 *>(e.p3 + o * 6 + 2);
                    const float2 pc = *reinterpret_cast<const float2 *>(e.p3 + o * 6 + 4);
                    float *obv = e.blob_bv + ((long long)f * e.cap + pos) * 5;
                    float *oim = e.blob_img + ((long long)f * e.cap + pos) * 5;
                    float *o3 = e.blob_3d + ((long long)f * e.cap + pos) * 7;
                    const float bi = (float)f;
                    obv[0] = bi; obv[1] = bx.x; obv[2] = bx.y; obv[3] = bx.z; obv[4] = bx.w;
                    oim[0] = bi; oim[1] = (float)im.x; oim[2] = (float)im.y; oim[3] = (float)im.z; oim[4] = (float)im.w;
                    o3[0] = bi; o3[1] = pa.x; o3[2] = pa.y; o3[3] = pb.x; o3[4] = pb.y; o3[5] = pc.x; o3[6] = pc.y;
                }
            }
        }
    }
    if (emit) {
        if (threadIdx.x == 0) e.num_out[f] = nk;
        for (int r = nk + threadIdx.x; r < e.cap; r += CHL_POST) {   // rows past the kept boxes are zero
            float *obv = e.blob_bv + ((long long)f * e.cap + r) * 5;
            float *oim = e.blob_img + ((long long)f * e.cap + r) * 5;
            float *o3 = e.blob_3d + ((long long)f * e.cap + r) * 7;
#pragma unroll
            for (int q = 0; q < 5; ++q) { obv[q] = 0.0f; oim[q] = 0.0f; }
#pragma unroll
            for (int q = 0; q < 7; ++q) o3[q] = 0.0f;
        }
    }
    if (d.trace && f == 0) {
        chl_sync(&s_sync, CHL_POST / 64, lane);
        if (threadIdx.x == 0) d.trace[4 * (long long)d.nbw + 3] = (long long)__builtin_readcyclecounter();
    }
}

size_t mv3d_nms_ws_bytes(int n_cap, int batch)
{
    const size_t nbw = (size_t)(n_cap + 63) / 64;
    const size_t ntiles = nbw * (nbw + 1) / 2;
    return (size_t)batch * mv3d_align_up(ntiles * 64 * 8) + 2 * mv3d_align_up((size_t)batch * nbw * 8) +
           mv3d_align_up((size_t)batch * 4 * 4);
}

int mv3d_launch_nms(const NmsLaunch &L, hipStream_t stream)
{
    if (L.n_cap < 0 || L.batch <= 0) return MV3D_ERR_INVALID_ARG;
    const int nbw = (L.n_cap + 63) / 64;
    if (nbw > NMS_MAX_WORDS) return MV3D_ERR_INVALID_ARG;
    NmsDev d;
    d.boxes = L.boxes; d.box_stride = L.box_stride; d.boxes_frame_stride = L.boxes_frame_stride;
    d.idx = L.idx; d.idx_frame_stride = L.idx_frame_stride; d.n_dev = L.n_dev; d.n_cap = L.n_cap;
    d.nbw = nbw; d.ntiles = nbw * (nbw + 1) / 2;
    d.tf = L.strict_gt ? nextafterf(L.thresh_f32, INFINITY) : L.thresh_f32; d.max_keep = L.max_keep;
    d.fast_ok = (d.tf >= 0x1p-10f && d.tf <= 0x1p10f) ? 1 : 0;
    d.neg_h = d.fast_ok ? -0.5f * (d.tf - nextafterf(d.tf, 0.0f)) : 0.0f;
    char *ws = (char *)L.workspace;
    d.tiles = (unsigned long long *)ws;
    ws += (size_t)L.batch * mv3d_align_up((size_t)d.ntiles * 64 * 8);
    d.kstate = (unsigned long long *)ws;
    ws += mv3d_align_up((size_t)L.batch * nbw * 8);
    d.cstate = (int32_t *)ws;
    ws += mv3d_align_up((size_t)L.batch * 4 * 4);
    d.rem = (unsigned long long *)ws;
    d.keep = L.keep; d.keep_frame_stride = L.keep_frame_stride; d.num_keep = L.num_keep; d.status = L.status;
    d.emit = L.emit;
    d.trace = L.trace;
    // Rounds over column-block ranges; a later round returns at once for a frame that is already finished
    // (a frame with no boxes is finished by the first round).  The first round is 32 blocks: the greedy pass
    // almost always reaches a small max_keep (TEST: 300) inside it and the rest is one skipped launch.  A
    // large or absent max_keep (TRAIN: 2000 = at least 32 blocks) gets narrow later rounds, so that the
    // chain only ever pulls the tiles of its own round and little is computed past the stopping point.
    // (a later round is at most 128 blocks wide: chain1_round<128>)
    int bounds[8] = {0, 32, nbw, nbw, nbw, nbw, nbw, nbw};
    if (L.max_keep <= 0 || L.max_keep > 512) { bounds[2] = 64; bounds[3] = 128; bounds[4] = 256; bounds[5] = 384; bounds[6] = nbw; }
    else if (nbw > 128) { bounds[2] = 128; bounds[3] = 256; bounds[4] = 384; bounds[5] = nbw; }
    for (int r = 0; r < 7; ++r) {
        d.b0 = bounds[r] < nbw ? bounds[r] : nbw; d.b1 = bounds[r + 1] < nbw ? bounds[r + 1] : nbw;
        d.first_round = (r == 0);
        if (r > 0 && d.b0 >= d.b1) break;
        const int ntr = d.b1 * (d.b1 + 1) / 2 - d.b0 * (d.b0 + 1) / 2;
        if (r == 0) {
            if (ntr > 0) hipLaunchKernelGGL(nms_tiles_kernel, dim3(ntr, 1, L.batch), dim3(256), 0, stream, d);
            hipLaunchKernelGGL(nms_chain_lds_kernel, dim3(L.batch), dim3(CHL_THREADS), 0, stream, d);
            continue;
        }
        const int wgs = (ntr + 3) / 4 < 256 ? (ntr + 3) / 4 : 256;
        const int width = d.b1 - d.b0;                         // <= 128 by construction
        if (width <= 32) hipLaunchKernelGGL(nms_round_kernel<32>, dim3(wgs, 1, L.batch), dim3(256), 0, stream, d);
        else if (width <= 64) hipLaunchKernelGGL(nms_round_kernel<64>, dim3(wgs, 1, L.batch), dim3(256), 0, stream, d);
        else hipLaunchKernelGGL(nms_round_kernel<128>, dim3(wgs, 1, L.batch), dim3(256), 0, stream, d);
    }
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
