// Score sort of proposal_layer_3d (lib/rpn_msr/proposal_layer_tf.py:161-167:
// scores.argsort()[::-1][:pre_nms_topN]).  order[rank(i)] = i for rank(i) < cap with
//
//   rank(i) = #{ j : key_j > key_i  or (key_j == key_i and j > i) }
//
// (descending score, ties by descending index = stable ascending sort reversed; the reference
// leaves ties to numpy's unstable sort).  Two implementations, both exact and deterministic:
//
//  A. merge-rank (key_stride <= 24576 keys per frame, i.e. every KITTI-sized grid; 2 kernels)
//     rank_local_kernel   sorts runs of 1024 keys by counting inside the run (keys wave-uniform
//                         through scalar loads, 2 VALU per pair) and writes each run sorted.
//     rank_merge_kernel   each workgroup stages ALL sorted runs of its frame in LDS (<= 96 KB of
//                         the CU's 160 KB) and every thread binary-searches its key in each other
//                         run: 11 LDS reads per run instead of 1024 compares.  Because runs own
//                         disjoint index ranges, the tie-break reduces to searching with `>` in
//                         runs of smaller index and `>=` in runs of larger index.  Scatters the
//                         order and totals n_valid (no atomics, no memset).
//  B. counting (any size; fallback): rank_partial_kernel + rank_scatter_kernel, O(N^2) pairs at two
//     VALU instructions per pair.
#include "kernels.h"

typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));

// cnt += (kj > ki) resp. (kj >= ki) for 16 wave-uniform keys kj held in SGPRs: exactly two VALU
// instructions per pair (compare into VCC, add-with-carry); hipcc's own lowering of the C form
// mixes in v_cndmask/v_add sequences.  No manual wait states are needed between a VALU write of
// VCC and a VALU carry-in read (CDNA3/4 ISA, "manually inserted wait states").
// 4 compares into 4 different SGPR pairs, then 4 add-with-carry into two accumulators: the
// VALU -> SGPR -> VALU carry path is ~8 cycles deep, so a single VCC chain (cmp, addc, cmp, addc ...)
// runs at a third of the issue rate when a SIMD holds only one or two waves.
#define RANK_C(OP, M, I) "v_cmp_" OP "_u32 %[" #M "], %" #I ", %[ki]\n\t"
#define RANK_A(ACC, M) "v_addc_co_u32 %[" #ACC "], %[" #M "], 0, %[" #ACC "], %[" #M "]\n\t"
#define RANK_G(OP, I0, I1, I2, I3)                                                         \
    RANK_C(OP, m0, I0) RANK_C(OP, m1, I1) RANK_C(OP, m2, I2) RANK_C(OP, m3, I3)            \
    RANK_A(a, m0) RANK_A(b, m1) RANK_A(a, m2) RANK_A(b, m3)
#define RANK_16(OP)                                                                                        \
    asm(RANK_G(OP, 7, 8, 9, 10) RANK_G(OP, 11, 12, 13, 14) RANK_G(OP, 15, 16, 17, 18) RANK_G(OP, 19, 20, 21, 22) \
        : [a] "+v"(cnt), [b] "+v"(cnt2), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3)     \
        : [ki] "v"(ki), "s"(v.s0), "s"(v.s1), "s"(v.s2), "s"(v.s3), "s"(v.s4), "s"(v.s5), "s"(v.s6), "s"(v.s7), \
          "s"(v.s8), "s"(v.s9), "s"(v.sa), "s"(v.sb), "s"(v.sc), "s"(v.sd), "s"(v.se), "s"(v.sf))
template <bool GE>
__device__ __forceinline__ void count16(unsigned &cnt, unsigned &cnt2, const uint32_t ki, const u32x16 v)
{
    unsigned long long m0, m1, m2, m3;
    if (GE) RANK_16("ge");
    else RANK_16("gt");
}

#define RANK_SEG 1024

// keys: (batch, key_stride) with key_stride a multiple of RANK_SEG and keys[N..key_stride) == 0,
// so every segment is full.  The keys of a segment are wave-uniform: they are read with scalar
// loads (s_load_dwordx16 through the scalar cache) and never touch LDS or the vector path.
__global__ __launch_bounds__(256) void rank_partial_kernel(const uint32_t *__restrict__ keys, int N, int key_stride,
                                                           int S, uint32_t *__restrict__ partial)
{
    const int f = blockIdx.z, s = blockIdx.y;
    const uint32_t *__restrict__ k = keys + (long long)f * key_stride;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t ki = k[i];                       // i < key_stride always (grid covers the padding)
    const int my_sub = blockIdx.x;                  // 256-key sub-tile holding this block's own keys
    const int t0 = s * RANK_SEG;
    unsigned cnt = 0, cnt2 = 0;
    for (int q4 = 0; q4 < RANK_SEG / 256; ++q4) {
        const int sub = t0 / 256 + q4;
        const u32x16 *__restrict__ q = reinterpret_cast<const u32x16 *>(k + sub * 256);
        if (sub < my_sub) {                         // all j < i : only a strictly larger key precedes
#pragma unroll 4
            for (int u = 0; u < 16; ++u) count16<false>(cnt, cnt2, ki, q[u]);
        } else if (sub > my_sub) {                  // all j > i : an equal key precedes too
#pragma unroll 4
            for (int u = 0; u < 16; ++u) count16<true>(cnt, cnt2, ki, q[u]);
        } else {                                    // own sub-tile: full rule
            const int jb = sub * 256;
            for (int u = 0; u < 256; ++u) {
                const uint32_t kj = k[jb + u];
                cnt += (kj > ki) || (kj == ki && jb + u > i);
            }
        }
    }
    if (i < N) partial[((long long)f * S + s) * N + i] = cnt + cnt2;
}

// ---------------------------------------------------------------------------- A. merge-rank
#define RANK_LDS_KEYS 24576
#define RANK_RUN 1024            // sorted run length (= RANK_SEG: one scalar-load segment)

// grid (key_stride/128, batch).  Workgroup bi owns 128 keys of run R = bi/8 and ranks them inside
// the run by counting over its 1024 keys (wave-uniform, scalar loads; `>` / `>=` per 256-key
// sub-tile as in rank_partial_kernel).  Candidates are written to their run position:
// sorted[f][R*1024 + r] = key, sidx = index inside the run.  Non-candidates (key 0) are not
// written at all; cnt128[f][run] = candidates of the run (counted by the run's first workgroup) lets the
// consumer treat the tail of every run as zeros.  (128 keys per workgroup: 184 workgroups for a KITTI frame, so the counting --
// the kernel is VALU-bound -- spreads over most of the 256 CUs.)
#define RANK_LK 128                                 // keys per workgroup
#define RANK_SUBW 128                               // keys per sub-tile
#define RANK_NSUB (RANK_RUN / RANK_SUBW)            // sub-tiles = thread groups of the workgroup (8)
__global__ __launch_bounds__(RANK_NSUB * RANK_LK) void rank_local_kernel(const uint32_t *__restrict__ keys, int key_stride,
                                                                         uint32_t *__restrict__ sorted,
                                                                         uint16_t *__restrict__ sidx, int32_t *__restrict__ cnt128)
{
    // 1024 threads = 128 keys x 8 sub-tiles of the run: thread (t, sub) counts its key against the 128 keys of
    // sub-tile `sub` (the kernel is VALU-bound: 16 waves of 260 instructions finish sooner than 8 of 520).
    __shared__ unsigned s_part[RANK_NSUB][RANK_LK];
    __shared__ int s_wc[RANK_NSUB * RANK_LK / 64];
    const int f = blockIdx.y, bi = blockIdx.x;
    const int t = threadIdx.x & (RANK_LK - 1), sub = __builtin_amdgcn_readfirstlane(threadIdx.x / RANK_LK);
    const int run = bi >> 3, kb = bi & 7;
    const int i = kb * RANK_LK + t;                 // key index inside the run
    const int my_sub = i / RANK_SUBW;               // its sub-tile (workgroup-uniform)
    const uint32_t *__restrict__ k = keys + (long long)f * key_stride + run * RANK_RUN;
    const uint32_t ki = k[i];
    const uint32_t kim1 = ki - 1u;                  // kj >= ki  <=>  kj > ki - 1   (ki >= 1 for candidates)
    unsigned cnt = 0, cnt2 = 0;
    const u32x16 *__restrict__ q = reinterpret_cast<const u32x16 *>(k + sub * RANK_SUBW);
    if (sub < my_sub) {                             // all j < i : only a strictly larger key precedes
#pragma unroll 4
        for (int u = 0; u < RANK_SUBW / 16; ++u) count16<false>(cnt, cnt2, ki, q[u]);
    } else if (sub > my_sub) {                      // all j > i : an equal key precedes too
#pragma unroll 4
        for (int u = 0; u < RANK_SUBW / 16; ++u) count16<true>(cnt, cnt2, ki, q[u]);
    } else {
        // own sub-tile: per 64-key part the rule is again uniform for a whole wave, except for the wave's own
        // part, where the later index wins a tie lane by lane
        const int il = i & (RANK_SUBW - 1);         // index inside the sub-tile
        const int wq = __builtin_amdgcn_readfirstlane(il >> 6);
        for (int qq = 0; qq < RANK_SUBW / 64; ++qq) {
            if (qq < wq) {
#pragma unroll
                for (int u = 0; u < 4; ++u) count16<false>(cnt, cnt2, ki, q[qq * 4 + u]);
            } else if (qq > wq) {
#pragma unroll
                for (int u = 0; u < 4; ++u) count16<true>(cnt, cnt2, ki, q[qq * 4 + u]);
            } else {
#pragma unroll 16
                for (int j = qq * 64; j < qq * 64 + 64; ++j) {
                    const uint32_t kj = k[sub * RANK_SUBW + j];  // wave-uniform: scalar load
                    const uint32_t thr = (j > il) ? kim1 : ki;   // kj >= ki  <=>  kj > ki - 1
                    cnt += (kj > thr) ? 1u : 0u;
                }
            }
        }
    }
    s_part[sub][t] = cnt + cnt2;
    if (kb == 0) {                                  // candidates of the whole run: one key per thread
        const int c = __popcll(__ballot(k[threadIdx.x] != 0u));
        if ((threadIdx.x & 63) == 0) s_wc[threadIdx.x >> 6] = c;
    }
    __syncthreads();
    if (sub == 0) {
        if (kb == 0 && t == 0) {
            int c = 0;
#pragma unroll
            for (int w = 0; w < RANK_NSUB * RANK_LK / 64; ++w) c += s_wc[w];
            cnt128[(long long)f * (gridDim.x >> 3) + run] = c;
        }
        if (ki != 0u) {
            unsigned r = 0;
#pragma unroll
            for (int s2 = 0; s2 < RANK_NSUB; ++s2) r += s_part[s2][t];
            const long long o = (long long)f * key_stride + run * RANK_RUN + r;
            sorted[o] = ki;
            sidx[o] = (uint16_t)i;
        }
    }
}

template <bool GE>
__device__ __forceinline__ int count_before(const uint32_t *sb, const uint32_t ki)
{
    // sb: RANK_RUN keys in descending order; number of keys kj with kj > ki (GE: kj >= ki)
    int pos = 0;
#pragma unroll
    for (int step = RANK_RUN / 2; step >= 1; step >>= 1) {
        const uint32_t v = sb[pos + step - 1];
        if (GE ? (v >= ki) : (v > ki)) pos += step;
    }
    const uint32_t v = sb[pos];                     // pos <= RANK_RUN - 1
    if (GE ? (v >= ki) : (v > ki)) pos += 1;
    return pos;
}

// grid (key_stride/128, batch), 256 threads: the thread pair (t & 127, half t >> 7) of workgroup bi owns
// position (bi&7)*128 + (t&127) of run bi/8 -- half h searches the runs of parity h -- so that the LDS
// traffic of the searches spreads over twice as many CUs.  All runs of the frame are staged in LDS (<= 96 KB).
__global__ __launch_bounds__(256) void rank_merge_kernel(const uint32_t *__restrict__ sorted,
                                                         const uint16_t *__restrict__ sidx,
                                                         const int32_t *__restrict__ cnt128, int N, int key_stride,
                                                         int32_t *order, int cap, const int32_t *part_counts, int n_parts,
                                                         int32_t *n_valid, const float4 *__restrict__ gsrc, float4 *gdst)
{
    __shared__ uint32_t s_keys[RANK_LDS_KEYS];
    __shared__ int s_half[128];
    const int f = blockIdx.y, bi = blockIdx.x, t = threadIdx.x;
    const int nrun = key_stride / RANK_RUN;
    const int R = bi >> 3, pos = (bi & 7) * 128 + (t & 127), h = __builtin_amdgcn_readfirstlane(t >> 7);
    // the scatter index of the own key is requested first: its latency hides under everything else
    const int own = (int)sidx[(long long)f * key_stride + R * RANK_RUN + pos];
    // ... and, when the caller wants its records in rank order too, the record of the own key (a slot past the
    // run's candidates holds a stale index: the address is clamped and the value unused)
    float4 rec = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gsrc && h == 0) rec = gsrc[(long long)f * N + min(R * RANK_RUN + own, N - 1)];
    const uint4 *__restrict__ src4 = reinterpret_cast<const uint4 *>(sorted + (long long)f * key_stride);
    const int32_t *__restrict__ crun = cnt128 + (long long)f * nrun;
    // staging: thread t owns keys 4t..4t+3 of every run; all loads are issued before the first use
    constexpr int MAXRUN = RANK_LDS_KEYS / RANK_RUN;
    uint4 v[MAXRUN];
#pragma unroll
    for (int r = 0; r < MAXRUN; ++r) v[r] = src4[t + 256 * min(r, nrun - 1)];
#pragma unroll
    for (int r = 0; r < MAXRUN; ++r) {
        if (r < nrun) {
            const int nc = crun[r] - 4 * t;          // candidates of the run (wave-uniform load)
            uint4 w = v[r];
            if (nc <= 0) w.x = 0u;                   // slots past the run's candidates were never written
            if (nc <= 1) w.y = 0u;
            if (nc <= 2) w.z = 0u;
            if (nc <= 3) w.w = 0u;
            reinterpret_cast<uint4 *>(s_keys)[t + 256 * r] = w;
        }
    }
    __syncthreads();
    const uint32_t ki = s_keys[R * RANK_RUN + pos];
    // count_r = #{keys of run r that precede ki}: runs of smaller index `>`; runs of larger index `>=`,
    // i.e. `> ki - 1` (ki >= 1); the own run and the runs that do not exist get a threshold nothing
    // exceeds.  The binary searches of all runs advance in lock-step, so every step is one batch of
    // independent LDS reads instead of a chain of dependent ones.
    constexpr int HR = MAXRUN / 2;
    int part = 0;
    if (ki != 0u) {
        uint32_t thr[HR];
        int cnt[HR];
#pragma unroll
        for (int i = 0; i < HR; ++i) {
            const int r = 2 * i + h;
            thr[i] = (r == R || r >= nrun) ? 0xffffffffu : (r > R ? ki - 1u : ki);
            cnt[i] = 0;
        }
        const uint32_t *base = s_keys + h * RANK_RUN;
#pragma unroll
        for (int step = RANK_RUN / 2; step >= 1; step >>= 1) {
            uint32_t q[HR];
#pragma unroll
            for (int i = 0; i < HR; ++i) q[i] = base[2 * i * RANK_RUN + cnt[i] + step - 1];
#pragma unroll
            for (int i = 0; i < HR; ++i) cnt[i] += (q[i] > thr[i]) ? step : 0;
        }
#pragma unroll
        for (int i = 0; i < HR; ++i) part += cnt[i] + ((base[2 * i * RANK_RUN + cnt[i]] > thr[i]) ? 1 : 0);
    }
    if (h == 1) s_half[t & 127] = part;
    __syncthreads();
    if (h == 0 && ki != 0u) {
        const int rank = pos + part + s_half[t];    // pos = position inside the own sorted run
        if (rank < cap) {
            order[(long long)f * cap + rank] = R * RANK_RUN + own;
            if (gdst) gdst[(long long)f * cap + rank] = rec;
        }
    }
    if (n_valid && bi == 0 && t < 64) {
        int nv = 0;
        for (int p = t; p < n_parts; p += 64) nv += part_counts[(long long)f * n_parts + p];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nv += __shfl_down(nv, o);
        if (t == 0) n_valid[f] = nv;
    }
}

// ---------------------------------------------------------------------------- B. counting (scatter)
__global__ __launch_bounds__(256) void rank_scatter_kernel(const uint32_t *__restrict__ keys, int N, int key_stride, int S,
                                                           const uint32_t *__restrict__ partial, int32_t *order, int cap,
                                                           const int32_t *part_counts, int n_parts, int32_t *n_valid,
                                                           const float4 *__restrict__ gsrc, float4 *gdst)
{
    const int f = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < N) {
        const uint32_t ki = keys[(long long)f * key_stride + i];
        if (ki != 0u) {
            unsigned cnt = 0;
            for (int s = 0; s < S; ++s) cnt += partial[((long long)f * S + s) * N + i];
            if ((int)cnt < cap) {
                order[(long long)f * cap + cnt] = i;
                if (gdst) gdst[(long long)f * cap + cnt] = gsrc[(long long)f * N + i];
            }
        }
    }
    if (n_valid && blockIdx.x == 0 && threadIdx.x < 64) {
        int v = 0;
        for (int p = threadIdx.x; p < n_parts; p += 64) v += part_counts[(long long)f * n_parts + p];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if (threadIdx.x == 0) n_valid[f] = v;
    }
}

int mv3d_rank_key_stride(int N) { return (N + RANK_SEG - 1) / RANK_SEG * RANK_SEG; }

size_t mv3d_rank_ws_bytes(int N, int batch)
{
    const size_t ks = (size_t)mv3d_rank_key_stride(N);
    if (ks <= RANK_LDS_KEYS)
        return mv3d_align_up((size_t)batch * ks * 4) + mv3d_align_up((size_t)batch * ks * 2) + mv3d_align_up((size_t)batch * (ks / 128) * 4);
    const size_t S = ks / RANK_SEG;
    return mv3d_align_up((size_t)batch * S * N * 4);
}

int mv3d_launch_rank(const uint32_t *keys, int N, int key_stride, int batch, int32_t *order, int cap,
                     const int32_t *part_counts, int n_parts, int32_t *n_valid, void *workspace, hipStream_t stream,
                     const float4 *gather_src, float4 *gather_dst)
{
    if (N <= 0 || batch <= 0 || cap <= 0 || !workspace || key_stride != mv3d_rank_key_stride(N)) return MV3D_ERR_INVALID_ARG;
    if (key_stride <= RANK_LDS_KEYS) {
        uint32_t *sorted = (uint32_t *)workspace;
        uint16_t *sidx = (uint16_t *)((char *)workspace + mv3d_align_up((size_t)batch * key_stride * 4));
        int32_t *cnt128 = (int32_t *)((char *)sidx + mv3d_align_up((size_t)batch * key_stride * 2));
        hipLaunchKernelGGL(rank_local_kernel, dim3(key_stride / 128, batch), dim3(RANK_NSUB * RANK_LK), 0, stream, keys, key_stride, sorted, sidx,
                           cnt128);
        hipLaunchKernelGGL(rank_merge_kernel, dim3(key_stride / 128, batch), dim3(256), 0, stream, sorted, sidx, cnt128, N,
                           key_stride, order, cap, part_counts, n_parts, n_valid, gather_src, gather_dst);
        return mv3d_launch_status();
    }
    const int S = key_stride / RANK_SEG, IB = (N + 255) / 256;
    uint32_t *partial = (uint32_t *)workspace;
    hipLaunchKernelGGL(rank_partial_kernel, dim3(IB, S, batch), dim3(256), 0, stream, keys, N, key_stride, S, partial);
    hipLaunchKernelGGL(rank_scatter_kernel, dim3(IB, batch), dim3(256), 0, stream, keys, N, key_stride, S, partial, order,
                       cap, part_counts, n_parts, n_valid, gather_src, gather_dst);
    return mv3d_launch_status();
}
