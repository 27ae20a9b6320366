// Shared host/device helpers for libmv3d_hip.so (gfx950 only; wave = 64 lanes).
// Build flags that matter for parity: -ffp-contract=off (every *, + is one IEEE
// rounding unless __fmaf_rn / fma is written out), no fast-math, IEEE f32 divide
// (hipcc default: -fhip-fp32-correctly-rounded-divide-sqrt), denormals preserved.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "../../include/mv3d_hip.h"

#define MV3D_WAVE 64
#define MV3D_ALIGN 256
#define MV3D_FLAG_ZERO_DIVISION 1

#define MV3D_HIP_TRY(expr)                           \
    do {                                             \
        if ((expr) != hipSuccess) return MV3D_ERR_HIP; \
    } while (0)

static inline int mv3d_launch_status() { return hipGetLastError() == hipSuccess ? MV3D_OK : MV3D_ERR_HIP; }
static inline size_t mv3d_align_up(size_t v) { return (v + MV3D_ALIGN - 1) / MV3D_ALIGN * MV3D_ALIGN; }

// Score -> order-preserving u32 key, larger key = processed earlier.  NaN first (numpy's
// ascending argsort puts NaN last; the reference reverses it).  Valid keys are >= 1, so
// 0 can mark "not a candidate".
__host__ __device__ static inline uint32_t mv3d_score_key(float s)
{
    union { float f; uint32_t u; } c;
    c.f = s;
    if (s != s) return 0xFFFFFFFFu;
    return (c.u & 0x80000000u) ? ~c.u : (c.u | 0x80000000u);
}

// Cython inline float32 max / min of lib/nms/cpu_nms.pyx:11-15
__device__ static inline float cy_max(float a, float b) { return a >= b ? a : b; }
__device__ static inline float cy_min(float a, float b) { return a <= b ? a : b; }
// numpy minimum / maximum: NaN in the first operand propagates
__device__ static inline float np_min32(float a, float b) { return (a < b || a != a) ? a : b; }
__device__ static inline float np_max32(float a, float b) { return (a >= b || a != a) ? a : b; }

// smallest f32 >= t : (double)ovr >= t  <=>  ovr >= ceil_f32(t) for every non-NaN f32 ovr
// (the Python-float compare of lib/nms/cpu_nms.pyx:65 without leaving f32 on the device)
static inline float mv3d_ceil_f32(double t)
{
    float f = (float)t;
    if ((double)f < t) f = nextafterf(f, INFINITY);
    return f;
}

// Ordered (ascending index) multi-list compaction of N items by ONE workgroup of 1024 threads.
// Wave w owns the contiguous index range [w*chunk, (w+1)*chunk); lanes read consecutive items
// (coalesced), four 64-item groups are loaded before any is consumed, list positions come from
// wave ballots + a 16-entry scan of the wave totals in LDS.  pred(i, flags[NP]) classifies item i,
// emit(k, pos, i) stores it at position pos of list k; totals[k] receives the list lengths.
template <int NP, typename Pred, typename Emit>
__device__ __forceinline__ void mv3d_block_compact(const int N, Pred pred, Emit emit, int totals[NP])
{
    __shared__ int s_cnt[16][NP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunk = (((N + 15) / 16) + 63) / 64 * 64;
    const int beg = wave * chunk, end = min(N, beg + chunk);
    int cnt[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) cnt[k] = 0;
    for (int b = beg; b < end; b += 256) {
        bool f[4][NP];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = b + 64 * u + lane;
#pragma unroll
            for (int k = 0; k < NP; ++k) f[u][k] = false;
            if (i < end) pred(i, f[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < NP; ++k) cnt[k] += __popcll(__ballot(f[u][k]));
    }
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < NP; ++k) s_cnt[wave][k] = cnt[k];
    __syncthreads();
    int off[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        int o = 0, t = 0;
        for (int w = 0; w < 16; ++w) { const int c = s_cnt[w][k]; if (w < wave) o += c; t += c; }
        off[k] = o; totals[k] = t;
    }
    for (int b = beg; b < end; b += 256) {
        bool f[4][NP];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = b + 64 * u + lane;
#pragma unroll
            for (int k = 0; k < NP; ++k) f[u][k] = false;
            if (i < end) pred(i, f[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = b + 64 * u + lane;
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const unsigned long long bal = __ballot(f[u][k]);
                if (f[u][k]) emit(k, off[k] + __popcll(bal & ((1ull << lane) - 1ull)), i);
                off[k] += __popcll(bal);
            }
        }
    }
}

// Ordered (ascending index) multi-list compaction of N items by a GRID of 256-thread workgroups, 1024 items
// each (wave w of workgroup b owns items [1024 b + 256 w, +256), four ballots of 64).  Positions across
// workgroups come from a decoupled look-back: every workgroup publishes its per-list counts (+1, so that 0
// means "not yet") in agg[b*NP + k] with agent-scope stores, then sums the entries of its predecessors
// (workgroups are dispatched in order, so a predecessor is always running or done).  agg has
// gridDim.x*NP + 1 words and must be zero on entry; the workgroup that finishes last clears it again.
// pred(i, flags[NP]) classifies item i, emit(k, pos, i) stores it at position pos of list k.  totals[] is
// valid in the LAST workgroup (blockIdx.x == gridDim.x - 1) only.
#define MV3D_GC_ITEMS 1024
template <int NP, typename Pred, typename Emit>
__device__ __forceinline__ void mv3d_grid_compact(const int N, Pred pred, Emit emit, int32_t *agg, int totals[NP])
{
    __shared__ int s_cnt[4][NP];
    __shared__ int s_off[NP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x, G = gridDim.x;
    const int beg = b * MV3D_GC_ITEMS + wave * 256;
    bool f[4][NP];
    int cnt[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) cnt[k] = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = beg + 64 * u + lane;
#pragma unroll
        for (int k = 0; k < NP; ++k) f[u][k] = false;
        if (i < N) pred(i, f[u]);
#pragma unroll
        for (int k = 0; k < NP; ++k) cnt[k] += __popcll(__ballot(f[u][k]));
    }
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < NP; ++k) s_cnt[wave][k] = cnt[k];
    if (threadIdx.x < NP) s_off[threadIdx.x] = 0;
    __syncthreads();
    if (threadIdx.x < NP) {
        const int k = threadIdx.x;
        __hip_atomic_store(&agg[b * NP + k], s_cnt[0][k] + s_cnt[1][k] + s_cnt[2][k] + s_cnt[3][k] + 1, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int t = threadIdx.x; t < b * NP; t += blockDim.x) {      // look back over all predecessors at once
        int v;
        while ((v = __hip_atomic_load(&agg[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) __builtin_amdgcn_s_sleep(2);
        atomicAdd(&s_off[t % NP], v - 1);
    }
    __syncthreads();
    int off[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        int o = s_off[k], t = o;
        for (int w = 0; w < 4; ++w) { const int c = s_cnt[w][k]; if (w < wave) o += c; t += c; }
        off[k] = o; totals[k] = t;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = beg + 64 * u + lane;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const unsigned long long bal = __ballot(f[u][k]);
            if (f[u][k]) emit(k, off[k] + __popcll(bal & ((1ull << lane) - 1ull)), i);
            off[k] += __popcll(bal);
        }
    }
    __syncthreads();                                              // every look-back of this workgroup is over
    if (threadIdx.x == 0 &&
        __hip_atomic_fetch_add(&agg[G * NP], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == G - 1) {
        for (int t = 0; t <= G * NP; ++t) __hip_atomic_store(&agg[t], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

