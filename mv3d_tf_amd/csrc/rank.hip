// Score sort of proposal_layer_3d (lib/rpn_msr/proposal_layer_tf.py:161-167:
// scores.argsort()[::-1][:pre_nms_topN]) as rank-by-counting across the whole chip.
//
// rank(i) = #{ j : key_j > key_i  or (key_j == key_i and j > i) }   (descending score,
// ties by descending index = stable ascending sort reversed; the reference leaves ties to
// numpy's unstable sort).  order[rank(i)] = i for rank(i) < cap.  No multi-pass sort and no
// inter-workgroup dependency:
//
//   rank_partial_kernel  grid (N/256, N/1024, batch): workgroup (bi, s) owns 256 candidates i
//        and the 1024 keys j of segment s, which are wave-uniform and arrive in SGPRs through
//        scalar loads; 2 VALU ops per pair (v_cmp into VCC + add-with-carry), nothing else.  The
//        index tie-break is folded into the choice between `>` and `>=` per 256-key sub-tile
//        (uniform per workgroup), so only the workgroup's own sub-tile pays for the full rule.
//        Writes partial[f][s][i].
//   rank_scatter_kernel  sums the partial counts of a candidate and scatters its index; block 0
//        also totals the per-workgroup candidate counts left by the producer (n_valid), so no
//        atomics and no memset are needed anywhere on the path.
#include "kernels.h"

typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));

// cnt += (kj > ki) resp. (kj >= ki) for 16 wave-uniform keys kj held in SGPRs: exactly two VALU
// instructions per pair (compare into VCC, add-with-carry); hipcc's own lowering of the C form
// mixes in v_cndmask/v_add sequences.  No manual wait states are needed between a VALU write of
// VCC and a VALU carry-in read (CDNA3/4 ISA, "manually inserted wait states").
#define RANK_P(OP, I) "v_cmp_" OP "_u32 vcc, %" #I ", %1\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
#define RANK_16(OP)                                                                                        \
    asm(RANK_P(OP, 2) RANK_P(OP, 3) RANK_P(OP, 4) RANK_P(OP, 5) RANK_P(OP, 6) RANK_P(OP, 7) RANK_P(OP, 8)   \
        RANK_P(OP, 9) RANK_P(OP, 10) RANK_P(OP, 11) RANK_P(OP, 12) RANK_P(OP, 13) RANK_P(OP, 14)             \
        RANK_P(OP, 15) RANK_P(OP, 16) RANK_P(OP, 17)                                                         \
        : "+v"(cnt)                                                                                          \
        : "v"(ki), "s"(v.s0), "s"(v.s1), "s"(v.s2), "s"(v.s3), "s"(v.s4), "s"(v.s5), "s"(v.s6), "s"(v.s7),   \
          "s"(v.s8), "s"(v.s9), "s"(v.sa), "s"(v.sb), "s"(v.sc), "s"(v.sd), "s"(v.se), "s"(v.sf)             \
        : "vcc")
template <bool GE>
__device__ __forceinline__ void count16(unsigned &cnt, const uint32_t ki, const u32x16 v)
{
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
    unsigned cnt = 0;
    for (int q4 = 0; q4 < RANK_SEG / 256; ++q4) {
        const int sub = t0 / 256 + q4;
        const u32x16 *__restrict__ q = reinterpret_cast<const u32x16 *>(k + sub * 256);
        if (sub < my_sub) {                         // all j < i : only a strictly larger key precedes
#pragma unroll 4
            for (int u = 0; u < 16; ++u) count16<false>(cnt, ki, q[u]);
        } else if (sub > my_sub) {                  // all j > i : an equal key precedes too
#pragma unroll 4
            for (int u = 0; u < 16; ++u) count16<true>(cnt, ki, q[u]);
        } else {                                    // own sub-tile: full rule
            const int jb = sub * 256;
            for (int u = 0; u < 256; ++u) {
                const uint32_t kj = k[jb + u];
                cnt += (kj > ki) || (kj == ki && jb + u > i);
            }
        }
    }
    if (i < N) partial[((long long)f * S + s) * N + i] = cnt;
}

__global__ __launch_bounds__(256) void rank_scatter_kernel(const uint32_t *__restrict__ keys, int N, int key_stride, int S,
                                                           const uint32_t *__restrict__ partial, int32_t *order, int cap,
                                                           const int32_t *part_counts, int n_parts, int32_t *n_valid)
{
    const int f = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < N) {
        const uint32_t ki = keys[(long long)f * key_stride + i];
        if (ki != 0u) {
            unsigned cnt = 0;
            for (int s = 0; s < S; ++s) cnt += partial[((long long)f * S + s) * N + i];
            if ((int)cnt < cap) order[(long long)f * cap + cnt] = i;
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

size_t mv3d_rank_ws_bytes(int N, int batch)
{
    const size_t S = (size_t)(N + RANK_SEG - 1) / RANK_SEG;
    return mv3d_align_up((size_t)batch * S * N * 4);
}

int mv3d_rank_key_stride(int N) { return (N + RANK_SEG - 1) / RANK_SEG * RANK_SEG; }

int mv3d_launch_rank(const uint32_t *keys, int N, int key_stride, int batch, int32_t *order, int cap,
                     const int32_t *part_counts, int n_parts, int32_t *n_valid, void *workspace, hipStream_t stream)
{
    if (N <= 0 || batch <= 0 || cap <= 0 || !workspace || key_stride != mv3d_rank_key_stride(N)) return MV3D_ERR_INVALID_ARG;
    const int S = key_stride / RANK_SEG, IB = (N + 255) / 256;
    uint32_t *partial = (uint32_t *)workspace;
    hipLaunchKernelGGL(rank_partial_kernel, dim3(IB, S, batch), dim3(256), 0, stream, keys, N, key_stride, S, partial);
    hipLaunchKernelGGL(rank_scatter_kernel, dim3(IB, batch), dim3(256), 0, stream, keys, N, key_stride, S, partial, order,
                       cap, part_counts, n_parts, n_valid);
    return mv3d_launch_status();
}
