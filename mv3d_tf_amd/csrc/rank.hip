// Score sort of proposal_layer_3d (lib/rpn_msr/proposal_layer_tf.py:161-167:
// scores.argsort()[::-1][:pre_nms_topN]) as rank-by-counting across the whole chip.
//
// rank(i) = #{ j : key_j > key_i  or (key_j == key_i and j > i) }   (descending score,
// ties by descending index = stable ascending sort reversed; the reference leaves ties to
// numpy's unstable sort).  order[rank(i)] = i for rank(i) < cap.  No multi-pass sort, no
// inter-workgroup dependency: every workgroup owns 256 candidates and streams all keys of
// its frame through LDS as broadcast ds_read_b128 (4 keys per LDS op), 2 VALU ops per pair
// (v_cmp + add-with-carry).  The index tie-break is folded into the choice between `>` and
// `>=` per 256-key sub-tile (uniform per workgroup), so only the workgroup's own sub-tile
// pays for the full comparison.
#include "kernels.h"

#define RANK_TILE 1024

__global__ __launch_bounds__(256) void rank_kernel(const uint32_t *__restrict__ keys, int N, int32_t *order, int cap)
{
    __shared__ uint4 s_keys[RANK_TILE / 4];
    const int f = blockIdx.y;
    const uint32_t *k = keys + (long long)f * N;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t ki = (i < N) ? k[i] : 0u;
    const int my_sub = blockIdx.x;                 // index of the 256-key sub-tile holding this block's own keys
    unsigned cnt = 0;
    for (int t0 = 0; t0 < N; t0 += RANK_TILE) {
        __syncthreads();
        {
            // each thread stages 4 consecutive keys (zero-padded past N)
            const int j = t0 + threadIdx.x * 4;
            uint4 v;
            v.x = (j + 0 < N) ? k[j + 0] : 0u;
            v.y = (j + 1 < N) ? k[j + 1] : 0u;
            v.z = (j + 2 < N) ? k[j + 2] : 0u;
            v.w = (j + 3 < N) ? k[j + 3] : 0u;
            s_keys[threadIdx.x] = v;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < RANK_TILE / 256; ++s) {
            const int sub = t0 / 256 + s;
            if (sub * 256 >= N) break;
            const uint4 *q = s_keys + s * 64;
            if (sub < my_sub) {                     // all j < i : strictly larger key precedes
#pragma unroll 16
                for (int u = 0; u < 64; ++u) {
                    const uint4 v = q[u];
                    cnt += (v.x > ki); cnt += (v.y > ki); cnt += (v.z > ki); cnt += (v.w > ki);
                }
            } else if (sub > my_sub) {              // all j > i : equal key precedes too
#pragma unroll 16
                for (int u = 0; u < 64; ++u) {
                    const uint4 v = q[u];
                    cnt += (v.x >= ki); cnt += (v.y >= ki); cnt += (v.z >= ki); cnt += (v.w >= ki);
                }
            } else {                                // own sub-tile: full rule
                const int jb = sub * 256;
#pragma unroll 4
                for (int u = 0; u < 64; ++u) {
                    const uint4 v = q[u];
                    const int j = jb + u * 4;
                    cnt += (v.x > ki) || (v.x == ki && j + 0 > i);
                    cnt += (v.y > ki) || (v.y == ki && j + 1 > i);
                    cnt += (v.z > ki) || (v.z == ki && j + 2 > i);
                    cnt += (v.w > ki) || (v.w == ki && j + 3 > i);
                }
            }
        }
    }
    if (ki != 0u && (int)cnt < cap) order[(long long)f * cap + cnt] = i;
}

int mv3d_launch_rank(const uint32_t *keys, int N, int batch, int32_t *order, int cap, hipStream_t stream)
{
    if (N <= 0 || batch <= 0 || cap <= 0) return MV3D_ERR_INVALID_ARG;
    hipLaunchKernelGGL(rank_kernel, dim3((N + 255) / 256, batch), dim3(256), 0, stream, keys, N, order, cap);
    return mv3d_launch_status();
}
