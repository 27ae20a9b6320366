// tf.train.AdamOptimizer's update (lib/fast_rcnn/train_mv.py:138-146: lr 1e-5, beta 0.9 / 0.999, eps 1e-8) for EVERY parameter tensor of the
// graph in ONE launch -- an HBM-bound elementwise pass: 16 B read (param, grad, both moments) + 12 B written per element, 214 M elements =
// 6.0 GB per step of the 3-view graph.  A tensor may name a 16-bit buffer that receives the updated parameter in the same pass (the dense
// head's weights: the next step's GEMMs read them in bf16, so the cast launch and its 4 B / element re-read disappear).
//
// The tensors are cut into chunks of ADAM_CHUNK elements by the host ONCE (chunk -> tensor, offset: two small device tables that only
// change when the parameter list does); workgroup = chunk, so a 100 M-element fc6 matrix and a 64-element bias go through the same grid
// without a tail problem.  Per thread four 16-byte pieces of each array in flight (64 loads per wave before the first use); the moments
// and the parameter are updated in place, f32 throughout, in the order torch's fused Adam evaluates the formula (lerp for the first
// moment), so a step taken here and a step taken by torch.optim.Adam agree to the last bits of the multiply-adds.
#include "common.h"

#define ADAM_CHUNK 4096            // elements per workgroup (256 threads x 4 pieces x 4 floats)

struct AdamArgs {
    const mv3d_adam_tensor *tensors;
    const int32_t *chunk_tensor;   // chunk -> tensor index
    const int32_t *chunk_first;    // chunk -> first element / ADAM_CHUNK inside its tensor
    float lr_over_bc1;             // lr / (1 - beta1^t)
    float inv_sqrt_bc2;            // 1 / sqrt(1 - beta2^t)
    float beta1, beta2, eps;
    float omb1, omb2;              // 1 - beta, rounded from the DOUBLE difference as torch rounds them (1.0f - 0.999f is 1.3e-5 off)
    int lowp;                      // type of the optional 16-bit copies (mv3d_adam_tensor.param_lowp): 1 = f16, 2 = bf16
};

// the updated parameter in the 16-bit type the next step's GEMMs read it in (round to nearest even: what a cast launch would write)
__device__ __forceinline__ unsigned short adam_lowp(const float x, const int kind)
{
    if (kind == 1) { const _Float16 h = (_Float16)x; return __builtin_bit_cast(unsigned short, h); }
    const __bf16 h = (__bf16)x;
    return __builtin_bit_cast(unsigned short, h);
}

__device__ __forceinline__ void adam_one(float &p, const float g, float &m, float &v, const AdamArgs &a)
{
    m = m + (g - m) * a.omb1;                                          // lerp(exp_avg, grad, 1 - beta1)
    v = a.beta2 * v + a.omb2 * g * g;
    const float denom = sqrtf(v) * a.inv_sqrt_bc2 + a.eps;
    p = p - a.lr_over_bc1 * (m / denom);
}

__global__ __launch_bounds__(256) void adam_step_kernel(AdamArgs a)
{
    const int c = blockIdx.x;
    const mv3d_adam_tensor t = a.tensors[a.chunk_tensor[c]];
    const long long first = (long long)a.chunk_first[c] * ADAM_CHUNK;
    const long long n = t.numel - first < ADAM_CHUNK ? t.numel - first : ADAM_CHUNK;
    float *const p = t.param + first;
    const float *const g = t.grad + first;
    float *const m = t.exp_avg + first, *const v = t.exp_avg_sq + first;
    if (n == ADAM_CHUNK && (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0) {
        float4 P[4], G[4], M[4], V[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = u * 256 + threadIdx.x;
            P[u] = reinterpret_cast<const float4 *>(p)[i]; G[u] = reinterpret_cast<const float4 *>(g)[i];
            M[u] = reinterpret_cast<const float4 *>(m)[i]; V[u] = reinterpret_cast<const float4 *>(v)[i];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = u * 256 + threadIdx.x;
            adam_one(P[u].x, G[u].x, M[u].x, V[u].x, a); adam_one(P[u].y, G[u].y, M[u].y, V[u].y, a);
            adam_one(P[u].z, G[u].z, M[u].z, V[u].z, a); adam_one(P[u].w, G[u].w, M[u].w, V[u].w, a);
            reinterpret_cast<float4 *>(p)[i] = P[u]; reinterpret_cast<float4 *>(m)[i] = M[u]; reinterpret_cast<float4 *>(v)[i] = V[u];
            if (t.param_lowp) {
                unsigned short *const q = (unsigned short *)t.param_lowp + first + 4 * (long long)i;
                const uint2 w = {(unsigned)adam_lowp(P[u].x, a.lowp) | ((unsigned)adam_lowp(P[u].y, a.lowp) << 16),
                                 (unsigned)adam_lowp(P[u].z, a.lowp) | ((unsigned)adam_lowp(P[u].w, a.lowp) << 16)};
                if (((uintptr_t)q & 7) == 0) *reinterpret_cast<uint2 *>(q) = w;
                else { q[0] = (unsigned short)w.x; q[1] = (unsigned short)(w.x >> 16); q[2] = (unsigned short)w.y; q[3] = (unsigned short)(w.y >> 16); }
            }
        }
        return;
    }
    for (long long i = threadIdx.x; i < n; i += 256) {                 // a tensor's last chunk, or a view at an odd offset
        float pp = p[i], mm = m[i], vv = v[i];
        adam_one(pp, g[i], mm, vv, a);
        p[i] = pp; m[i] = mm; v[i] = vv;
        if (t.param_lowp) ((unsigned short *)t.param_lowp)[first + i] = adam_lowp(pp, a.lowp);
    }
}

extern "C" int mv3d_adam_chunk_elements(void) { return ADAM_CHUNK; }

extern "C" int mv3d_adam_step(const mv3d_adam_tensor *tensors_dev, const int32_t *chunk_tensor_dev, const int32_t *chunk_first_dev,
                              int num_chunks, double lr, double beta1, double beta2, double eps, int step, int lowp_dtype, void *stream)
{
    if (!tensors_dev || !chunk_tensor_dev || !chunk_first_dev || num_chunks < 0 || step < 1) return MV3D_ERR_INVALID_ARG;
    if (lowp_dtype < 0 || lowp_dtype > 2) return MV3D_ERR_INVALID_ARG;
    if (!(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0)) return MV3D_ERR_INVALID_ARG;
    if (num_chunks == 0) return MV3D_OK;
    AdamArgs a;
    a.tensors = tensors_dev; a.chunk_tensor = chunk_tensor_dev; a.chunk_first = chunk_first_dev;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    a.lr_over_bc1 = (float)(lr / bc1);
    a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.eps = (float)eps;
    a.omb1 = (float)(1.0 - beta1); a.omb2 = (float)(1.0 - beta2);
    a.lowp = lowp_dtype ? lowp_dtype : 2;
    hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)num_chunks), dim3(256), 0, (hipStream_t)stream, a);
    return mv3d_launch_status();
}
