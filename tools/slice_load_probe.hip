// What bounds a gather of 256-B record slices?  NREC records of 2 KB (512 f32) in two arrays; XCD x (= blockIdx % 8) reads slice x
// (bytes [256 x, 256 x + 256)) of every record of a pseudo-random index list, from both arrays, 16 wave-loads per array in
// flight, sums them and stores one value per wave.  A: one record per wave-load (64 lanes x 4 B).  B: four records per wave-load
// (16 lanes x 16 B each: lane group g takes every 4th list entry) -- the same bytes with a quarter of the load instructions.
// hipcc --offload-arch=gfx950 -O3 tools/slice_load_probe.hip -o tools/bin/slice_load_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define W 16
__global__ __launch_bounds__(256) void k_a(const float *am, const float *td, const int *list, int per_wave, float *out)
{
    const int lane = threadIdx.x & 63, xcd = blockIdx.x & 7;
    const int wave = (blockIdx.x >> 3) * 4 + (threadIdx.x >> 6);
    const int *l = list + (long long)wave * per_wave;
    float a = 0.f;
    for (int i = 0; i < per_wave; i += W) {
        float x[W], y[W];
#pragma unroll
        for (int u = 0; u < W; ++u) {
            const long long o = (long long)__builtin_amdgcn_readfirstlane(l[i + u]) * 512 + xcd * 64 + lane;
            x[u] = am[o]; y[u] = td[o];
        }
#pragma unroll
        for (int u = 0; u < W; ++u) a += (x[u] == 3.0f) ? y[u] : 0.f;
    }
    out[(long long)blockIdx.x * 256 + threadIdx.x] = a;
}
__global__ __launch_bounds__(256) void k_b(const float *am, const float *td, const int *list, int per_wave, float *out)
{
    const int lane = threadIdx.x & 63, xcd = blockIdx.x & 7, grp = lane >> 4, sub = lane & 15;
    const int wave = (blockIdx.x >> 3) * 4 + (threadIdx.x >> 6);
    const int *l = list + (long long)wave * per_wave;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < per_wave; i += 4 * W) {
        float4 x[W], y[W];
#pragma unroll
        for (int u = 0; u < W; ++u) {
            const long long o = (long long)l[i + 4 * u + grp] * 512 + xcd * 64 + sub * 4;
            x[u] = *reinterpret_cast<const float4 *>(am + o); y[u] = *reinterpret_cast<const float4 *>(td + o);
        }
#pragma unroll
        for (int u = 0; u < W; ++u) {
            a.x += (x[u].x == 3.0f) ? y[u].x : 0.f; a.y += (x[u].y == 3.0f) ? y[u].y : 0.f;
            a.z += (x[u].z == 3.0f) ? y[u].z : 0.f; a.w += (x[u].w == 3.0f) ? y[u].w : 0.f;
        }
    }
    out[(long long)blockIdx.x * 256 + threadIdx.x] = a.x + a.y + a.z + a.w;
}
// C: like A, but the slice a workgroup reads ROTATES with the record ((xcd + record) & 7): every XCD then touches all eight address
// residues mod 2 KB instead of one -- is the ceiling of A / B the L2's channels (a fixed slice = a fixed set of address bits 8..10)?
__global__ __launch_bounds__(256) void k_c(const float *am, const float *td, const int *list, int per_wave, float *out)
{
    const int lane = threadIdx.x & 63, xcd = blockIdx.x & 7;
    const int wave = (blockIdx.x >> 3) * 4 + (threadIdx.x >> 6);
    const int *l = list + (long long)wave * per_wave;
    float a = 0.f;
    for (int i = 0; i < per_wave; i += W) {
        float x[W], y[W];
#pragma unroll
        for (int u = 0; u < W; ++u) {
            const int r = __builtin_amdgcn_readfirstlane(l[i + u]);
            const long long o = (long long)r * 512 + ((xcd + r) & 7) * 64 + lane;
            x[u] = am[o]; y[u] = td[o];
        }
#pragma unroll
        for (int u = 0; u < W; ++u) a += (x[u] == 3.0f) ? y[u] : 0.f;
    }
    out[(long long)blockIdx.x * 256 + threadIdx.x] = a;
}
// D: like A, but through the LDS-DMA path (buffer_load_dword ... lds into a per-wave LDS area, then ds_read): does the CU pull
// more from its L2 that way than through VGPR loads (the convolution's DMA reaches ~33 B/clk/CU on L2-resident operands)?
typedef __attribute__((address_space(3))) void lds_void_t;
__global__ __launch_bounds__(256) void k_d(const float *am, const float *td, const int *list, int per_wave, float *out)
{
    __shared__ float buf[4][2][W][64];
    const int lane = threadIdx.x & 63, xcd = blockIdx.x & 7, wv = threadIdx.x >> 6;
    const int wave = (blockIdx.x >> 3) * 4 + wv;
    const int *l = list + (long long)wave * per_wave;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)am, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void *)td, 0, 0x7fffffff, 0x00020000);
    float a = 0.f;
    for (int i = 0; i < per_wave; i += W) {
#pragma unroll
        for (int u = 0; u < W; ++u) {
            const int so = __builtin_amdgcn_readfirstlane(l[i + u]) * 2048 + xcd * 256;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t *)&buf[wv][0][u][0], 4, lane * 4, so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (lds_void_t *)&buf[wv][1][u][0], 4, lane * 4, so, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < W; ++u) a += (buf[wv][0][u][lane] == 3.0f) ? buf[wv][1][u][lane] : 0.f;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    out[(long long)blockIdx.x * 256 + threadIdx.x] = a;
}
int main()
{
    const int NREC = 37632;                       // 3 views x 256 ROIs x 49 bins
    const size_t bytes = (size_t)NREC * 2048;
    float *am, *td, *out; int *list;
    (void)hipMalloc(&am, bytes); (void)hipMalloc(&td, bytes); (void)hipMemset(am, 0, bytes); (void)hipMemset(td, 0, bytes);
    for (int groups : {256, 160}) {
        const int waves = groups * 4;              // per XCD
        for (int visits : {1, 2}) {                // every record once / twice per XCD (the gather visits a record ~2.2 times)
            const int per_wave = ((NREC * visits + waves - 1) / waves + 63) / 64 * 64;
            std::vector<int> h((size_t)waves * per_wave);
            unsigned s = 12345u;
            for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (int)((s >> 8) % NREC); }
            // neighbouring list entries = neighbouring records, as candidate lists have them (roi, ph, pw consecutive)
            for (size_t i = 0; i < h.size(); ++i) if (i % 8) h[i] = (h[i - 1] + 1) % NREC;
            (void)hipMalloc(&list, h.size() * 4); (void)hipMemcpy(list, h.data(), h.size() * 4, hipMemcpyHostToDevice);
            (void)hipMalloc(&out, (size_t)groups * 8 * 256 * 4);
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            for (int var = 0; var < 4; ++var) {
                float best = 1e9f;
                for (int rep = 0; rep < 6; ++rep) {
                    (void)hipEventRecord(e0);
                    if (var == 0) hipLaunchKernelGGL(k_a, dim3(groups * 8), dim3(256), 0, 0, am, td, list, per_wave, out);
                    else if (var == 1) hipLaunchKernelGGL(k_b, dim3(groups * 8), dim3(256), 0, 0, am, td, list, per_wave, out);
                    else if (var == 2) hipLaunchKernelGGL(k_c, dim3(groups * 8), dim3(256), 0, 0, am, td, list, per_wave, out);
                    else hipLaunchKernelGGL(k_d, dim3(groups * 8), dim3(256), 0, 0, am, td, list, per_wave, out);
                    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
                }
                const double gb = 2.0 * (double)waves * per_wave * 256.0 * 8.0 / 1e9;      // bytes through the vector memory path
                printf("groups %3d  %d visit(s)/record  %s: %7.1f us  %6.0f GB/s at the L1 (%.0f MB unique)\n", groups, visits,
                       var == 0 ? "A dword,   1 record / load " : (var == 1 ? "B dwordx4, 4 records / load" : (var == 2 ? "C dword, slice rotates     " : "D dword through LDS-DMA    ")), best * 1e3, gb / (best * 1e-3), 2 * bytes / 1e6);
            }
            (void)hipFree(list); (void)hipFree(out);
        }
    }
    return 0;
}
