// Streaming micro-benchmark (runs on the GPU box): write-only, read-only and copy ceilings for the
// buffer sizes the RoiPool kernels move.  hipcc --offload-arch=gfx950 -O3 tools/hbm_probe.hip -o /tmp/hbm_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k_fill(float4 *p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void k_fill_nt(float4 *p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 v = {1.f, 2.f, 3.f, 4.f};
        __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(p) + i);
    }
}
__global__ void k_copy(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_read(const float4 *__restrict__ a, float *out, size_t n4) {
    float s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678f) out[0] = s;
}
template <typename F> float timeit(F f, int it) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e0); for (int i = 0; i < it; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / it;
}
int main() {
    const size_t sizes[] = {15u << 20, 60u << 20, 75u << 20, 256u << 20, 1024u << 20};
    float *out; hipMalloc(&out, 4);
    for (size_t bytes : sizes) {
        float4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes);
        hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
        size_t n4 = bytes / 16;
        for (int grid : {2048, 8192, 0}) {
            int g = grid ? grid : (int)((n4 + 255) / 256);
            float tf = timeit([&] { hipLaunchKernelGGL(k_fill, dim3(g), dim3(256), 0, 0, a, n4); }, 20);
            float tn = timeit([&] { hipLaunchKernelGGL(k_fill_nt, dim3(g), dim3(256), 0, 0, a, n4); }, 20);
            float tc = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, a, b, n4); }, 20);
            float tr = timeit([&] { hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, a, out, n4); }, 20);
            printf("%5zu MiB grid %7d | fill %7.1f us %6.0f GB/s | fill_nt %7.1f us %6.0f GB/s | copy %7.1f us %6.0f GB/s (r+w) | read %7.1f us %6.0f GB/s\n",
                   bytes >> 20, g, tf * 1e3, bytes / tf / 1e6, tn * 1e3, bytes / tn / 1e6, tc * 1e3, 2.0 * bytes / tc / 1e6, tr * 1e3, bytes / tr / 1e6);
        }
        hipFree(a); hipFree(b);
    }
    return 0;
}
