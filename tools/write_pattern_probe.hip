// Write-pattern micro-benchmark (runs on the GPU box): how fast do the RoiPool forward's output patterns go
// when the output is far larger than the 256 MB infinity cache?  Two arrays (top f32 + argmax i32) of
// `rows` x 2 KB, written
//   stream   one workgroup per 4 KB, plain order                       (the fill ceiling)
//   sliced   the XCD channel slicing: workgroup b writes the 256-B slice b%8 of 32 consecutive rows
//   sliced2  the same, workgroup b writes slice b%8 of 64 rows as 2 x (16 rows x 512 B) ... (variants)
// hipcc --offload-arch=gfx950 -O3 tools/write_pattern_probe.hip -o /tmp/wpp && /tmp/wpp
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_stream(float4 *a, float4 *b) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    a[i] = make_float4(1, 2, 3, 4); b[i] = make_float4(5, 6, 7, 8);
}
// rows of 2 KB = 128 float4; slice = 16 float4 (256 B); per pass 16 rows, PASSES passes
template <int PASSES, int SLICE4>   // SLICE4 float4 per row and workgroup (16 = 256 B, 32 = 512 B, 64 = 1 KB)
__global__ void k_sliced(float4 *a, float4 *b, size_t rows) {
    constexpr int NSL = 128 / SLICE4;              // slices per row
    constexpr int RPP = 256 / SLICE4;              // rows per pass
    const int slice = blockIdx.x % NSL;
    const size_t row0 = (size_t)(blockIdx.x / NSL) * (PASSES * RPP);
    const int sub = threadIdx.x / SLICE4, l = threadIdx.x % SLICE4;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const size_t r = row0 + p * RPP + sub;
        if (r < rows) {
            const size_t i = r * 128 + slice * SLICE4 + l;
            a[i] = make_float4(1, 2, 3, 4); b[i] = make_float4(5, 6, 7, 8);
        }
    }
}
template <typename F> float timeit(F f, int it) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) f();
    hipEventRecord(e0); for (int i = 0; i < it; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / it;
}
int main() {
    for (size_t rows : {(size_t)14700, (size_t)14700 * 4, (size_t)14700 * 16, (size_t)14700 * 32}) {   // 300 rois x 49 bins x batch
        const size_t bytes = rows * 2048;
        float4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes);
        const double gb = 2.0 * bytes / 1e9;
        float t0 = timeit([&] { hipLaunchKernelGGL(k_stream, dim3(rows * 128 / 256), dim3(256), 0, 0, a, b); }, 10);
        float t1 = timeit([&] { hipLaunchKernelGGL((k_sliced<2, 16>), dim3((rows + 31) / 32 * 8), dim3(256), 0, 0, a, b, rows); }, 10);
        float t2 = timeit([&] { hipLaunchKernelGGL((k_sliced<4, 16>), dim3((rows + 63) / 64 * 8), dim3(256), 0, 0, a, b, rows); }, 10);
        float t3 = timeit([&] { hipLaunchKernelGGL((k_sliced<2, 32>), dim3((rows + 15) / 16 * 4), dim3(256), 0, 0, a, b, rows); }, 10);
        float t4 = timeit([&] { hipLaunchKernelGGL((k_sliced<2, 64>), dim3((rows + 7) / 8 * 2), dim3(256), 0, 0, a, b, rows); }, 10);
        float t5 = timeit([&] { hipLaunchKernelGGL((k_sliced<1, 128>), dim3((rows + 1) / 2), dim3(256), 0, 0, a, b, rows); }, 10);
        printf("%6.0f MB x2 | stream %7.1f us %5.0f GB/s | 256B x32rows %7.1f us %5.0f | 256B x64rows %7.1f us %5.0f | 512B x16 %7.1f us %5.0f | 1KB x8 %7.1f us %5.0f | 2KB x2 %7.1f us %5.0f\n",
               bytes / 1e6, t0 * 1e3, gb / t0 * 1e3, t1 * 1e3, gb / t1 * 1e3, t2 * 1e3, gb / t2 * 1e3, t3 * 1e3, gb / t3 * 1e3, t4 * 1e3, gb / t4 * 1e3,
               t5 * 1e3, gb / t5 * 1e3);
        hipFree(a); hipFree(b);
    }
    return 0;
}
