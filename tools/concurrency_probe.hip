// How many kernels of different streams run at the same time on the box?  S streams, each a chain of K launches of a
// one-workgroup kernel that spins for T us (s_memrealtime); serial execution would take S*K*T, full concurrency K*T.
// Second table: the same chains while another stream keeps the chip full with a streaming kernel (2048 x 256 threads).
// hipcc --offload-arch=gfx950 -O3 tools/concurrency_probe.hip -o /tmp/cp && /tmp/cp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <vector>
__global__ void k_spin(long long ticks, int *sink)
{
    const long long t0 = __builtin_readcyclecounter();           // s_memtime: 100 MHz constant clock
    long long t = t0;
    while (t - t0 < ticks) t = __builtin_readcyclecounter();
    if (ticks < 0) sink[0] = (int)t;
}
__global__ void k_stream(float4 *p, size_t n, int rounds)
{
    for (int r = 0; r < rounds; ++r)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
            float4 v = p[i]; v.x += 1.0f; p[i] = v;
        }
}
int main()
{
    int *sink; (void)hipMalloc(&sink, 64);
    float4 *big; const size_t nbig = (size_t)65536 * 256; (void)hipMalloc(&big, nbig * 16); (void)hipMemset(big, 0, nbig * 16);
    // calibrate the cycle counter against the host clock
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, 0, 1000, sink); (void)hipDeviceSynchronize();
    auto h0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, 0, 10000000LL, sink); (void)hipDeviceSynchronize();
    const double us_total = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();
    const double ticks_per_us = 10000000.0 / us_total;
    printf("cycle counter: %.1f ticks / us\n", ticks_per_us);
    const int K = 50;
    // bg 0: idle chip; 1: persistent streaming kernel on every wave slot (2048 x 256); 2: on half of them (1024 x 256);
    // 3: on a quarter; 4: NOT persistent: 65536 short workgroups (one 4 KB piece each) per launch, launched back to back
    for (int bg = 0; bg < 5; ++bg)
        for (double T : {5.0}) {
            for (int S : {1, 2, 4}) {
                std::vector<hipStream_t> st(S); for (auto &s : st) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
                hipStream_t bgs; (void)hipStreamCreateWithFlags(&bgs, hipStreamNonBlocking);
                const long long ticks = (long long)(T * ticks_per_us);
                for (auto &s : st) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, ticks, sink);
                (void)hipDeviceSynchronize();
                auto t0 = std::chrono::steady_clock::now();
                if (bg >= 1 && bg <= 3) hipLaunchKernelGGL(k_stream, dim3(4096 >> bg), dim3(256), 0, bgs, big, nbig, 100 << bg);
                if (bg == 4) for (int r = 0; r < 60; ++r) hipLaunchKernelGGL(k_stream, dim3(65536), dim3(256), 0, bgs, big, (size_t)65536 * 256, 1);
                for (int k = 0; k < K; ++k) for (auto &s : st) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, ticks, sink);
                for (auto &s : st) (void)hipStreamSynchronize(s);
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                auto t1 = std::chrono::steady_clock::now();
                (void)hipDeviceSynchronize();
                const double us_all = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                (void)t1;
                printf("[bg ends at %8.1f us] ", us_all);
                printf("%s T=%4.0f us  streams=%2d: %8.1f us for %d launches each  -> %.2f us per launch slot, concurrency %.2f\n",
                       bg == 0 ? "idle chip" : bg == 1 ? "bg persistent x2048" : bg == 2 ? "bg persistent x1024" : bg == 3 ? "bg persistent x512" : "bg 65536 short wgs", T, S, us, K, us / K, S * K * T / us);
                for (auto &s : st) (void)hipStreamDestroy(s);
                (void)hipStreamDestroy(bgs);
            }
        }
    return 0;
}
