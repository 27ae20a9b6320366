// Dependent-launch floor on the box: K back-to-back launches in one stream of (a) an empty kernel, (b) a kernel whose
// every workgroup reads one word the previous kernel wrote and exits, (c) the same with 256 / 2048 workgroups.
// hipcc --offload-arch=gfx950 -O3 tools/launch_floor.hip -o /tmp/lf && /tmp/lf
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_empty() {}
__global__ void k_dep(const int *in, int *out) { if (in[0] == 12345) out[1] = 1; if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = in[0] + 1; }
template <typename F> float timeit(F f, int it) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) f();
    (void)hipEventRecord(e0); for (int i = 0; i < it; ++i) f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / it * 1e3f;
}
int main() {
    int *a, *b; (void)hipMalloc(&a, 64); (void)hipMalloc(&b, 64); (void)hipMemset(a, 0, 64); (void)hipMemset(b, 0, 64);
    printf("empty kernel, 1 wg x 64:            %.2f us per launch\n", timeit([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0); }, 2000));
    printf("empty kernel, 256 wg x 256:         %.2f us per launch\n", timeit([&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0); }, 2000));
    for (int g : {1, 256, 2048, 8192}) {
        float t = timeit([&] { hipLaunchKernelGGL(k_dep, dim3(g), dim3(256), 0, 0, a, b); hipLaunchKernelGGL(k_dep, dim3(g), dim3(256), 0, 0, b, a); }, 1000) / 2;
        printf("dependent word, %5d wg x 256:     %.2f us per launch\n", g, t);
    }
    return 0;
}
