// v_permlane32_swap_b32 semantics on gfx950: prints, per lane, which (register, lane) each output came from.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *o)
{
    unsigned x = 1000 + threadIdx.x, y = 2000 + threadIdx.x;
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    o[threadIdx.x] = x; o[64 + threadIdx.x] = y;
    unsigned a = 1000 + threadIdx.x, b = 2000 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[128 + threadIdx.x] = r[0]; o[192 + threadIdx.x] = r[1];
}
int main()
{
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int part = 0; part < 4; ++part) {
        printf("%s:", part == 0 ? "asm x" : part == 1 ? "asm y" : part == 2 ? "builtin [0]" : "builtin [1]");
        for (int l = 0; l < 64; l += 8) printf(" l%d=%u", l, h[part * 64 + l]);
        printf("\n");
    }
    return 0;
}
