#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[64 * 64];     // [row][col], 64 cols (128-B rows)
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (short)i;   // value = row * 64 + col
    __syncthreads();
    const int l = threadIdx.x, t = l & 15, G = l >> 4;
    // group G reads the block rows [4G, 4G+4) x cols [0, 16): lane t supplies row 4G + (t >> 2), cols 4 (t & 3) ..
    const short* p = lds + (4 * G + (t >> 2)) * 64 + 4 * (t & 3);
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)p);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" (r%d,c%d)", h[l*4+j] / 64, h[l*4+j] % 64); printf("\n"); }
    return 0;
}
