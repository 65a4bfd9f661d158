#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *out) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = x;
}
int main() {
    unsigned *d, h[64];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; ++i) printf("%d:%08x%s", i, h[i], (i % 8 == 7) ? "\n" : "  ");
    return 0;
}
