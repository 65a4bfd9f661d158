// rcp_check -- exhaustive check of the short reciprocal the producers use (nmi_warp_device.h: warp_rcp_ok / warp_rcp_fast /
// warp_rcp) against the correctly rounded division 1.0f / x, for every one of the 2^32 floats, on the GPU it runs on.
// Prints "RCP OK" and returns 0 when: wherever warp_rcp_ok(x) holds, warp_rcp_fast(x) has the bits of 1.0f / x; and warp_rcp(x)
// has them everywhere (two NaNs count as equal).  Run by tests/test_render.py::test_gpu_reciprocal_exhaustive.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "nmi_warp_device.h"

__global__ void check(unsigned long long *counts, uint32_t *first)
{
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    unsigned long long fast_bad = 0, any_bad = 0, fast_n = 0;
    for (uint64_t b = t; b < (1ull << 32); b += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t bits = (uint32_t)b;
        const float x = __uint_as_float(bits);
        const float want = 1.0f / x;
        if (nmi::warp_rcp_ok(x)) {
            ++fast_n;
            if (__float_as_uint(nmi::warp_rcp_fast(x)) != __float_as_uint(want)) {
                if (fast_bad == 0) atomicMin(first, bits);
                ++fast_bad;
            }
        }
        const float got = nmi::warp_rcp(x);
        if (__float_as_uint(got) != __float_as_uint(want) && !(got != got && want != want)) ++any_bad;
    }
    atomicAdd(&counts[0], fast_bad), atomicAdd(&counts[1], any_bad), atomicAdd(&counts[2], fast_n);
}

int main()
{
    unsigned long long *counts, h[3] = {0, 0, 0};
    uint32_t *first, hf = 0xFFFFFFFFu;
    if (hipMalloc((void **)&counts, sizeof h) != hipSuccess || hipMalloc((void **)&first, 4) != hipSuccess) return 2;
    if (hipMemcpy(counts, h, sizeof h, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(first, &hf, 4, hipMemcpyHostToDevice) != hipSuccess) return 2;
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, counts, first);
    if (hipMemcpy(h, counts, sizeof h, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    printf("floats on the short path: %llu; short path != division: %llu (lowest bit pattern %08x); warp_rcp != division: %llu\n", h[2], h[0], hf, h[1]);
    if (h[0] || h[1] || h[2] == 0) return 1;
    printf("RCP OK\n");
    return 0;
}
