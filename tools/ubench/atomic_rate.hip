// atomic_rate.hip -- rate of scattered 32-bit atomicMin on a 44 MB buffer (the point-cloud renderer's anchor buffer):
// agent scope (performed at the memory side: the XCDs' L2s are not coherent with each other) against workgroup scope
// (performed in the issuing XCD's L2; only valid when all writers of an address sit on one XCD).
// hipcc --offload-arch=gfx950 -O2 -o atomic_rate atomic_rate.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int SCOPE, bool PER_XCD>
__global__ void k(uint32_t *buf, size_t words, int per_lane, uint32_t seed)
{
    uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + seed;
    uint32_t xcc = 0;
    if (PER_XCD) {
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
    }
    const size_t slice = words / 8;
    for (int i = 0; i < per_lane; ++i) {
        x = x * 1664525u + 1013904223u;
        size_t idx = PER_XCD ? (size_t)xcc * slice + (x % (uint32_t)slice) : x % (uint32_t)words;
        if (SCOPE == 0)
            (void)__hip_atomic_fetch_min(&buf[idx], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
            (void)__hip_atomic_fetch_min(&buf[idx], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

template <int SCOPE, bool PER_XCD>
static void run(const char *name, uint32_t *buf, size_t words)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 2048, threads = 256, per_lane = 16;  // 8.4 M atomics
    hipMemset(buf, 0xFF, words * 4);
    hipLaunchKernelGGL((k<SCOPE, PER_XCD>), dim3(blocks), dim3(threads), 0, 0, buf, words, per_lane, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<SCOPE, PER_XCD>), dim3(blocks), dim3(threads), 0, 0, buf, words, per_lane, 7u);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * threads * per_lane;
    printf("%-64s %8.1f us for %.1f M atomics = %6.1f G atomics/s\n", name, ms * 1e3, n / 1e6, n / (ms * 1e-3) / 1e9);
}

int main()
{
    const size_t words = (size_t)11 << 20;  // 44 MB
    uint32_t *buf;
    hipMalloc(&buf, words * 4);
    run<0, false>("agent scope, whole buffer", buf, words);
    run<1, false>("workgroup scope, whole buffer (not coherent: rate only)", buf, words);
    run<0, true>("agent scope, each XCD in its own eighth", buf, words);
    run<1, true>("workgroup scope, each XCD in its own eighth (coherent use)", buf, words);
    hipFree(buf);
    return 0;
}
