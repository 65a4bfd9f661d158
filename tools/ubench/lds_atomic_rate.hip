// What an LDS atomic costs: cycles per 64-lane non-returning ds_add_u32 with every CU's 16 wavefronts issuing them back to back,
// by address pattern.  hipcc --offload-arch=gfx950 -O2 tools/ubench/lds_atomic_rate.hip -o tools/ubench/lds_atomic_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
constexpr int kWordsLds = 32768;  // 128 KiB, like the packed joint histogram
template <int MODE>
__global__ __launch_bounds__(1024) void k(uint32_t *out, int n, unsigned long long *cyc)
{
    __shared__ uint32_t h[kWordsLds];
    for (int i = threadIdx.x; i < kWordsLds; i += 1024) h[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            uint32_t a;
            x = x * 1664525u + 1013904223u;
            if (MODE == 0) a = (uint32_t)lane + 64u * ((x >> 10) & 255u);                 // conflict-free: bank = lane mod 32, any row
            else if (MODE == 1) a = (x >> 9) & (kWordsLds - 1);                           // uniformly random words
            else if (MODE == 2) a = ((x >> 9) & (kWordsLds - 1) & ~31u) | (lane & 7u);    // 8 banks only, random rows: 4-way at least
            else if (MODE == 3) a = 5u;                                                    // one address
            else a = ((x >> 9) & (kWordsLds - 1) & ~31u) | (lane & 15u);                  // 16 banks: 2-way at least
            (void)__hip_atomic_fetch_add(&h[a], (x >> 31) ? 0x10000u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    uint32_t s = 0;
    for (int i = threadIdx.x; i < kWordsLds; i += 1024) s += h[i];
    if (s == 0xDEADBEEFu) out[0] = s;
}
template <int MODE>
void run(const char *name, uint32_t *out, unsigned long long *cyc, int grid)
{
    const int n = 300;
    unsigned long long h[256];
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(1024), 0, 0, out, n, cyc);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(1024), 0, 0, out, n, cyc);
    (void)hipMemcpy(h, cyc, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < grid; ++i) m += (double)h[i];
    m /= grid;
    printf("%-44s grid %3d: %6.2f cycles per wavefront atomic per CU (16 wavefronts, %d each)\n", name, grid, m / (16.0 * n * 16), n * 16);
}
int main()
{
    uint32_t *out;
    unsigned long long *cyc;
    (void)hipMalloc(&out, 64);
    (void)hipMalloc(&cyc, 256 * sizeof(unsigned long long));
    for (int grid : {1, 256}) {
        run<0>("conflict-free (bank = lane mod 32)", out, cyc, grid);
        run<4>("16 banks (2-way at least)", out, cyc, grid);
        run<2>("8 banks (4-way at least)", out, cyc, grid);
        run<1>("uniformly random words", out, cyc, grid);
        run<3>("one address", out, cyc, grid);
    }
    return 0;
}
