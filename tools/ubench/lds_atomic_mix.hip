// Cycles per 64-lane non-returning ds_add_u32 with 16 wavefronts per CU issuing them, by the amount of VALU work between two atomics:
// (a) none (16 addresses per lane computed once), (b) N extra VALU instructions per atomic.  The addresses are uniformly random words of
// a 128 KiB array (no two lanes on one word, banks at random).
// hipcc --offload-arch=gfx950 -O2 tools/ubench/lds_atomic_mix.hip -o tools/ubench/lds_atomic_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
constexpr int kWordsLds = 32768;
template <int NVALU>
__global__ __launch_bounds__(1024) void k(uint32_t *out, int n, unsigned long long *cyc)
{
    __shared__ uint32_t h[kWordsLds];
    for (int i = threadIdx.x; i < kWordsLds; i += 1024) h[i] = 0;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    uint32_t addr[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        x = x * 1664525u + 1013904223u;
        // distinct words per lane within an instruction: word = (random row) * 64 + lane-derived column, column permuted per u
        addr[u] = ((((x >> 10) & 511u) << 6) | ((threadIdx.x * 17u + u * 5u) & 63u)) << 2;
    }
    char *base = reinterpret_cast<char *>(h);
    uint32_t v = x | 1u;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            uint32_t a = addr[u];
#pragma unroll
            for (int e = 0; e < NVALU; ++e) {
                // cheap full-rate ops that depend on v so that they cannot be hoisted, and leave the address a valid word
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v) : "v"(a));
            }
            if (NVALU > 0) a = (a & 0x1FFFCu) ^ (v & 0u);
            (void)__hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(base + a), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    uint32_t s = v;
    for (int i = threadIdx.x; i < kWordsLds; i += 1024) s += h[i];
    if (s == 0xDEADBEEFu) out[0] = s;
}
template <int NVALU>
void run(uint32_t *out, unsigned long long *cyc, int grid)
{
    const int n = 300;
    unsigned long long h[256];
    hipLaunchKernelGGL(k<NVALU>, dim3(grid), dim3(1024), 0, 0, out, n, cyc);
    hipLaunchKernelGGL(k<NVALU>, dim3(grid), dim3(1024), 0, 0, out, n, cyc);
    (void)hipMemcpy(h, cyc, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < grid; ++i) m += (double)h[i];
    m /= grid;
    printf("%2d extra VALU instructions per atomic, grid %3d: %6.2f cycles per wavefront atomic per CU\n", NVALU, grid, m / (16.0 * n * 16));
}
int main()
{
    uint32_t *out;
    unsigned long long *cyc;
    (void)hipMalloc(&out, 64);
    (void)hipMalloc(&cyc, 256 * sizeof(unsigned long long));
    for (int grid : {1, 256}) {
        run<0>(out, cyc, grid);
        run<2>(out, cyc, grid);
        run<4>(out, cyc, grid);
        run<5>(out, cyc, grid);
        run<6>(out, cyc, grid);
        run<8>(out, cyc, grid);
        run<12>(out, cyc, grid);
    }
    return 0;
}
