// nmi_grid_kernel's pixel loop in isolation (uniform-noise 640x480 pair, every workgroup the same pair, 20 pairs back to back):
// cycles per 64-lane LDS atomic by workgroup size and by how many pixels' addresses are computed before their atomics issue.
//   NT = 1024: 16 wavefronts per CU under the 128-register cap (the kernel's shape); NT = 512: 8 wavefronts, 256 registers each.
//   BATCH = 1: per pixel 5 VALU, then its atomic (the kernel's order); 4 / 16: that many pixels' 5 VALU first, then their atomics.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/hist_loop.hip -o tools/ubench/hist_loop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
constexpr int kWordsLds = 32768;
template <int J>
__device__ __forceinline__ void addr_val(uint32_t r, uint32_t w, uint32_t &addr, uint32_t &val)
{
    // the kernel's five instructions (nmi_kernels.hip, add_chunk)
    uint32_t a1, a2, hi;
    const uint32_t nine = 9;
    if (J == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(a1) : "v"(nine), "v"(r));
    if (J == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(a1) : "v"(nine), "v"(r));
    if (J == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(a1) : "v"(nine), "v"(r));
    if (J == 3) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(a1) : "v"(nine), "v"(r));
    a2 = J == 0 ? (w << 2) : (w >> (8 * J - 2));
    addr = (a2 & 0x1FCu) | a1;
    hi = __builtin_amdgcn_ubfe(w, 8 * J + 7, 1);
    val = hi * 0xFFFFu + 1u;
}
template <int NT, int BATCH, int SETS = 2, bool COLD = false>
__global__ __launch_bounds__(NT) void k(const uint8_t *__restrict__ render0, const uint8_t *__restrict__ warped0, int nchunks, int reps, uint32_t *out,
                                        unsigned long long *cyc)
{
    __shared__ uint32_t h[kWordsLds];
    const int tid = threadIdx.x;
    for (int i = tid; i < kWordsLds; i += NT) h[i] = 0;
    __syncthreads();
    char *base = reinterpret_cast<char *>(h);
    const unsigned long long t0 = clock64();
    for (int rep = 0; rep < reps; ++rep) {
        // COLD: a pair of this workgroup's own for every repetition (256 workgroups x 20 x 2 x 300 KiB = 3 GB: nothing comes from a cache)
        const size_t shift = COLD ? ((size_t)blockIdx.x * reps + rep) * 2 * ((size_t)nchunks << 4) : 0;
        const uint8_t *render = render0 + shift, *warped = warped0 + shift + (COLD ? ((size_t)nchunks << 4) : 0);
        const int last = nchunks - 1;
        auto ld = [&](const uint8_t *p, int c) { return *reinterpret_cast<const uint4 *>(p + ((uint32_t)min(c, last) << 4)); };
        auto body = [&](const uint4 &rv, const uint4 &wv) {
            const uint32_t r[4] = {rv.x, rv.y, rv.z, rv.w}, w[4] = {wv.x, wv.y, wv.z, wv.w};
            uint32_t ad[16], va[16];
#pragma unroll
            for (int g = 0; g < 16; g += BATCH) {
#pragma unroll
                for (int u = g; u < g + BATCH; ++u) {
                    const int q = u >> 2;
                    if ((u & 3) == 0) addr_val<0>(r[q], w[q], ad[u], va[u]);
                    if ((u & 3) == 1) addr_val<1>(r[q], w[q], ad[u], va[u]);
                    if ((u & 3) == 2) addr_val<2>(r[q], w[q], ad[u], va[u]);
                    if ((u & 3) == 3) addr_val<3>(r[q], w[q], ad[u], va[u]);
                }
#pragma unroll
                for (int u = g; u < g + BATCH; ++u)
                    (void)__hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(base + ad[u]), va[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        };
        int ch = tid;
        const int iters = (nchunks + NT - 1) / NT;
        if (SETS == 2) {
            uint4 wa = ld(warped, ch), ra = ld(render, ch), wb, rb;
            for (int it = 0; it < iters; it += 2) {
                wb = ld(warped, ch + NT);
                rb = ld(render, ch + NT);
                if (ch < nchunks) body(ra, wa);
                wa = ld(warped, ch + 2 * NT);
                ra = ld(render, ch + 2 * NT);
                if (ch + NT < nchunks) body(rb, wb);
                ch += 2 * NT;
            }
        } else {
            // three sets: the loads of the chunk after next are issued before the current chunk's atomics
            uint4 wa = ld(warped, ch), ra = ld(render, ch), wb = ld(warped, ch + NT), rb = ld(render, ch + NT), wc, rc;
            for (int it = 0; it < iters; it += 3) {
                wc = ld(warped, ch + 2 * NT);
                rc = ld(render, ch + 2 * NT);
                if (ch < nchunks) body(ra, wa);
                wa = ld(warped, ch + 3 * NT);
                ra = ld(render, ch + 3 * NT);
                if (ch + NT < nchunks) body(rb, wb);
                wb = ld(warped, ch + 4 * NT);
                rb = ld(render, ch + 4 * NT);
                if (ch + 2 * NT < nchunks) body(rc, wc);
                ch += 3 * NT;
            }
        }
        __syncthreads();
    }
    const unsigned long long t1 = clock64();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    uint32_t s = 0;
    for (int i = tid; i < kWordsLds; i += NT) s += h[i];
    if (s == 0xDEADBEEFu) out[0] = s;
}
template <int NT, int BATCH, int SETS = 2, bool COLD = false>
void run(const uint8_t *r, const uint8_t *w, int nchunks, uint32_t *out, unsigned long long *cyc, int grid)
{
    const int reps = 20;
    unsigned long long h[256];
    hipLaunchKernelGGL((k<NT, BATCH, SETS, COLD>), dim3(grid), dim3(NT), 0, 0, r, w, nchunks, reps, out, cyc);
    hipLaunchKernelGGL((k<NT, BATCH, SETS, COLD>), dim3(grid), dim3(NT), 0, 0, r, w, nchunks, reps, out, cyc);
    (void)hipMemcpy(h, cyc, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < grid; ++i) m += (double)h[i];
    m /= grid;
    printf("%4d threads, %2d pixels' addresses ahead of their atomics, %d register sets, %s, grid %3d: %6.2f cycles per wavefront atomic (%.1f k cycles per pair)\n", NT, BATCH,
           SETS, COLD ? "pairs from memory" : "one pair, cached  ", grid, m / reps / (nchunks * 16.0 / 64.0), m / reps / 1e3);
}
int main()
{
    const int npix = 640 * 480, nchunks = npix / 16;
    uint8_t *hr = (uint8_t *)malloc(npix), *hw = (uint8_t *)malloc(npix);
    srand(1);
    for (int i = 0; i < npix; ++i) hr[i] = rand() >> 8, hw[i] = rand() >> 8;
    uint8_t *r, *w;
    uint32_t *out;
    unsigned long long *cyc;
    (void)hipMalloc(&r, npix), (void)hipMalloc(&w, npix), (void)hipMalloc(&out, 64), (void)hipMalloc(&cyc, 256 * 8);
    (void)hipMemcpy(r, hr, npix, hipMemcpyHostToDevice), (void)hipMemcpy(w, hw, npix, hipMemcpyHostToDevice);
    // COLD: 256 workgroups x 20 repetitions x (render, frame): filled on the device with the same noise, shifted
    uint8_t *big;
    const size_t big_bytes = (size_t)256 * 20 * 2 * npix;
    (void)hipMalloc(&big, big_bytes + 64);
    for (size_t o = 0; o < big_bytes; o += (size_t)2 * npix) {
        (void)hipMemcpyAsync(big + o, r, npix, hipMemcpyDeviceToDevice, 0);
        (void)hipMemcpyAsync(big + o + npix, w, npix, hipMemcpyDeviceToDevice, 0);
    }
    (void)hipDeviceSynchronize();
    for (int grid : {1, 256}) {
        run<1024, 1>(r, w, nchunks, out, cyc, grid);
        run<1024, 4>(r, w, nchunks, out, cyc, grid);
        run<1024, 16>(r, w, nchunks, out, cyc, grid);
        run<512, 1>(r, w, nchunks, out, cyc, grid);
        run<512, 4>(r, w, nchunks, out, cyc, grid);
        run<512, 16>(r, w, nchunks, out, cyc, grid);
        run<256, 16>(r, w, nchunks, out, cyc, grid);
        run<1024, 1, 3>(r, w, nchunks, out, cyc, grid);
        run<1024, 1, 2, true>(big, big, nchunks, out, cyc, grid);
        run<1024, 1, 3, true>(big, big, nchunks, out, cyc, grid);
        run<512, 1, 3, true>(big, big, nchunks, out, cyc, grid);
    }
    return 0;
}
