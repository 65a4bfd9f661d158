// valu_rate.hip -- cycles per wave-instruction of the VALU forms the histogram loops are made of, measured per SIMD with
// 1, 2, 4 wavefronts per SIMD (256 / 512 / 1024-lane workgroups on one CU).  hipcc --offload-arch=gfx950 -O2 -o valu_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int KIND>
__global__ void k(uint32_t *out, long long *cyc, int iters, uint32_t seed)
{
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x5bd1e995u, c = a + 77, d = b + 99, e = 0, two = 2, sel = 0x05010400u;
    uint32_t f = a * 3, g = b * 5, h = c * 7;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {  // independent v_xor (4 chains)
            REP16(asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(two));)
        } else if (KIND == 1) {  // v_perm_b32, 4 chains
            REP16(asm volatile("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(two), "v"(sel));)
        } else if (KIND == 2) {  // SDWA shift, 4 chains
            REP16(asm volatile("v_lshlrev_b32_sdwa %0, %4, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                               "v_lshlrev_b32_sdwa %1, %4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
                               "v_lshlrev_b32_sdwa %2, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n"
                               "v_lshlrev_b32_sdwa %3, %4, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(two));)
        } else if (KIND == 3) {  // one dependent chain of v_xor
            REP16(asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(two));)
        } else if (KIND == 4) {  // v_cmp -> s_and_saveexec -> v_add -> s_or exec (the predicated pixel)
            REP16(asm volatile("v_cmp_gt_u32 vcc, %1, %0\n s_and_saveexec_b64 s[20:21], vcc\n v_add_u32 %0, %0, %2\n s_or_b64 exec, exec, s[20:21]\n"
                               "v_cmp_gt_u32 vcc, %1, %3\n s_and_saveexec_b64 s[20:21], vcc\n v_add_u32 %3, %3, %2\n s_or_b64 exec, exec, s[20:21]" : "+v"(a) : "v"(b), "v"(two), "v"(c) : "vcc", "s20", "s21");)
        } else if (KIND == 5) {  // v_and_or_b32 + v_mad_u32_u24 + v_bfe (main kernel's mix), independent
            REP16(asm volatile("v_and_or_b32 %0, %0, %4, %1\n v_mad_u32_u24 %1, %1, %4, 1\n v_bfe_u32 %2, %2, 7, 1\n v_lshrrev_b32 %3, 6, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(two));)
        } else if (KIND == 6) {  // the split kernel's dword: xor, 2 perm, 4 sdwa, 4 cmp (no exec change), all dependent as in the loop
            REP16(asm volatile("v_xor_b32 %0, %1, %4\n v_perm_b32 %2, %0, %5, %6\n v_perm_b32 %3, %0, %5, %7\n"
                               "v_lshlrev_b32_sdwa %0, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
                               "v_lshlrev_b32_sdwa %1, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                               "v_lshlrev_b32_sdwa %2, %4, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
                               "v_lshlrev_b32_sdwa %3, %4, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                               "v_cmp_gt_u32 vcc, %5, %0\n v_cmp_gt_u32 vcc, %5, %1\n v_cmp_gt_u32 vcc, %5, %2\n v_cmp_gt_u32 vcc, %5, %3"
                               : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(two), "v"(f), "v"(g), "v"(h) : "vcc");)
        }
    }
    const long long t1 = clock64();
    e = a ^ b ^ c ^ d;
    if (e == 0x12345678u) out[0] = e;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// the same loop on `blocks` workgroups of 1024 lanes at once: does the rate per SIMD hold when the whole chip is busy?
template <int KIND>
static void run_chip(const char *name, int n_per_iter)
{
    uint32_t *out;
    long long *cyc;
    hipMalloc(&out, 4);
    hipMalloc(&cyc, 8 * 4096);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int blocks : {1, 8, 64, 256, 512}) {
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(1024), 0, 0, out, cyc, iters, 1u);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(1024), 0, 0, out, cyc, iters, 2u);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        long long c = 0;
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double instr_per_wave = (double)iters * 16 * n_per_iter;
        printf("%-28s %4d workgroups x 16 waves: %6.2f ticks per instr per wave, wall %8.1f us -> %6.2f ns per instr per wave, %7.1f G wave-instr/s chip\n",
               name, blocks, (double)c / instr_per_wave, ms * 1e3, ms * 1e6 / instr_per_wave, blocks * 16 * instr_per_wave / (ms * 1e-3) / 1e9);
    }
    hipFree(out);
    hipFree(cyc);
}

template <int KIND>
static void run(const char *name, int n_per_iter)
{
    uint32_t *out;
    long long *cyc;
    hipMalloc(&out, 4);
    hipMalloc(&cyc, 8 * 64);
    const int iters = 200;
    for (int threads : {256, 512, 1024}) {
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, out, cyc, iters, 1u);
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, out, cyc, iters, 2u);
        hipDeviceSynchronize();
        long long c = 0;
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double per_wave_instr = (double)c / ((double)iters * 16 * n_per_iter);
        const int waves_per_simd = threads / 256;
        printf("%-44s %d wave(s)/SIMD: %6.2f cycles per instr per wave  -> %5.2f cycles per instr per SIMD\n", name, waves_per_simd, per_wave_instr,
               per_wave_instr / waves_per_simd);
    }
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    run<0>("v_xor_b32 x4 independent", 4);
    run<3>("v_xor_b32 dependent chain", 4);
    run<1>("v_perm_b32 x4", 4);
    run<2>("v_lshlrev_b32_sdwa x4", 4);
    run<5>("and_or / mad_u24 / bfe / lshr", 4);
    run<4>("cmp+saveexec+add+s_or (VALU only counted: 4)", 4);
    run<6>("split dword: xor 2perm 4sdwa 4cmp", 11);
    run_chip<0>("v_xor_b32 x4 independent", 4);
    run_chip<5>("and_or/mad_u24/bfe/lshr", 4);
    return 0;
}
