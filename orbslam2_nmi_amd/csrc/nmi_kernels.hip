// nmi_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the NMI pose-candidate scoring path.
//
// What is computed is fixed by the reference (gsanya/orbslam2_NMI, paths relative to its root):
//   Thirdparty/CUDA_Functions/NMI.cu:42-104   joint + marginal 256-bin histograms of (render, warped frame)
//   Thirdparty/CUDA_Functions/NMI.cu:230-267  per-bin term  (c/len) * log2f(c/len), len = W*H
//   Thirdparty/CUDA_Functions/NMI.cu:270-339  stride-halving fp32 trees (joint rows first, then row sums)
//   Thirdparty/CUDA_Functions/NMI.cu:342-362  SUC / ENMI score with the all-zero guard
//   Thirdparty/Localization/helperFunctions.cpp:50-103  arg-max (strict '>' from 0, lowest index on ties)
// How it is computed is new: one 1024-lane workgroup per pose candidate owns the whole
// 256x256 joint histogram in LDS as packed 16-bit counters (128 KiB of the CU's 160 KiB), with exact
// wrap bookkeeping so counts above 65535 stay exact; the entropy terms come from a per-context table
// indexed by count; the trees are evaluated in registers / cross-lane in the reference's order; the
// score, the rating-table store and the arg-max (one 64-bit atomicMax per candidate) are fused into the
// same launch.  Histogramming is integer scatter work: no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nmi_kernels.h"

namespace nmi {

namespace {

constexpr int kBlock = NMI_BLOCK_THREADS;  // 1024 lanes = 16 wavefronts, one workgroup per CU (LDS-limited)
constexpr int kWaves = kBlock / 64;
constexpr int kBins = 256;
constexpr int kWords = kBins * kBins / 2;  // two 16-bit counters per LDS word
constexpr int kOvfCap = 1024;              // >= 2 * floor(2^24 / 65536) + 1 wrap events per candidate
constexpr int kRowsPerWave = kBins / kWaves;

// LDS word of joint bin (d1 = render intensity, d2 = warped-frame intensity):
//   word = d1 * 128 + (d2 & 127), low half for d2 < 128, high half for d2 >= 128.
// Lane l of a wavefront that reads words d1*128 + l and d1*128 + 64 + l therefore holds bins
// d2 = l, l+64, l+128, l+192 of row d1 -- exactly the operands of the first two tree steps
// (a[t] += a[t+128], a[t] += a[t+64]; NMI.cu:276-284), so those steps need no cross-lane traffic.
// The LDS bank of a bin is (d2 & 31): neighbouring render intensities do not collide.

struct Lds {
    uint32_t joint[kWords];    // 128 KiB
    uint32_t hist_render[kBins];
    uint32_t hist_warped[kBins];
    float joint_row_sums[kBins];  // d_JointEntropyShort, kernel.cu:60,90
    uint32_t ovf[kOvfCap];        // wrap events: (word << 1) | field
    uint32_t ovf_n;
    float sums[3];
};

__device__ __forceinline__ float wave_tree_64(float x)
{
    // tree steps n = 32,16,...,1 over one value per lane: lane t < n takes a[t] += a[t + n].
#pragma unroll
    for (int n = 32; n >= 1; n >>= 1) x = x + __shfl_down(x, n, 64);
    return x;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t x)
{
#pragma unroll
    for (int n = 32; n >= 1; n >>= 1) x += __shfl_down(x, n, 64);
    return x;
}

// One pixel -> one LDS atomic on the packed joint histogram (the reference does three atomics per
// pixel, NMI.cu:46-48; the marginals are recovered as row / column sums of the joint).
// Each 16-bit field is only ever incremented by one, by an add that returns the old word, so every
// wrap of a field is seen by exactly one lane: a low-field wrap carries into the high field (the
// high field then counts d2>=128 hits plus low wraps), a high-field wrap is seen either by a high
// add (old high == 0xFFFF) or by the carrying low add (old word == 0xFFFFFFFF).  Events are rare
// (at most about 2 * W*H / 65536 per candidate) and are replayed when the counters are decoded.
template <bool BG, bool SHIFTED>
__device__ __forceinline__ void add_pixel(Lds &lds, uint32_t d1, uint32_t d2, int shift)
{
    if (!BG && (d1 == 0 || d2 == 0)) return;  // NMI.cu:85
    if (SHIFTED) {
        d1 >>= shift;
        d2 >>= shift;
    }
    const uint32_t word = (d1 << 7) | (d2 & 127u);
    const uint32_t val = (d2 & 128u) ? 0x10000u : 1u;
    const uint32_t old = __hip_atomic_fetch_add(&lds.joint[word], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint32_t field = val * 0xFFFFu;
    if (__builtin_expect((old & field) == field, 0)) {
        uint32_t k = __hip_atomic_fetch_add(&lds.ovf_n, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (k < kOvfCap) lds.ovf[k] = (word << 1) | (val >> 16);
        if (val == 1u && old == 0xFFFFFFFFu) {
            k = __hip_atomic_fetch_add(&lds.ovf_n, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (k < kOvfCap) lds.ovf[k] = (word << 1) | 1u;
        }
    }
}

template <bool BG, bool SHIFTED>
__device__ __forceinline__ void add_dword(Lds &lds, uint32_t r, uint32_t w, int shift)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) add_pixel<BG, SHIFTED>(lds, (r >> (8 * j)) & 0xFFu, (w >> (8 * j)) & 0xFFu, shift);
}

// Histogram phase for one candidate: histogram256Kernel's pixel loop, NMI.cu:79-87.
template <bool BG, bool SHIFTED>
__device__ __forceinline__ void histogram_phase(Lds &lds, const GridArgs &a, const uint8_t *__restrict__ render,
                                                const uint8_t *__restrict__ warped)
{
    const int tid = threadIdx.x;
    if (a.vec_ok) {
        // 16 pixels per lane per step: one 16-byte load from each image, 1 KiB per wavefront instruction.
        const int nchunks = a.npix >> 4;
        const uint4 *__restrict__ wp = reinterpret_cast<const uint4 *>(warped);
        for (int ch = tid; ch < nchunks; ch += kBlock) {
            int rch = ch;
            if (a.flip) {  // NMI.cu:82: row y of the frame meets row H-1-y of the bottom-up render
                const int y = (a.chunks_per_row == 1) ? ch : (int)__umulhi((uint32_t)ch, a.cpr_magic);
                const int cx = ch - y * a.chunks_per_row;
                rch = (a.height - 1 - y) * a.chunks_per_row + cx;
            }
            const uint4 wv = wp[ch];
            const uint4 rv = reinterpret_cast<const uint4 *>(render)[rch];
            add_dword<BG, SHIFTED>(lds, rv.x, wv.x, a.shift);
            add_dword<BG, SHIFTED>(lds, rv.y, wv.y, a.shift);
            add_dword<BG, SHIFTED>(lds, rv.z, wv.z, a.shift);
            add_dword<BG, SHIFTED>(lds, rv.w, wv.w, a.shift);
        }
    } else {
        // Any width / alignment: byte loads, position arithmetic as written in NMI.cu:79-83.
        for (int pos = tid; pos < a.npix; pos += kBlock) {
            const int y = pos / a.width;
            const int x = pos - y * a.width;
            const int ry = a.flip ? (a.height - 1 - y) : y;
            add_pixel<BG, SHIFTED>(lds, render[ry * a.width + x], warped[pos], a.shift);
        }
    }
}

__device__ __forceinline__ float term(const float *__restrict__ table, uint32_t c)
{
    // ComputeEntropyKernel, NMI.cu:242-263; table[c] = (c/len) * log2f(c/len), table[0] = 0.
    return c ? table[c] : 0.0f;
}

// Replays the wrap events of one LDS word onto its two decoded counters.
__device__ __forceinline__ void apply_wraps(const Lds &lds, uint32_t novf, uint32_t word, uint32_t &lo, uint32_t &hi)
{
    for (uint32_t e = 0; e < novf; ++e) {
        const uint32_t ev = lds.ovf[e];
        if ((ev >> 1) == word) {
            if (ev & 1u) {
                hi += 65536u;
            } else {
                lo += 65536u;
                hi -= 1u;  // the carry that the low wrap pushed into the high field
            }
        }
    }
}

}  // namespace

// One workgroup per candidate (grid-stride over the candidates of this launch).
template <bool BG, bool SHIFTED>
__global__ __launch_bounds__(NMI_BLOCK_THREADS) void nmi_grid_kernel(GridArgs a)
{
    __shared__ Lds lds;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    for (int i = tid; i < kWords; i += kBlock) lds.joint[i] = 0;
    if (tid < kBins) lds.hist_warped[tid] = 0;
    if (tid == 0) lds.ovf_n = 0;
    __syncthreads();

    const int total = a.S_local * a.Wn;
    for (int p = blockIdx.x; p < total; p += gridDim.x) {
        const int w = p / a.S_local;
        const int s = p - w * a.S_local;
        const uint8_t *render = a.render_stack + (size_t)s * a.npix;
        const uint8_t *warped = a.warp_stack + (size_t)w * a.npix;

        histogram_phase<BG, SHIFTED>(lds, a, render, warped);
        __syncthreads();

        // ---- decode + per-bin terms + row trees (ComputeEntropyKernel + AddvectorParwiseMidKernel) ----
        const uint32_t novf = lds.ovf_n < (uint32_t)kOvfCap ? lds.ovf_n : (uint32_t)kOvfCap;
        uint32_t col0 = 0, col1 = 0, col2 = 0, col3 = 0;  // column sums for d2 = lane, +64, +128, +192
#pragma unroll 4
        for (int k = 0; k < kRowsPerWave; ++k) {
            const int d1 = wave * kRowsPerWave + k;
            const uint32_t i0 = d1 * 128 + lane, i1 = i0 + 64;
            const uint32_t w0 = lds.joint[i0], w1 = lds.joint[i1];
            lds.joint[i0] = 0;  // ready for the next candidate
            lds.joint[i1] = 0;
            uint32_t c0 = w0 & 0xFFFFu, c2 = w0 >> 16, c1 = w1 & 0xFFFFu, c3 = w1 >> 16;
            if (novf) {
                apply_wraps(lds, novf, i0, c0, c2);
                apply_wraps(lds, novf, i1, c1, c3);
            }
            col0 += c0;
            col1 += c1;
            col2 += c2;
            col3 += c3;
            const uint32_t rsum = wave_sum_u32(c0 + c1 + c2 + c3);
            const float e0 = term(a.table, c0), e1 = term(a.table, c1), e2 = term(a.table, c2), e3 = term(a.table, c3);
            const float x0 = e0 + e2;  // n = 128: a[t] += a[t+128], t = lane
            const float x1 = e1 + e3;  //          a[t] += a[t+128], t = lane + 64
            float x = x0 + x1;         // n = 64
            x = wave_tree_64(x);       // n = 32..1
            if (lane == 0) {
                lds.hist_render[d1] = rsum;
                lds.joint_row_sums[d1] = x;
            }
            if (a.dbg_joint) {
                uint32_t *row = a.dbg_joint + d1 * kBins;
                row[lane] = c0;
                row[lane + 64] = c1;
                row[lane + 128] = c2;
                row[lane + 192] = c3;
            }
        }
        atomicAdd(&lds.hist_warped[lane], col0);
        atomicAdd(&lds.hist_warped[lane + 64], col1);
        atomicAdd(&lds.hist_warped[lane + 128], col2);
        atomicAdd(&lds.hist_warped[lane + 192], col3);
        __syncthreads();

        // ---- three 256-element trees (AddVectorPairwiseKernel, NMI.cu:295-339) ----
        if (wave < 3) {
            float v0, v1, v2, v3;
            if (wave == 2) {
                v0 = lds.joint_row_sums[lane];
                v1 = lds.joint_row_sums[lane + 64];
                v2 = lds.joint_row_sums[lane + 128];
                v3 = lds.joint_row_sums[lane + 192];
            } else {
                const uint32_t *h = wave == 0 ? lds.hist_render : lds.hist_warped;
                v0 = term(a.table, h[lane]);
                v1 = term(a.table, h[lane + 64]);
                v2 = term(a.table, h[lane + 128]);
                v3 = term(a.table, h[lane + 192]);
                uint32_t *o = wave == 0 ? a.dbg_h1 : a.dbg_h2;
                if (o) {
                    o[lane] = h[lane];
                    o[lane + 64] = h[lane + 64];
                    o[lane + 128] = h[lane + 128];
                    o[lane + 192] = h[lane + 192];
                }
            }
            const float x0 = v0 + v2, x1 = v1 + v3;
            const float x = wave_tree_64(x0 + x1);
            if (lane == 0) lds.sums[wave] = x;
        }
        __syncthreads();

        if (tid == 0) {
            // NMI.cu:342-362, evaluated from the three completed sums (the reference reads them
            // across blocks without synchronisation, NMI.cu:340-342).
            const float a1 = lds.sums[0], a2 = lds.sums[1], a3 = lds.sums[2];
            float score;
            if (a1 == 0.0f && a2 == 0.0f && a3 == 0.0f)
                score = 0.0f;
            else if (a.mode == NMI_MODE_ENMI_)
                score = ((-a1) + (-a2)) / (-a3);
            else if (a.mode == NMI_MODE_SUC_)
                score = 2.0f * (1.0f - ((-a3) / ((-a1) + (-a2))));
            else
                score = -1.0f;
            if (a.ratings) a.ratings[p] = score;
            if (a.dbg_sums) {
                a.dbg_sums[0] = a1;
                a.dbg_sums[1] = a2;
                a.dbg_sums[2] = a3;
            }
            // find_max_elements, helperFunctions.cpp:52-101: max starts at 0, strict '>', first cell
            // equal to the max wins.  Non-negative floats order like their bit patterns, so one
            // 64-bit max of (score bits, inverted global index) reproduces it; negative / NaN scores
            // contribute nothing.
            if (score >= 0.0f) {
                const uint32_t bits = score == 0.0f ? 0u : __float_as_uint(score);
                const uint32_t gidx = (uint32_t)w * (uint32_t)a.S_total + (uint32_t)(a.s_offset + s);
                const unsigned long long key = ((unsigned long long)bits << 32) | (unsigned long long)(0xFFFFFFFFu - gidx);
                atomicMax(a.key, key);
            }
        }
        if (tid < kBins) lds.hist_warped[tid] = 0;
        if (tid == 0) lds.ovf_n = 0;
        __syncthreads();
    }
}

// table[c] = (c/len) * log2(c/len) in the reference's fp32 form (NMI.cu:245): p = fl32(c/len),
// l = log2 of p rounded once to fp32 (evaluated in fp64 so the rounding is the correct one; CUDA's
// and glibc's log2f are each within 1 ulp of it), term = fl32(p * l).
__global__ void nmi_table_kernel(float *table, int npix)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > npix) return;
    if (c == 0) {
        table[0] = 0.0f;
        return;
    }
    const float p = (float)c / (float)npix;
    const float l = (float)log2((double)p);
    table[c] = p * l;
}

hipError_t launch_table(float *table, int npix, hipStream_t stream)
{
    const int threads = 256;
    const int blocks = (npix + 1 + threads - 1) / threads;
    hipLaunchKernelGGL(nmi_table_kernel, dim3(blocks), dim3(threads), 0, stream, table, npix);
    return hipGetLastError();
}

hipError_t launch_grid(const GridArgs &a, int workgroups, bool use_bg, hipStream_t stream)
{
    const bool shifted = a.shift != 0;
    dim3 grid(workgroups), block(kBlock);
    if (use_bg) {
        if (shifted)
            hipLaunchKernelGGL((nmi_grid_kernel<true, true>), grid, block, 0, stream, a);
        else
            hipLaunchKernelGGL((nmi_grid_kernel<true, false>), grid, block, 0, stream, a);
    } else {
        if (shifted)
            hipLaunchKernelGGL((nmi_grid_kernel<false, true>), grid, block, 0, stream, a);
        else
            hipLaunchKernelGGL((nmi_grid_kernel<false, false>), grid, block, 0, stream, a);
    }
    return hipGetLastError();
}

int grid_kernel_lds_bytes() { return (int)sizeof(Lds); }

}  // namespace nmi
