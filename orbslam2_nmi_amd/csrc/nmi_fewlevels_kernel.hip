// nmi_fewlevels_kernel.hip -- scoring path for frames with FEW distinct intensities (posterised, thresholded, dark or
// saturated-and-quantised content), where nmi_grid_kernel's lanes queue on a handful of LDS addresses (DESIGN.md
// section 4: 3x the time at 16 grey levels, 8x at 4).  Same results, bit for bit: the same counts go through the same term
// table and the same trees (NMI.cu:242-263, :270-287, :290-362); only the way the counts are collected differs.
//
//   nmi_levels_kernel     which of the 256 intensities occur in the render stack / in the warp stack (one pass over the
//                         S + Wn images); its last workgroup turns the two presence masks into a LevelPlan: intensity
//                         -> rank tables, nr x nw = size of the joint histogram that can be non-zero, and the verdict
//                         `use` (nr * nw <= the context's limit).  Also posts (nr, nw) to pinned host memory: that is
//                         the hint the host picks the path of the NEXT search from (nmi_capi.cpp).
//   nmi_rank_kernel       rewrites both stacks as rank images (render rows put top-down on the way, NMI.cu:82).
//   nmi_fewlevels_kernel  one workgroup per candidate like nmi_grid_kernel, but the nr x nw counters are 32 bits wide
//                         (no wrap bookkeeping) and kept in R = 8..32 interleaved copies, copy = lane mod R: with R = 32
//                         every lane of an LDS access group owns a bank, so 64 lanes adding to ONE bin cost what 64
//                         lanes adding to 64 bins cost.  The copies are summed, the counts put back at their intensities
//                         in 256-wide rows (absent intensities are the zeros the reference would add), then the trees.
// With fewer than 256 bins (intensity >> shift) read "bin" for "intensity": textured content then qualifies from 64 bins down.
// All three exit at once when the plan says `use` = 0; the host enqueues nmi_grid_kernel behind them in its gated form,
// which runs exactly then.  Nothing here waits for another workgroup except through kernel boundaries.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nmi_device.h"

namespace nmi {

namespace {

constexpr int kBlock = NMI_BLOCK_THREADS;
constexpr int kWaves = kBlock / 64;
constexpr int kBins = 256;
constexpr int kCounterWords = 32768;       // 128 KiB of 32-bit counters: bins * copies
constexpr int kMaxJoint = 4096;            // nr * nw the kernel accepts (copies >= 8)
constexpr uint32_t kAbsent = 0xFFFFu;

struct FewLds {
    uint32_t cnt[kCounterWords];      // [bin][copy]
    uint32_t joint[kMaxJoint];        // copies summed: [rank_r][rank_w]
    uint32_t hist_render[kBins];      // by intensity
    uint32_t hist_warped[kBins];
    float joint_row_sums[kBins];      // d_JointEntropyShort, kernel.cu:60,90
    uint16_t rank_w[kBins];           // intensity -> rank in the warp stack, kAbsent when it does not occur
    uint8_t level_r[kBins];           // rank -> intensity in the render stack
};

// ---- probe --------------------------------------------------------------------------------------------------------
// One workgroup per (image, slice).  Presence is an LDS table of 256 intensities x 32 copies, copy = lane mod 32: every
// lane of an LDS access group stores into a bank of its own, so the pass costs the same whatever the content (stores of
// many lanes to ONE word are served one lane at a time: 100 us instead of 5 for a posterised 27 + 27 image search).
constexpr int kProbeCopies = 32;
constexpr int kProbeBlock = 1024;
__global__ __launch_bounds__(kProbeBlock) void nmi_levels_kernel(const uint8_t *__restrict__ render_stack, int S,
                                                          const uint8_t *__restrict__ warp_stack, int Wn, int npix, int slices,
                                                          LevelPlan *plan, unsigned long long *post, uint32_t seq,
                                                          uint32_t max_joint, int commit, int shift)
{
    __shared__ uint32_t present[kBins * kProbeCopies];
    __shared__ uint32_t am_last;
    const int tid = threadIdx.x;
    {
        uint4 *p4 = reinterpret_cast<uint4 *>(present);
        for (int i = tid; i < kBins * kProbeCopies / 4; i += kProbeBlock) p4[i] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    const int image = blockIdx.x / slices, slice = blockIdx.x - image * slices;
    const int stack = image < S ? 0 : 1;
    const uint8_t *img = stack == 0 ? render_stack + (size_t)image * npix : warp_stack + (size_t)(image - S) * npix;
    const int nchunks = npix >> 4;  // callers guarantee npix % 16 == 0 and 16-byte aligned stacks
    const int per = (nchunks + slices - 1) / slices;
    const int c0 = slice * per, c1 = min(c0 + per, nchunks);
    uint32_t *const mine = present + (tid & (kProbeCopies - 1));
    auto mark = [&](const uint4 &v) {
        const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int j = 0; j < 4; ++j) mine[(((d[q] >> (8 * j)) & 0xFFu) >> shift) * kProbeCopies] = 1;  // fire and forget
        }
    };
    // four loads in flight per lane; out-of-range chunks re-read the slice's last one
    for (int c = c0 + tid; c < c1; c += 4 * kProbeBlock) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const uint4 *>(img + ((size_t)min(c + u * kProbeBlock, c1 - 1) << 4));
#pragma unroll
        for (int u = 0; u < 4; ++u) mark(v[u]);
    }
    __syncthreads();
    // merge: lane v ORs the 32 copies of intensity v; wavefront k of this workgroup holds intensities 64k .. 64k + 63
    uint32_t any = 0;
    if (tid < kBins) {
        const uint4 *p4 = reinterpret_cast<const uint4 *>(present + tid * kProbeCopies);
#pragma unroll
        for (int i = 0; i < kProbeCopies / 4; ++i) {
            const uint4 v = p4[(i + tid) & (kProbeCopies / 4 - 1)];  // rotated start: lanes spread over the banks
            any |= v.x | v.y | v.z | v.w;
        }
    }
    // Masks of the four wavefronts -> 8 words in LDS -> 8 device-scope ORs by lanes 0..7.  The ORs are returning atomics
    // performed at the memory side: once their results are back (vmcnt 0) they have been performed, so the workgroup that
    // draws the last ticket finds every workgroup's bits in the masks.  No fence: a release fence writes the L2 back.
    __shared__ uint32_t wg_mask[8];
    const unsigned long long m = __ballot(any != 0);
    if ((tid & 63) == 0 && tid < kBins) {
        wg_mask[2 * (tid >> 6)] = (uint32_t)m;
        wg_mask[2 * (tid >> 6) + 1] = (uint32_t)(m >> 32);
    }
    __syncthreads();
    if (tid < 64) {
        if (tid < 8) {
            const uint32_t bits = wg_mask[tid];
            const uint32_t old = __hip_atomic_fetch_or(plan->mask[stack] + tid, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("" ::"v"(old));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) {
            const uint32_t arrived = __hip_atomic_fetch_add(&plan->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            am_last = arrived == gridDim.x - 1 ? 1u : 0u;
        }
    }
    __syncthreads();
    if (!am_last) return;
    // Last workgroup: masks -> plan.  Lane v speaks for intensity v.
    __shared__ uint32_t mk[2][8];
    if (tid < 16) {
        uint32_t *src = &plan->mask[0][0] + tid;
        mk[tid >> 3][tid & 7] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(src, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next probe
    }
    __syncthreads();
    // With fewer than 256 bins (intensity >> shift, NMI.cu:46-48 via the bin count) the masks are over bins; every raw
    // intensity gets the rank of its bin, and level[] holds bins.
    uint32_t n[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        const uint32_t bin = (uint32_t)(tid & 255) >> shift;
        uint32_t below = 0, all = 0;
        for (int k = 0; k < 8; ++k) {
            const uint32_t w = mk[st][k];
            all += __popc(w);
            if (k < (int)(bin >> 5)) below += __popc(w);
        }
        const uint32_t mine = mk[st][bin >> 5];
        below += __popc(mine & ((1u << (bin & 31)) - 1u));
        const bool here = (mine >> (bin & 31)) & 1u;
        uint8_t *rank = st == 0 ? plan->rank_r : plan->rank_w;
        uint8_t *level = st == 0 ? plan->level_r : plan->level_w;
        if (tid < kBins) {
            rank[tid] = here ? (uint8_t)below : (uint8_t)0;
            if (here && (uint32_t)tid == (bin << shift)) level[below] = (uint8_t)bin;
        }
        n[st] = all;
    }
    if (tid == 0) {
        const uint32_t joint = n[0] * n[1];
        uint32_t copies = 32;
        while (copies > 8 && joint * copies > (uint32_t)kCounterWords) copies >>= 1;
        const bool fits = joint > 0 && joint <= max_joint && joint <= (uint32_t)kMaxJoint && joint * copies <= (uint32_t)kCounterWords;
        plan->nr = n[0];
        plan->nw = n[1];
        plan->copies = copies;
        plan->use = (commit && fits) ? 1u : 0u;
        plan->ticket = 0;
        if (post)
            __hip_atomic_store(post, ((unsigned long long)seq << 32) | ((unsigned long long)n[0] << 16) | n[1], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- rank images --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nmi_rank_kernel(const uint8_t *__restrict__ render_stack, int S,
                                                        const uint8_t *__restrict__ warp_stack, int Wn, int npix, int chunks_per_row,
                                                        int height, int flip, const LevelPlan *__restrict__ plan,
                                                        uint8_t *__restrict__ out_render, uint8_t *__restrict__ out_warp)
{
    __shared__ uint8_t lut[2][kBins];
    if (!plan->use) return;
    const int tid = threadIdx.x;
    lut[0][tid] = plan->rank_r[tid];
    lut[1][tid] = plan->rank_w[tid];
    __syncthreads();
    const int cpi = npix >> 4;  // chunks per image
    const long long total = (long long)(S + Wn) * cpi, stride = (long long)gridDim.x * 256;
    const uint4 *src[4];
    uint4 *dst[4];
    uint4 v[4];
    for (long long g0 = (long long)blockIdx.x * 256 + tid; g0 < total; g0 += 4 * stride) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // four loads in flight per lane; chunks past the end repeat the last one (same value stored twice)
            const long long g = min(g0 + u * stride, total - 1);
            const int image = (int)(g / cpi), c = (int)(g - (long long)image * cpi);
            int sc = c;
            if (image < S && flip) {  // NMI.cu:82: frame row y meets render row H-1-y
                const int y = c / chunks_per_row;
                sc = (height - 1 - y) * chunks_per_row + (c - y * chunks_per_row);
            }
            const uint8_t *in = image < S ? render_stack + (size_t)image * npix : warp_stack + (size_t)(image - S) * npix;
            uint8_t *out = image < S ? out_render + (size_t)image * npix : out_warp + (size_t)(image - S) * npix;
            src[u] = reinterpret_cast<const uint4 *>(in + ((size_t)sc << 4));
            dst[u] = reinterpret_cast<uint4 *>(out + ((size_t)c << 4));
            v[u] = *src[u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long g = min(g0 + u * stride, total - 1);
            const uint8_t *l = lut[g / cpi < S ? 0 : 1];
            const uint32_t d[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
            uint32_t o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                o[q] = (uint32_t)l[d[q] & 0xFFu] | ((uint32_t)l[(d[q] >> 8) & 0xFFu] << 8) | ((uint32_t)l[(d[q] >> 16) & 0xFFu] << 16) |
                       ((uint32_t)l[d[q] >> 24] << 24);
            *dst[u] = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}

// ---- scoring ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int slot_in_round(int b, int grid) { return (grid & 7) == 0 ? (b & 7) * (grid >> 3) + (b >> 3) : b; }

// 16 pixels of one lane: byte address of the counter = rank_r * stride_r + (rank_w << shift_w) + copy * 4.
__device__ __forceinline__ void add_chunk_ranks(char *base, const uint4 &rv, const uint4 &wv, uint32_t stride_r, uint32_t shift_w,
                                                uint32_t copy4)
{
    const uint32_t r[4] = {rv.x, rv.y, rv.z, rv.w};
    const uint32_t w[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t rb = (r[q] >> (8 * j)) & 0xFFu, wb = (w[q] >> (8 * j)) & 0xFFu;
            const uint32_t addr = __umul24(rb, stride_r) + (wb << shift_w) + copy4;
            (void)__hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(base + addr), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

// The three 256-element trees of AddVectorPairwiseKernel (NMI.cu:295-339) side by side in DPP rows 0 (render marginal),
// 1 (frame marginal), 2 (joint row sums), then the score; one wavefront.
__device__ __forceinline__ void final_trees(FewLds &lds, const GridArgs &a, int lane, int p, int w, int s, unsigned long long &prev_key)
{
    const int i = lane & 15, r = lane >> 4;
    float lo[8], hi[8];
    const uint32_t *h = r == 0 ? lds.hist_render : lds.hist_warped;
    uint32_t cl[8], ch[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        cl[k] = r < 2 ? h[i + 16 * k] : 0u;
        ch[k] = r < 2 ? h[i + 16 * k + 128] : 0u;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        lo[k] = a.table[cl[k]];
        hi[k] = a.table[ch[k]];
    }
    if (r == 2) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            lo[k] = lds.joint_row_sums[i + 16 * k];
            hi[k] = lds.joint_row_sums[i + 16 * k + 128];
        }
    }
    const float x = row_tree_16(lane_tree_16(lo, hi));
    const float a1 = __shfl(x, 0, 64), a2 = __shfl(x, 16, 64), a3 = __shfl(x, 32, 64);
    if (lane == 0) commit_score(a, p, w, s, a1, a2, a3, prev_key);
}

}  // namespace

template <bool BG>
__global__ __launch_bounds__(NMI_BLOCK_THREADS) void nmi_fewlevels_kernel(GridArgs a)
{
    __shared__ FewLds lds;
    const LevelPlan *__restrict__ plan = a.plan;
    if (!plan->use) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (blockIdx.x == 0 && tid == 0 && a.reset_key) *a.reset_key = 0ull;  // next launch's slot; idle during this one
    const uint32_t nr = plan->nr, nw = plan->nw, copies = plan->copies;
    const uint32_t joint = nr * nw;
    const uint32_t log_copies = 31u - (uint32_t)__clz((int)copies);
    {
        uint4 *c4 = reinterpret_cast<uint4 *>(lds.cnt);
        const uint4 z = {0, 0, 0, 0};
        for (int i = tid; i < kCounterWords / 4; i += kBlock) c4[i] = z;
    }
    if (tid < kBins) {
        // rank_w: bin -> rank, or kAbsent (the plan's table is by raw intensity and cannot tell rank 0 from "does not occur")
        const uint32_t raw = (uint32_t)tid << a.shift;
        const uint32_t rk = raw < (uint32_t)kBins ? plan->rank_w[raw] : 0u;
        lds.rank_w[tid] = (raw < (uint32_t)kBins && rk < nw && plan->level_w[rk] == (uint8_t)tid) ? (uint16_t)rk : (uint16_t)kAbsent;
        lds.level_r[tid] = (uint32_t)tid < nr ? plan->level_r[tid] : (uint8_t)0;
    }
    // !BG (NMI.cu:85): pixels with render or frame intensity 0 are not counted = row / column of intensity 0 cleared
    // (256 bins only: with fewer, bin 0 also holds intensities the rule keeps; the host then does not come here)
    const bool zero_r0 = !BG && nr > 0 && plan->level_r[0] == 0, zero_w0 = !BG && nw > 0 && plan->level_w[0] == 0;
    __syncthreads();

    // this lane's 16 column ranks in the 256-wide rows of the row phase: d2 = i + 16 k (k < 8) and + 128
    const int i16 = lane & 15, r4 = lane >> 4;
    uint32_t rk_lo[8], rk_hi[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        rk_lo[k] = lds.rank_w[i16 + 16 * k];
        rk_hi[k] = lds.rank_w[i16 + 16 * k + 128];
    }

    const int total = a.S_local * a.Wn;
    const int nchunks = a.npix >> 4, last = nchunks - 1;
    const uint32_t stride_r = (nw << log_copies) << 2, shift_w = log_copies + 2, copy4 = ((uint32_t)lane & (copies - 1u)) << 2;
    const uint32_t lanes_per_bin = copies >> 2;  // uint4s per bin: 2, 4 or 8
    const uint32_t quads = (joint * copies) >> 2;
    char *const base = reinterpret_cast<char *>(lds.cnt);
    unsigned long long prev_key = 0;
    const int slot = slot_in_round(blockIdx.x, gridDim.x);
    for (int ordinal = slot; ordinal < total; ordinal += gridDim.x) {
        const int p = a.order ? a.order[ordinal] : ordinal;
        const int w = p / a.S_local, s = p - w * a.S_local;
        const uint8_t *render = a.render_stack + (size_t)s * a.npix;
        const uint8_t *warped = a.warp_stack + (size_t)w * a.npix;
        // histogram phase (NMI.cu:79-87 on rank images): one LDS add per pixel, next chunk's loads in flight
        {
            int c = min(tid, last);
            uint4 rc = *reinterpret_cast<const uint4 *>(render + ((uint32_t)c << 4));
            uint4 wc = *reinterpret_cast<const uint4 *>(warped + ((uint32_t)c << 4));
            for (int ch = tid; ch < nchunks; ch += kBlock) {
                const int cn = min(ch + kBlock, last);
                const uint4 rn = *reinterpret_cast<const uint4 *>(render + ((uint32_t)cn << 4));
                const uint4 wn = *reinterpret_cast<const uint4 *>(warped + ((uint32_t)cn << 4));
                add_chunk_ranks(base, rc, wc, stride_r, shift_w, copy4);
                rc = rn;
                wc = wn;
            }
        }
        __syncthreads();  // B1: counters complete; wavefront 0 is past the previous candidate's final trees
        // collapse: sum the copies of each bin (consecutive lanes hold consecutive uint4s of a bin), clear the counters
        {
            uint4 *c4 = reinterpret_cast<uint4 *>(lds.cnt);
            for (uint32_t q0 = 0; q0 < quads; q0 += kBlock) {  // workgroup-uniform trip count
                const uint32_t q = q0 + tid;
                uint32_t sum = 0;
                if (q < quads) {
                    const uint4 v = c4[q];
                    c4[q] = make_uint4(0, 0, 0, 0);
                    sum = v.x + v.y + v.z + v.w;
                }
                sum += row_shl<1>(sum);
                if (lanes_per_bin > 2) sum += row_shl<2>(sum);
                if (lanes_per_bin > 4) sum += row_shl<4>(sum);
                if (q < quads && (q & (lanes_per_bin - 1u)) == 0) {
                    const uint32_t bin = q / lanes_per_bin;
                    if (!BG) {
                        const uint32_t ri = bin / nw, wi = bin - ri * nw;
                        if ((zero_r0 && ri == 0) || (zero_w0 && wi == 0)) sum = 0;
                    }
                    lds.joint[bin] = sum;
                }
            }
            if (tid < kBins) {
                lds.hist_render[tid] = 0;
                lds.hist_warped[tid] = 0;
                lds.joint_row_sums[tid] = 0.0f;
            }
        }
        __syncthreads();  // B2
        // row phase (ComputeEntropyKernel + AddvectorParwiseMidKernel, NMI.cu:242-287): a 16-lane DPP row takes one joint
        // row; lane i of it owns the columns i + 16 j, so the tree steps n >= 16 stay in the lane (lane_tree_16)
        {
            uint32_t col_lo[8], col_hi[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) col_lo[k] = col_hi[k] = 0;
            for (uint32_t row0 = (uint32_t)wave * 4; row0 < nr; row0 += kWaves * 4) {  // wavefront-uniform
                const uint32_t ri = row0 + r4;
                const bool live = ri < nr;
                const uint32_t *jr = lds.joint + (live ? ri : 0u) * nw;
                uint32_t cl[8], ch[8], rsum = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    cl[k] = (live && rk_lo[k] != kAbsent) ? jr[rk_lo[k] == kAbsent ? 0u : rk_lo[k]] : 0u;
                    ch[k] = (live && rk_hi[k] != kAbsent) ? jr[rk_hi[k] == kAbsent ? 0u : rk_hi[k]] : 0u;
                    col_lo[k] += cl[k];
                    col_hi[k] += ch[k];
                    rsum += cl[k] + ch[k];
                }
                float tl[8], th[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    tl[k] = a.table[cl[k]];  // table[0] = 0
                    th[k] = a.table[ch[k]];
                }
                rsum = row_sum_16(rsum);
                const float x = row_tree_16(lane_tree_16(tl, th));
                if (live && i16 == 0) {
                    const uint32_t d1 = lds.level_r[ri];
                    lds.hist_render[d1] = rsum;
                    lds.joint_row_sums[d1] = x;
                }
            }
            if ((uint32_t)wave * 4 < nr) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (col_lo[k]) atomicAdd(&lds.hist_warped[i16 + 16 * k], col_lo[k]);
                    if (col_hi[k]) atomicAdd(&lds.hist_warped[i16 + 16 * k + 128], col_hi[k]);
                }
            }
        }
        __syncthreads();  // B3
        if (wave == 0) final_trees(lds, a, lane, p, w, s, prev_key);  // the others start the next candidate's pixels
    }
    if (tid == 0) publish_winner(a, prev_key);
}

hipError_t launch_levels(const uint8_t *render_stack, int S, const uint8_t *warp_stack, int Wn, int npix, int shift, LevelPlan *plan,
                         unsigned long long *post, uint32_t seq, uint32_t max_joint, bool commit, hipStream_t stream)
{
    // measured with 2 / 3 / 4 / 8 slices per image at 640x480: 11.8 / 10.7 / 11.0 / 13.3 us for 54 images, of which 6.9 us
    // are there without the scan (launch, 32 KiB of LDS cleared, merge, ticket, plan, kernel-end cache maintenance)
    const int slices = 4;
    hipLaunchKernelGGL(nmi_levels_kernel, dim3((S + Wn) * slices), dim3(kProbeBlock), 0, stream, render_stack, S, warp_stack, Wn, npix, slices,
                       plan, post, seq, max_joint, commit ? 1 : 0, shift);
    return hipGetLastError();
}

hipError_t launch_fewlevels(const GridArgs &a, uint8_t *rank_renders, uint8_t *rank_warps, int workgroups, bool use_bg,
                            hipStream_t stream)
{
    // rank images: 16 pixels per lane, a few chunks per lane
    const long long chunks = (long long)(a.S_local + a.Wn) * (a.npix >> 4);
    const int rank_wgs = (int)((chunks + 256 * 4 - 1) / (256 * 4));
    hipLaunchKernelGGL(nmi_rank_kernel, dim3(rank_wgs > 0 ? rank_wgs : 1), dim3(256), 0, stream, a.render_stack, a.S_local,
                       a.warp_stack, a.Wn, a.npix, a.chunks_per_row, a.height, a.flip, a.plan, rank_renders, rank_warps);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    GridArgs b = a;
    b.render_stack = rank_renders;
    b.warp_stack = rank_warps;
    if (use_bg)
        hipLaunchKernelGGL((nmi_fewlevels_kernel<true>), dim3(workgroups), dim3(kBlock), 0, stream, b);
    else
        hipLaunchKernelGGL((nmi_fewlevels_kernel<false>), dim3(workgroups), dim3(kBlock), 0, stream, b);
    return hipGetLastError();
}

int fewlevels_max_joint() { return kMaxJoint; }

}  // namespace nmi
