// nmi_split_kernel.hip -- the scoring path for grids with fewer candidates than compute units: K workgroups per candidate.
//
// Why: nmi_grid_kernel gives a candidate to ONE workgroup on one of the 256 CUs.  The reference's own call site evaluates
// one candidate per call (CUDAF::NMIWithCuda_noMask, Thirdparty/CUDA_Functions/kernel.cu:49-114, called from the loop at
// src/Tracking.cc:1879-1902), and the search strategy routinely collapses axes to a single cell (Tracking.cc:2014-2043,
// nmiSearchKernel.cpp:124-141): grids of 1, 3, 9, 27, 81 candidates are the common case, and they leave most of the chip idle.
//
// How: the K workgroups of a candidate split the joint histogram by RENDER INTENSITY (its rows): part j owns rows
// [j * 256/K, (j+1) * 256/K) as 32-bit LDS counters (K = 8: 32 KiB).  Every part streams ALL pixels of the pair (0.6 MB,
// from the XCD's L2 after the first part touched them: the parts of a candidate are placed on one XCD) but issues the
// LDS atomic only for pixels whose render intensity falls into its rows.  Each part therefore holds COMPLETE rows:
//   * no merge of partial histograms (what sank the pixel-split attempt of round 1: device-scope atomics, profiles/NOTES.md; nmi_pix_kernel.hip is the form that works: one owner, write-through units, masks as flags);
//   * 32-bit counters: no wrap bookkeeping, BG-off and reduced-bin contexts take the same path;
//   * the row trees of AddvectorParwiseMidKernel (NMI.cu:270-287) run per row exactly as in nmi_grid_kernel, so the
//     256 row sums and the render marginal are final when a part stores them.
// What crosses workgroups is small: 256/K row sums + 256/K render-marginal counts + 256 partial column sums per part
// (2.3 KB of payload at K = 8), written to a per-candidate slab.  Part 0 of a candidate is its scorer: it collects the
// slab, sums the K column partials (integers: order-free), runs the three 256-element trees of AddVectorPairwiseKernel
// (NMI.cu:290-339) in the reference's order and forms the score.  Results are bit-identical to nmi_grid_kernel's
// (tests/test_gpu_parity.py runs every grid test through both).
//
// Pixel parts (P = 2 or 4, with K = 8, for the smallest grids: one pair = 32 workgroups).  A part's time is its instruction
// stream over ALL pixels (an LDS atomic issues at one per ~6 cycles per CU however few lanes take part; 4800 of them
// per 640x480 pair = 12 us), so for 1..16 candidates the pixels are cut as well: workgroup (row part j, pixel range q)
// counts range q's pixels into rows j.  Pixel range 0 is the row part's owner: the other P - 1 workgroups send it their
// 32 KiB block of counters, it adds them to its own and carries on (decode, slab) with whole rows again, so everything
// downstream is unchanged and results stay bit-identical.
//
// Hand-off form (both stages): data-tagged granules (MI355X_MICROARCH.md, "handoff-1to1": ~1 us per hop; "R2's granule
// needs no ordering at all").  Every 8-byte unit carries its payload AND the launch's epoch and is written by one sc1
// store; the consumer polls the granules it needs with sc1 loads until they show the epoch.  Nothing waits on a
// store, no ticket is drawn: a stage costs one store -> load trip instead of store, wait, atomic, load (the ticket
// form of this kernel's first version: 5 us for the pixel-part merge, 16 us per pair; now 2 us and 11 us).
// The consumers spin, so every workgroup of the launch must be able to run at once: the host launches this kernel only
// with one unit per workgroup and no more workgroups than compute units (one per CU: LDS padded), and every poll loop is
// bounded (2 ms) -- on a timeout the kernel posts its epoch to GridArgs::split_error and the host redoes that launch's
// search with nmi_grid_kernel, keeps the split forms off for a number of calls, then tries them again (nmi_capi.cpp).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nmi_device.h"
#include "nmi_kernels.h"

namespace nmi {

namespace {

constexpr int kBlock = NMI_BLOCK_THREADS;
constexpr int kWaves = kBlock / 64;
constexpr int kBins = 256;
constexpr int kMinLdsBytes = 84 * 1024;  // more than half of the CU's 160 KiB: one workgroup per CU

template <int K>
struct SplitLds {
    static constexpr int kRows = kBins / K;
    static constexpr int kUsed = (kRows * kBins + 4 * kBins + 4) * 4;
    uint32_t joint[kRows * kBins];  // [row = render intensity - part * kRows][frame intensity]; bank = frame intensity & 31
    uint32_t hist_warped[kBins];    // column sums over this part's rows
    uint32_t fin_render[kBins];     // the last part's copy of the candidate's slab: render marginal,
    uint32_t fin_warped[kBins];     //   frame marginal (sum of the K column partials),
    float fin_rows[kBins];          //   joint row sums
    uint32_t pad0[4];
    uint32_t pad[kUsed < kMinLdsBytes ? (kMinLdsBytes - kUsed) / 4 : 4];
};

template <typename T>
__device__ __forceinline__ void store_sc1(T *p, T v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t load_sc1(const uint32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float load_sc1(const float *p)
{
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

__device__ __forceinline__ void store_granule(unsigned long long *p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_store_dwordx2 sc1
}
__device__ __forceinline__ unsigned long long load_granule(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_load_dwordx2 sc1
}
__device__ __forceinline__ unsigned long long granule32(uint32_t payload, uint32_t epoch) { return (unsigned long long)payload | ((unsigned long long)epoch << 32); }
// Every poll loop is bounded in real time: a legitimate hand-off arrives within ~5 us; after kTimeoutTicks of the 100 MHz
// wall clock (2 ms) the waiter gives up.  (The clock is read every 16th unsuccessful poll only; kMaxPolls is the backstop
// should the clock misbehave.)  The timeout is reported per launch: the launch's epoch goes into word (epoch % kSplitRing) of the
// pinned error ring, so the host can tell WHICH of several launches in flight failed (nmi_capi.cpp, split_launch_failed).
constexpr unsigned long long kTimeoutTicks = 200000ull;
constexpr int kMaxPolls = 1 << 15;
struct PollGuard {
    unsigned long long t0 = 0;
    int tries = 0;
    __device__ __forceinline__ bool expired()
    {
        if (++tries >= kMaxPolls) return true;
        if ((tries & 15) != 1) return false;
        const unsigned long long now = wall_clock64();
        if (tries == 1) {
            t0 = now;
            return false;
        }
        return now - t0 > kTimeoutTicks;
    }
};
__device__ __forceinline__ void raise_timeout(const GridArgs &a)
{
    if (a.split_error) __hip_atomic_store(a.split_error + (a.epoch % kSplitRing), a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// 16 pixels of one lane, hot case (BG on, 256 bins).  xorpat holds (part * kRows) in every byte: after r ^= xorpat a pixel
// belongs to this part iff its render byte is < kRows, and that byte is its local row.  One v_perm_b32 then builds the
// 16-bit counter indices (row << 8 | frame intensity) of TWO pixels; index * 4 is the LDS byte address (the joint array
// sits at LDS address 0: it is the first member of the kernel's only __shared__ object; checked at kernel start), and
// "address inside the joint array" is the ownership test.  A wavefront retires one instruction per ~5 cycles whatever
// the other wavefronts do (tools/ubench/valu_rate.hip), so the loop is bound by its instruction count: the 23
// instructions per 4 pixels are spelled out (hipcc's version of the same C++ had 39: duplicated xors, moves, nops and
// waits between the predicated adds).  Per pixel: 1/4 xor + 1/2 perm + 1 SDWA shift + compare, then EXEC <- mask + ds_add.
template <int K>
__device__ __forceinline__ void add_chunk_split(const uint4 &rv, const uint4 &wv, uint32_t xorpat)
{
    constexpr uint32_t kLimit = (uint32_t)(kBins / K) * kBins * 4u;
    const uint32_t r[4] = {rv.x, rv.y, rv.z, rv.w};
    const uint32_t w[4] = {wv.x, wv.y, wv.z, wv.w};
    const uint32_t two = 2, one = 1, sel01 = 0x05010400u, sel23 = 0x07030602u, limit = kLimit;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t t0, t1, t2, t3;
        unsigned long long m0, m1, m2, m3, saved;
        // v_perm_b32: selector bytes 0-3 pick from the second operand (frame), 4-7 from the first (render):
        // t0 = (r0 << 8 | w0) | (r1 << 8 | w1) << 16, t1 the same for pixels 2, 3.
        // The four ownership masks are formed first (VALU -> SGPR pairs, back to back), then EXEC is loaded with each in
        // turn for its ds_add: one pixel at a time through compare -> s_and_saveexec -> ds_add -> restore costs ~90
        // cycles of the wavefront per pixel in VALU <-> SALU <-> EXEC round trips (measured: 13 us per 640x480 pair).
        asm volatile(
            "v_xor_b32 %0, %10, %9\n\t"
            "v_perm_b32 %1, %0, %11, %13\n\t"
            "v_perm_b32 %0, %0, %11, %12\n\t"
            "v_lshlrev_b32_sdwa %2, %14, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
            "v_lshlrev_b32_sdwa %3, %14, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
            "v_lshlrev_b32_sdwa %1, %14, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
            "v_lshlrev_b32_sdwa %0, %14, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
            "s_mov_b64 %8, exec\n\t"
            "v_cmp_gt_u32_e64 %6, %16, %2\n\t"
            "v_cmp_gt_u32_e64 %7, %16, %3\n\t"
            "v_cmp_gt_u32_e64 %5, %16, %1\n\t"
            "v_cmp_gt_u32_e64 %4, %16, %0\n\t"
            "s_nop 0\n\t"
            "s_mov_b64 exec, %6\n\t"
            "ds_add_u32 %2, %15\n\t"
            "s_mov_b64 exec, %7\n\t"
            "ds_add_u32 %3, %15\n\t"
            "s_mov_b64 exec, %5\n\t"
            "ds_add_u32 %1, %15\n\t"
            "s_mov_b64 exec, %4\n\t"
            "ds_add_u32 %0, %15\n\t"
            "s_mov_b64 exec, %8"
            : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(saved)
            : "s"(xorpat), "v"(r[q]), "v"(w[q]), "s"(sel01), "s"(sel23), "v"(two), "v"(one), "s"(limit)
            : "memory");
    }
}

// One pixel, any switch setting (NMI.cu:85 background rule on the raw intensities, then intensity >> shift).
template <int K>
__device__ __forceinline__ void add_pixel_split(uint32_t *joint, uint32_t d1, uint32_t d2, int part, bool use_bg, int shift, uint32_t weight = 1)
{
    constexpr uint32_t kRows = kBins / K;
    if (!use_bg && (d1 == 0 || d2 == 0)) return;
    d1 >>= shift;
    d2 >>= shift;
    if (d1 / kRows != (uint32_t)part) return;
    (void)__hip_atomic_fetch_add(&joint[(d1 % kRows) * kBins + d2], weight, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// A chunk in the careful loop of the hot path: folded when every active lane's 16 pixels carry one (render, frame) pair
// (see fold_flat_chunk in nmi_kernels.hip for why: 64 lanes adding to one LDS address serialise), added plainly otherwise.
template <int K>
__device__ __forceinline__ void add_chunk_careful(uint32_t *joint, const uint4 &rv, const uint4 &wv, uint32_t xorpat, int part)
{
    const uint32_t rb = rv.x & 0xFFu, wb = wv.x & 0xFFu;
    const bool flat = rv.x == rv.y && rv.y == rv.z && rv.z == rv.w && wv.x == wv.y && wv.y == wv.z && wv.z == wv.w &&
                      rv.x == rb * 0x01010101u && wv.x == wb * 0x01010101u;
    if (!__all(flat)) {
        add_chunk_split<K>(rv, wv, xorpat);
        return;
    }
    const uint32_t key = (rb << 8) | wb;
    const uint32_t key0 = __builtin_amdgcn_readfirstlane(key);
    uint32_t weight = 16;
    bool issue = true;
    if (__all(key == key0)) {  // one lane speaks for the wavefront
        const unsigned long long active = __ballot(1);
        weight = 16u * (uint32_t)__popcll(active);
        issue = __lane_id() == (uint32_t)__ffsll((long long)active) - 1u;
    }
    if (issue) add_pixel_split<K>(joint, rb, wb, part, true, 0, weight);
}

// 16-byte chunk c of the frame / of the render row that meets it (NMI.cu:82: row y of the frame meets row H-1-y of a
// bottom-up render; see histogram_phase in nmi_kernels.hip).  Indices are clamped to the last chunk: loads are unconditional.
__device__ __forceinline__ uint4 load_frame_chunk(const uint8_t *__restrict__ warped, int c, int last)
{
    return *reinterpret_cast<const uint4 *>(warped + ((uint32_t)min(c, last) << 4));
}
__device__ __forceinline__ uint4 load_render_chunk(const GridArgs &a, const uint8_t *__restrict__ render, int c, int last)
{
    c = min(c, last);
    const int y = (int)__umulhi((uint32_t)c, a.cpr_magic);
    return *reinterpret_cast<const uint4 *>(render + ((uint32_t)(__mul24(y, a.flip_row) + c + a.flip_base) << 4));
}
__device__ __forceinline__ void split_range(int all_chunks, int pix_part, int pix_parts, int &first, int &end)
{
    const int per_part = (all_chunks + pix_parts - 1) / pix_parts;
    first = min(pix_part * per_part, all_chunks);
    end = min(first + per_part, all_chunks);
}

// Histogram phase of one part: histogram256Kernel's pixel loop (NMI.cu:79-87) over ALL pixels of the pair, counting
// only the pixels whose render intensity belongs to this part's rows.
template <int K, bool FAST>
__device__ __forceinline__ void histogram_split(uint32_t *joint, const GridArgs &a, const uint8_t *__restrict__ render,
                                                const uint8_t *__restrict__ warped, int tid, int part, bool use_bg, int pix_part,
                                                int pix_parts)
{
    constexpr int NT = kBlock;
    constexpr uint32_t kRows = kBins / K;
    if (a.vec_ok) {
        const int all_chunks = a.npix >> 4;
        const int last = all_chunks - 1;
        // this workgroup's pixel range, in 16-pixel chunks: [first, nchunks)
        int first, nchunks;
        split_range(all_chunks, pix_part, pix_parts, first, nchunks);
        auto ldw = [&](int c) { return load_frame_chunk(warped, c, last); };
        auto ldr = [&](int c) { return load_render_chunk(a, render, c, last); };
        if (FAST) {
            // Software pipeline with two named register sets of TWO chunk pairs each (64 B per lane and set): while one
            // set's 32 pixels are added the other set's four loads are in flight -- a part has to pull the whole pair
            // through one CU in a few microseconds, and one set of adds (~1400 cycles per SIMD) covers an L2 round trip.
            // Loads are unconditional (index clamped), only the adds are predicated.  A flat hint (first dword == last
            // dword in both images for the whole wavefront) sends the wavefront to the careful loop for the rest of the
            // candidate.
            const uint32_t xorpat = (uint32_t)part * kRows * 0x01010101u;
            const int iters = (nchunks - first + 2 * NT - 1) / (2 * NT);  // workgroup-uniform
            int resume = -1;
            int ch = first + tid;
            uint4 wa0 = ldw(ch), ra0 = ldr(ch), wa1 = ldw(ch + NT), ra1 = ldr(ch + NT), wb0, rb0, wb1, rb1;
            const bool fold = !(a.phase_mask & 4);
#define NMI_SPLIT_STEP(RC, WC, OFF)                                          \
    if (fold && __builtin_expect(flat_hint(RC, WC), 0)) {                    \
        resume = ch + (OFF) * NT;                                            \
        break;                                                               \
    }                                                                        \
    if (ch + (OFF) * NT < nchunks) add_chunk_split<K>(RC, WC, xorpat);
            for (int it = 0; it < iters; it += 2) {
                wb0 = ldw(ch + 2 * NT);
                rb0 = ldr(ch + 2 * NT);
                wb1 = ldw(ch + 3 * NT);
                rb1 = ldr(ch + 3 * NT);
                NMI_SPLIT_STEP(ra0, wa0, 0)
                NMI_SPLIT_STEP(ra1, wa1, 1)
                wa0 = ldw(ch + 4 * NT);
                ra0 = ldr(ch + 4 * NT);
                wa1 = ldw(ch + 5 * NT);
                ra1 = ldr(ch + 5 * NT);
                NMI_SPLIT_STEP(rb0, wb0, 2)
                NMI_SPLIT_STEP(rb1, wb1, 3)
                ch += 4 * NT;
            }
#undef NMI_SPLIT_STEP
            if (resume >= 0) {
                uint4 wc = ldw(resume), rc = ldr(resume);
#pragma unroll 1
                for (int c = resume; c < nchunks; c += NT) {
                    const uint4 wn = ldw(c + NT), rn = ldr(c + NT);
                    add_chunk_careful<K>(joint, rc, wc, xorpat, part);
                    wc = wn;
                    rc = rn;
                }
            }
        } else {
            // BG off and / or reduced bins: per-pixel form on 16-byte loads, one chunk of prefetch
            uint4 wc = ldw(first + tid), rc = ldr(first + tid);
#pragma unroll 1
            for (int c = first + tid; c < nchunks; c += NT) {
                const uint4 wn = ldw(c + NT), rn = ldr(c + NT);
                const uint32_t r[4] = {rc.x, rc.y, rc.z, rc.w}, w[4] = {wc.x, wc.y, wc.z, wc.w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        add_pixel_split<K>(joint, (r[q] >> (8 * j)) & 0xFFu, (w[q] >> (8 * j)) & 0xFFu, part, use_bg, a.shift);
                wc = wn;
                rc = rn;
            }
        }
    } else {
        // Any width / alignment: byte loads, position arithmetic as written in NMI.cu:79-83.
        const int per_part = (a.npix + pix_parts - 1) / pix_parts;
        const int pos0 = min(pix_part * per_part, a.npix), pos1 = min(pos0 + per_part, a.npix);
        for (int pos = pos0 + tid; pos < pos1; pos += NT) {
            const int y = pos / a.width;
            const int x = pos - y * a.width;
            const int ry = a.flip ? (a.height - 1 - y) : y;
            add_pixel_split<K>(joint, render[ry * a.width + x], warped[pos], part, use_bg, a.shift);
        }
    }
}

// A pixel range short enough for a lane to hold ALL its chunks at once (at most kShortChunks pairs of 16 bytes: one pair of a
// 640x480 frame cut into 4 ranges is 4.7 chunks per lane): the loads are issued together, before the workgroup has even
// cleared its counters, and their latency -- the histogram phase of such a range is three dependent L2 / HBM round trips
// in the pipelined loop -- is paid once, under the clear and its barrier.
constexpr int kShortChunks = 5;
struct ShortRange {
    uint4 r[kShortChunks], w[kShortChunks];
    int first, nchunks;
    bool on;
};

__device__ __forceinline__ void load_short(ShortRange &sr, const GridArgs &a, const uint8_t *__restrict__ render, const uint8_t *__restrict__ warped,
                                           int tid, int pix_part, int pix_parts)
{
    const int all_chunks = a.npix >> 4, last = all_chunks - 1;
    split_range(all_chunks, pix_part, pix_parts, sr.first, sr.nchunks);
#pragma unroll
    for (int k = 0; k < kShortChunks; ++k) {
        const int c = sr.first + tid + k * kBlock;
        sr.w[k] = load_frame_chunk(warped, c, last);
        sr.r[k] = load_render_chunk(a, render, c, last);
    }
}

template <int K>
__device__ __forceinline__ void add_short(uint32_t *joint, const ShortRange &sr, const GridArgs &a, int tid, int part)
{
    constexpr uint32_t kRows = kBins / K;
    const uint32_t xorpat = (uint32_t)part * kRows * 0x01010101u;
    const bool fold = !(a.phase_mask & 4);
#pragma unroll
    for (int k = 0; k < kShortChunks; ++k) {
        if (sr.first + tid + k * kBlock >= sr.nchunks) continue;
        if (fold && __builtin_expect(flat_hint(sr.r[k], sr.w[k]), 0))
            add_chunk_careful<K>(joint, sr.r[k], sr.w[k], xorpat, part);
        else
            add_chunk_split<K>(sr.r[k], sr.w[k], xorpat);
    }
}

// Decode of this part's rows: counters -> per-bin terms (ComputeEntropyKernel, NMI.cu:242-263, through the per-count
// table) -> row trees (AddvectorParwiseMidKernel, NMI.cu:270-287).  Same lane / word ownership as decode_phase in
// nmi_kernels.hip: a wavefront takes 4 rows per pass, one per 16-lane DPP row; lane i of a row owns the bins
// d2 = i + 16*j, so the tree steps n = 128..16 are in-lane adds and n = 8..1 DPP shifts.  Odd DPP rows start one group
// later so that the two rows of a 32-lane LDS access group hit disjoint banks.  Counters are cleared as they are read.
template <int K>
__device__ __forceinline__ void decode_split(SplitLds<K> &lds, const GridArgs &a, SplitSlab *slab, int part, int wave, int lane)
{
    constexpr int kRows = kBins / K;
    const int i = lane & 15, r = lane >> 4, o = r & 1;
    uint32_t col_lo[8], col_hi[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) col_lo[k] = col_hi[k] = 0;
    bool any = false;
#pragma unroll 1
    for (int row0 = wave * 4; row0 < kRows; row0 += kWaves * 4) {
        any = true;
        const int row = row0 + r, d1 = part * kRows + row;
        uint32_t *jr = lds.joint + row * kBins;
        uint32_t lo[8], hi[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int q = i + 16 * ((k + o) & 7);
            lo[k] = jr[q];
            hi[k] = jr[q + 128];
            jr[q] = 0;  // ready for the next candidate
            jr[q + 128] = 0;
        }
        float tl[8], th[8];
        uint32_t rsum = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            col_lo[k] += lo[k];
            col_hi[k] += hi[k];
            rsum += lo[k] + hi[k];
            tl[k] = a.table[lo[k]];  // table[0] = 0: no branch for empty bins (most of them; one cached line)
            th[k] = a.table[hi[k]];
        }
        rsum = row_sum_16(rsum);
        const float x = row_tree_16(lane_tree_16(tl, th));
        if (i == 0) {
            store_granule(&slab->row_sums[d1], granule32(__float_as_uint(x), a.epoch));
            store_granule(&slab->hist_render[d1], granule32(rsum, a.epoch));
        }
        if (a.dbg_joint) {
            uint32_t *out = a.dbg_joint + d1 * kBins;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int q = i + 16 * ((k + o) & 7);
                out[q] = lo[k];
                out[q + 128] = hi[k];
            }
        }
    }
    if (any) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int q = i + 16 * ((k + o) & 7);
            if (col_lo[k]) atomicAdd(&lds.hist_warped[q], col_lo[k]);
            if (col_hi[k]) atomicAdd(&lds.hist_warped[q + 128], col_hi[k]);
        }
    }
}

// The scoring part (part 0).  All wavefronts first collect the candidate's slab -- lane t polls the K column partials of
// bin t and sums them, others take the render marginal and the row sums -- into LDS; then wavefront 0 runs the three 256-element trees of
// AddVectorPairwiseKernel (NMI.cu:295-339) side by side in DPP rows 0 (render marginal), 1 (frame marginal), 2 (joint
// row sums), and the score.  (Fetched by wavefront 0 alone, 16 + 16K dependent loads per lane, this took 12 us.)
template <int K>
__device__ __forceinline__ void gather_slab(SplitLds<K> &lds, const GridArgs &a, const SplitSlab *slab, int tid)
{
    if (tid < kBins) {
        uint32_t c = 0;
        unsigned long long v[K];
        PollGuard guard;
        bool ok;
        do {
            ok = true;
#pragma unroll
            for (int j = 0; j < K; ++j) v[j] = load_granule(&slab->hw_part[j][tid]);
#pragma unroll
            for (int j = 0; j < K; ++j) ok = ok && (uint32_t)(v[j] >> 32) == a.epoch;
            if (!ok) __builtin_amdgcn_s_sleep(8);
        } while (!ok && !guard.expired());
        if (!ok) raise_timeout(a);
#pragma unroll
        for (int j = 0; j < K; ++j) c += (uint32_t)v[j];
        lds.fin_warped[tid] = c;
    } else if (tid < 3 * kBins) {
        const unsigned long long *src = tid < 2 * kBins ? &slab->hist_render[tid - kBins] : &slab->row_sums[tid - 2 * kBins];
        unsigned long long v;
        PollGuard guard;
        while ((uint32_t)((v = load_granule(src)) >> 32) != a.epoch && !guard.expired()) __builtin_amdgcn_s_sleep(8);
        if ((uint32_t)(v >> 32) != a.epoch) raise_timeout(a);
        if (tid < 2 * kBins)
            lds.fin_render[tid - kBins] = (uint32_t)v;
        else
            lds.fin_rows[tid - 2 * kBins] = __uint_as_float((uint32_t)v);
    }
}

template <int K>
__device__ __forceinline__ void final_split(const SplitLds<K> &lds, const GridArgs &a, int lane, int p, int w, int s,
                                            unsigned long long &prev_key)
{
    const int i = lane & 15, r = lane >> 4;
    const uint32_t *h = r == 0 ? lds.fin_render : lds.fin_warped;
    float lo[8], hi[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        lo[k] = a.table[r < 2 ? h[i + 16 * k] : 0u];  // rows 2, 3 fetch table[0] = 0
        hi[k] = a.table[r < 2 ? h[i + 16 * k + 128] : 0u];
    }
    if (r == 2) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            lo[k] = lds.fin_rows[i + 16 * k];
            hi[k] = lds.fin_rows[i + 16 * k + 128];
        }
    }
    const float x = row_tree_16(lane_tree_16(lo, hi));
    const float a1 = __shfl(x, 0, 64), a2 = __shfl(x, 16, 64), a3 = __shfl(x, 32, 64);
    if (a.dbg_h1 || a.dbg_h2) {
        for (int t = lane; t < kBins; t += 64) {
            if (a.dbg_h1) a.dbg_h1[t] = lds.fin_render[t];
            if (a.dbg_h2) a.dbg_h2[t] = lds.fin_warped[t];
        }
    }
    if (lane == 0) commit_score(a, p, w, s, a1, a2, a3, prev_key);
}

}  // namespace

// Unit u of a launch = (candidate, row part, pixel range).  Workgroups are dealt to the 8 XCDs round-robin (blocks b and
// b + 8 share one), and hand-offs inside one XCD are the fast ones.
//   8 or more candidates: taken in groups of 8, one per XCD; all workgroups of a candidate are blocks b, b + 8, ... of
//     that XCD, so they read the pair from one L2 and both hand-off stages stay inside the XCD (the grid is padded to a
//     multiple of 8 candidates; surplus workgroups exit).
//   fewer than 8 candidates with 8 row parts: the grid is exactly candidates x 8 x P workgroups and ROW PART j lives on
//     XCD j: the pixel ranges of a row part (the 32 KiB merges) share an XCD, only the small slab crosses XCDs, and
//     every XCD's L2 serves a share of the pixel stream.  (One pair: 32 workgroups instead of 256 dispatched, 1.2 us
//     less per blocking call; the same 32 workgroups dealt without regard to XCDs finish 1-6 us later.)
// Placement is a speed matter only.
__host__ __device__ inline bool split_exact_grid(int total, int parts) { return total < 8 && parts == 8; }  // (row part j on XCD j needs 8 of them)
__device__ __forceinline__ void split_unit(int u, int total, int parts, int pix_parts, int &cand, int &part, int &pix_part)
{
    if (split_exact_grid(total, parts)) {
        part = u & 7;
        const int t = u >> 3;
        cand = t % total;
        pix_part = t / total;
    } else {
        const int x = u & 7, t = u >> 3, sub = t % (parts * pix_parts);
        cand = (t / (parts * pix_parts)) * 8 + x;
        part = sub % parts;
        pix_part = sub / parts;
    }
}

// Merge of the pixel parts of one row part (P > 1).  Pixel ranges 1..P-1 send their counters to range 0 as granules of
// two 24-bit counts + the 16-bit epoch tag (counts stay below 2^24: the host uses pixel parts only for frames of fewer
// than 2^24 pixels) and are done; range 0 polls for them, adds them to its LDS counters and returns true.
template <int K, int P>
__device__ __forceinline__ bool merge_pixel_parts(SplitLds<K> &lds, const GridArgs &a, int cand, int part, int pix_part, int tid)
{
    constexpr int kRows = kBins / K;
    constexpr int kPairs = kRows * kBins / 2;   // granules per block: two counters each
    constexpr int kPer = kPairs / kBlock;       // per lane
    static_assert(kPairs % kBlock == 0, "a block is a whole number of granules per lane");
    const unsigned long long tag = (unsigned long long)(a.epoch & 0xFFFFu) << 48;
    unsigned long long *const j2 = reinterpret_cast<unsigned long long *>(lds.joint);
    unsigned long long *const base = a.blocks + (size_t)(cand * K + part) * P * kPairs;
    if (pix_part != 0) {
        unsigned long long *const mine = base + (size_t)pix_part * kPairs;
#pragma unroll
        for (int e = 0; e < kPer; ++e) {
            const int i = tid + e * kBlock;
            const unsigned long long w = j2[i];
            store_granule(&mine[i], (w & 0xFFFFFFull) | ((w >> 32) << 24) | tag);
            j2[i] = 0ull;
        }
        return false;
    }
    unsigned long long v[kPer][P - 1];
    PollGuard guard;
    bool ok;
    do {
        ok = true;
#pragma unroll
        for (int e = 0; e < kPer; ++e)
#pragma unroll
            for (int q = 1; q < P; ++q) v[e][q - 1] = load_granule(&base[(size_t)q * kPairs + tid + e * kBlock]);
#pragma unroll
        for (int e = 0; e < kPer; ++e)
#pragma unroll
            for (int q = 1; q < P; ++q) ok = ok && (v[e][q - 1] >> 48) == (tag >> 48);
        if (!ok) __builtin_amdgcn_s_sleep(2);
    } while (!ok && !guard.expired());
    if (!ok) raise_timeout(a);
#pragma unroll
    for (int e = 0; e < kPer; ++e) {
        const int i = tid + e * kBlock;
        unsigned long long acc = j2[i];
        uint32_t lo = (uint32_t)acc, hi = (uint32_t)(acc >> 32);
#pragma unroll
        for (int q = 1; q < P; ++q) {
            lo += (uint32_t)(v[e][q - 1] & 0xFFFFFFull);
            hi += (uint32_t)((v[e][q - 1] >> 24) & 0xFFFFFFull);
        }
        j2[i] = (unsigned long long)lo | ((unsigned long long)hi << 32);
    }
    __syncthreads();  // the merged counters are complete before decode
    return true;
}

// Kernel arguments are read afresh in every phase (fresh(): the pointer to the kernel-argument segment goes through an
// empty asm so that the loads of one phase cannot be merged with those of another).  By-value arguments are otherwise
// loaded once and kept in scalar registers for the whole kernel; this kernel then spilled 60 of them to VGPR lanes, and
// two of their stack slots survived in the kernel descriptor as 36 bytes of private segment that no instruction touches.
// Now: no spills, no private segment.  (What a launch pays for merely asking for scratch is small -- 0.5-0.7 us of the
// 8.2 us a blocking launch of 32 empty workgroups takes, tools/ubench/launch_latency.hip; an nmi_eval_pair call went
// from 21.0 to 20.1-20.3 us.)
template <int K, int P, bool FAST>
__global__ __launch_bounds__(NMI_BLOCK_THREADS) void nmi_split_kernel(GridArgs a_by_value, int use_bg)
{
    __shared__ SplitLds<K> lds;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    typedef const GridArgs __attribute__((address_space(4))) *kernargs_t;
    auto fresh = [&]() {
        kernargs_t ka = (kernargs_t)__builtin_amdgcn_kernarg_segment_ptr();  // GridArgs is the first argument
        asm volatile("" : "+s"(ka));
        return *(const GridArgs *)ka;  // (the compiler sees through the cast: scalar loads from the constant address space)
    };
    (void)a_by_value;

    auto stamp = [&](int k) {  // tools/small_grid_time.py --stamps: where a part's time goes (100 MHz clock)
        const GridArgs a = fresh();
        if (a.dbg_stamps && tid == 0) a.dbg_stamps[blockIdx.x * 8 + k] = wall_clock64();
    };
    stamp(0);
    long long clk0 = 0;
    int total, phase_mask;
    {
        const GridArgs a = fresh();
        clk0 = a.dbg_stamps ? clock64() : 0;  // shader-clock counter: with the wall-clock stamps it gives the clock held
        if (blockIdx.x == 0 && tid == 0 && a.reset_key) *a.reset_key = 0ull;  // next launch's slot; idle during this one
        total = a.S_local * a.Wn;
        phase_mask = a.phase_mask;
    }
    ShortRange sr;
    sr.on = false;
    int short_part = 0;
    if (FAST && P > 1 && (phase_mask & 1)) {  // (pixel ranges: the one-pair and few-pair launches, where latency is the whole cost)
        const GridArgs a = fresh();
        const int units0 = split_workgroups(total, K, P);
        int p0, part0, pix0;
        split_unit((int)blockIdx.x, total, K, P, p0, part0, pix0);
        const int per_part = ((a.npix >> 4) + P - 1) / P;
        if (a.vec_ok && (int)blockIdx.x < units0 && p0 < total && per_part <= kShortChunks * kBlock) {
            const int w0 = p0 / a.S_local, s0 = p0 - w0 * a.S_local;
            load_short(sr, a, a.pair_renders ? a.pair_renders[p0] : a.render_stack + (size_t)s0 * a.npix,
                       a.pair_warps ? a.pair_warps[p0] : a.warp_stack + (size_t)w0 * a.npix, tid, pix0, P);
            sr.on = true;
            short_part = part0;
        }
    }
    {
        uint4 *j4 = reinterpret_cast<uint4 *>(lds.joint);
        const uint4 z = {0, 0, 0, 0};
        for (int k = tid; k < SplitLds<K>::kRows * kBins / 4; k += kBlock) j4[k] = z;
    }
    if (tid < kBins) lds.hist_warped[tid] = 0;
    // add_chunk_split addresses the counters from LDS address 0
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds.joint != 0u) __builtin_trap();
    __syncthreads();
    const bool hist_done = sr.on;
    if (sr.on) {  // this workgroup's first unit: its chunks are already in registers (and die here)
        const GridArgs a = fresh();
        add_short<K>(lds.joint, sr, a, tid, short_part);
    }

    const int units = split_workgroups(total, K, P);
    unsigned long long prev_key = 0;
    for (int u = blockIdx.x; u < units; u += gridDim.x) {
        int p, part, pix_part;
        split_unit(u, total, K, P, p, part, pix_part);
        if (p >= total) continue;  // workgroup-uniform

        stamp(1);
        if ((phase_mask & 1) && !(hist_done && u == (int)blockIdx.x)) {
            const GridArgs a = fresh();
            const int w = p / a.S_local, s = p - w * a.S_local;
            histogram_split<K, FAST>(lds.joint, a, a.pair_renders ? a.pair_renders[p] : a.render_stack + (size_t)s * a.npix,
                                     a.pair_warps ? a.pair_warps[p] : a.warp_stack + (size_t)w * a.npix, tid, part, use_bg != 0, pix_part, P);
        }
        __syncthreads();
        stamp(2);
        if (P > 1) {
            const GridArgs a = fresh();
            if (!merge_pixel_parts<K, P>(lds, a, p, part, pix_part, tid)) continue;  // workgroup-uniform
        }
        {
            const GridArgs a = fresh();
            decode_split<K>(lds, a, a.slabs + p, part, wave, lane);
        }
        __syncthreads();
        stamp(3);
        {
            const GridArgs a = fresh();
            SplitSlab *slab = a.slabs + p;
            if (tid < kBins) {
                // (phase mask bit 9, tests only: part 1 keeps its column sums to itself, so the scorer's wait must time out)
                if (!((phase_mask & 512) && part == 1)) store_granule(&slab->hw_part[part][tid], granule32(lds.hist_warped[tid], a.epoch));
                lds.hist_warped[tid] = 0;
            }
            if (part != 0) continue;  // workgroup-uniform: only part 0 scores the candidate
            gather_slab<K>(lds, a, slab, tid);
        }
        __syncthreads();
        stamp(4);
        if (wave == 0) {
            const GridArgs a = fresh();
            const int w = p / a.S_local, s = p - w * a.S_local;
            final_split<K>(lds, a, lane, p, w, s, prev_key);
            stamp(5);
        }
    }
    if (tid == 0 && !(phase_mask & 16)) {
        const GridArgs a = fresh();
        publish_winner(a, prev_key);
    }
    stamp(6);
    {
        const GridArgs a = fresh();
        if (a.dbg_stamps && tid == 0) a.dbg_stamps[blockIdx.x * 8 + 7] = (unsigned long long)(clock64() - clk0);
    }
}

__host__ __device__ int split_workgroups(int candidates, int parts, int pix_parts)
{
    return (split_exact_grid(candidates, parts) ? candidates : ((candidates + 7) / 8) * 8) * parts * pix_parts;
}

template <int K, int P>
static void launch_split_k(const GridArgs &a, dim3 grid, dim3 block, bool use_bg, hipStream_t stream)
{
    const bool fast = use_bg && a.shift == 0;
    if (fast)
        hipLaunchKernelGGL((nmi_split_kernel<K, P, true>), grid, block, 0, stream, a, 1);
    else
        hipLaunchKernelGGL((nmi_split_kernel<K, P, false>), grid, block, 0, stream, a, use_bg ? 1 : 0);
}

hipError_t launch_split(const GridArgs &a, int parts, int pix_parts, int workgroups, bool use_bg, hipStream_t stream)
{
    if (!a.slabs || workgroups <= 0 || (workgroups & 7)) return hipErrorInvalidValue;
    if (pix_parts > 1 && !a.blocks) return hipErrorInvalidValue;
    dim3 grid(workgroups), block(kBlock);
    if (pix_parts == 4 && parts == 8)
        launch_split_k<8, 4>(a, grid, block, use_bg, stream);
    else if (pix_parts == 2 && parts == 8)
        launch_split_k<8, 2>(a, grid, block, use_bg, stream);
    else if (pix_parts == 2 && parts == 4)
        launch_split_k<4, 2>(a, grid, block, use_bg, stream);
    else if (pix_parts != 1)
        return hipErrorInvalidValue;
    else
        switch (parts) {
        case 2: launch_split_k<2, 1>(a, grid, block, use_bg, stream); break;
        case 4: launch_split_k<4, 1>(a, grid, block, use_bg, stream); break;
        case 8: launch_split_k<8, 1>(a, grid, block, use_bg, stream); break;
        default: return hipErrorInvalidValue;
        }
    return hipGetLastError();
}

}  // namespace nmi
