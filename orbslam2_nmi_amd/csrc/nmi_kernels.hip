// nmi_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the NMI pose-candidate scoring path.
//
// What is computed is fixed by the reference (gsanya/orbslam2_NMI, paths relative to its root):
//   Thirdparty/CUDA_Functions/NMI.cu:42-104   joint + marginal 256-bin histograms of (render, warped frame)
//   Thirdparty/CUDA_Functions/NMI.cu:230-267  per-bin term  (c/len) * log2f(c/len), len = W*H
//   Thirdparty/CUDA_Functions/NMI.cu:270-339  stride-halving fp32 trees (joint rows first, then row sums)
//   Thirdparty/CUDA_Functions/NMI.cu:342-362  SUC / ENMI score with the all-zero guard
//   Thirdparty/Localization/helperFunctions.cpp:50-103  arg-max (strict '>' from 0, lowest index on ties)
// How it is computed is new: one 1024-lane workgroup per pose candidate owns the whole
// 256x256 joint histogram in LDS as packed 16-bit counters (128 KiB of the CU's 160 KiB); counts above 65535
// are caught by a pixel-count test and redone with exact wrap bookkeeping; the entropy terms come from a
// per-context table indexed by count; the trees are evaluated in registers / DPP in the reference's order; the
// score, the rating-table store and the arg-max (one 64-bit atomicMax per candidate) are fused into the
// same launch.  Also here: the warp-stack and point-cloud render-stack producers.
// Histogramming is integer scatter work: no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nmi_kernels.h"
#include "nmi_device.h"

namespace nmi {

namespace {

constexpr int kBlock = NMI_BLOCK_THREADS;  // 1024 lanes = 16 wavefronts, one workgroup per CU (LDS-limited)
constexpr int kWaves = kBlock / 64;
constexpr int kBins = 256;
constexpr int kWords = kBins * kBins / 2;  // two 16-bit counters per LDS word
constexpr int kOvfCap = 1024;              // >= 2 * floor(2^24 / 65536) + 1 wrap events per candidate
constexpr int kRowsPerWave = kBins / kWaves;
constexpr int kLdsTable = 4096;            // per-count entropy terms kept in LDS for counts below this
constexpr int kSide = 8;                   // side counters for bins fed by flat image regions (see fold_flat_chunk)

// LDS word of joint bin (d1 = render intensity, d2 = warped-frame intensity):
//   word = d1 * 128 + (d2 & 127), low half for d2 < 128, high half for d2 >= 128.
// Each word thus holds the pair (d2, d2 + 128) -- the two operands of the first tree step a[t] += a[t+128]
// (NMI.cu:276-284) -- and a lane that owns the words i, i+16, ..., i+112 of a row owns all operands of the steps
// n = 128, 64, 32, 16 (decode_phase).  The LDS bank of a bin is (d2 & 31): the render intensity does not enter it.

struct Lds {
    uint32_t joint[kWords];    // 128 KiB
    uint32_t hist_render[kBins];
    uint32_t hist_warped[kBins];
    float joint_row_sums[kBins];  // d_JointEntropyShort, kernel.cu:60,90
    uint32_t ovf[2][kOvfCap];     // wrap events: (word << 1) | field; double-buffered by candidate parity
    uint32_t ovf_n[2];
    uint32_t total[2];            // sum of all decoded counters of the candidate (wrap detector), by parity
    uint32_t side_key[2][kSide];  // flat-region side counters: ((word << 1) | field) + 1, 0 = free; by candidate parity
    uint32_t side_cnt[2][kSide];  // their 32-bit counts (added to the decoded counters in decode_phase)
    float table[kLdsTable];       // table[c] for c < kLdsTable (16 KiB); larger counts read the global table
    uint32_t fallback;            // pipelined kernel: a candidate wrapped, finish sequentially on the exact path
    uint32_t redo_n;
    uint32_t redo[4];             // ordinals (within this workgroup) of candidates to score again exactly
};

// Candidate visited by workgroup `b` in its round `r`.  Workgroups are dealt to the 8 XCDs round-robin (b and b + 8
// share an XCD and its L2), so the 32 workgroups of an XCD take 32 CONSECUTIVE ordinals of the visiting order, and the
// host lays the order out in tiles of (few warps) x (few renders): an XCD's round then touches ~12 images (~3.6 MB at
// 640x480, inside its 4 MiB L2) instead of ~29.  Placement is a speed matter only; any order gives the same results.
__device__ __forceinline__ int slot_in_round(int b, int grid) { return (grid & 7) == 0 ? (b & 7) * (grid >> 3) + (b >> 3) : b; }
__device__ __forceinline__ int candidate_at(const GridArgs &a, int ordinal) { return a.order ? a.order[ordinal] : ordinal; }

// ---- histogram phase -------------------------------------------------------------------------------
// One pixel -> one LDS atomic on the packed joint histogram (the reference does three atomics per
// pixel, NMI.cu:46-48; the marginals are recovered as row / column sums of the joint).
// Each 16-bit field is only ever incremented by one, by an add that returns the old word, so every
// wrap of a field is seen by exactly one lane: a low-field wrap carries into the high field (the
// high field then counts d2>=128 hits plus low wraps), a high-field wrap is seen either by a high
// add (old high == 0xFFFF) or by the carrying low add (old word == 0xFFFFFFFF).  Events are rare
// (at most about 2 * W*H / 65536 per candidate) and are replayed when the counters are decoded.
__device__ __forceinline__ uint32_t joint_word(uint32_t d1, uint32_t d2) { return (d1 << 7) | (d2 & 127u); }
__device__ __forceinline__ uint32_t joint_inc(uint32_t d2) { return (d2 & 128u) ? 0x10000u : 1u; }

__device__ __forceinline__ void record_wrap(Lds &lds, int par, uint32_t word, uint32_t val, uint32_t old)
{
    uint32_t k = __hip_atomic_fetch_add(&lds.ovf_n[par], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (k < kOvfCap) lds.ovf[par][k] = (word << 1) | (val >> 16);
    if (val == 1u && old == 0xFFFFFFFFu) {
        k = __hip_atomic_fetch_add(&lds.ovf_n[par], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (k < kOvfCap) lds.ovf[par][k] = (word << 1) | 1u;
    }
}

template <bool BG, bool SHIFTED>
__device__ __forceinline__ void add_pixel(Lds &lds, int par, uint32_t d1, uint32_t d2, int shift)
{
    if (!BG && (d1 == 0 || d2 == 0)) return;  // NMI.cu:85
    if (SHIFTED) {
        d1 >>= shift;
        d2 >>= shift;
    }
    const uint32_t word = joint_word(d1, d2), val = joint_inc(d2);
    const uint32_t old = __hip_atomic_fetch_add(&lds.joint[word], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint32_t field = val * 0xFFFFu;
    if (__builtin_expect((old & field) == field, 0)) record_wrap(lds, par, word, val, old);
}

// Flat chunks.  When, for every active lane of the wavefront, all 16 pixels of the lane carry the same (render, frame)
// pair -- render background over saturated sky or over the frame border, clipped regions -- the plain path would queue
// 64 lanes on one LDS address 16 times over (2 cycles per lane each time: ~12x the cost of a textured chunk).  Folded,
// the 16 updates of a lane become one weighted add, and if the whole wavefront agrees on the pair, one add by one lane
// into a 32-bit side counter (kSide per candidate, replayed in decode_phase): a large flat region then neither
// serialises the LDS nor wraps a 16-bit field, so such frames stay on the one-pass optimistic path.
// Returns false (nothing done) when some active lane is not flat.
__device__ __forceinline__ bool side_add(Lds &lds, int par, uint32_t word, uint32_t high, uint32_t weight)
{
    const uint32_t key1 = ((word << 1) | high) + 1u;
    for (int e = 0; e < kSide; ++e) {
        const uint32_t old = atomicCAS(&lds.side_key[par][e], 0u, key1);
        if (old == 0u || old == key1) {
            atomicAdd(&lds.side_cnt[par][e], weight);
            return true;
        }
    }
    return false;
}

template <bool BG, bool SHIFTED, int HIST>
__device__ __forceinline__ bool fold_flat_chunk(Lds &lds, int par, const uint32_t (&r)[4], const uint32_t (&w)[4], int shift)
{
    const uint32_t rb = r[0] & 0xFFu, wb = w[0] & 0xFFu;
    const bool flat = r[0] == r[1] && r[1] == r[2] && r[2] == r[3] && w[0] == w[1] && w[1] == w[2] && w[2] == w[3] &&
                      r[0] == rb * 0x01010101u && w[0] == wb * 0x01010101u;
    if (!__all(flat)) return false;
    uint32_t d1 = rb, d2 = wb;
    const bool skip = !BG && (d1 == 0 || d2 == 0);  // NMI.cu:85
    if (SHIFTED) {
        d1 >>= shift;
        d2 >>= shift;
    }
    const uint32_t key = (d1 << 8) | d2;
    const uint32_t key0 = __builtin_amdgcn_readfirstlane(key);
    const bool skip0 = __builtin_amdgcn_readfirstlane((uint32_t)skip) != 0;
    const uint32_t word = joint_word(d1, d2), high = d2 >> 7;
    uint32_t weight = 16;
    bool issue = !skip;
    if (__all(key == key0 && skip == skip0)) {  // one lane speaks for the wavefront
        const unsigned long long active = __ballot(1);
        weight = 16u * (uint32_t)__popcll(active);
        issue = issue && (__lane_id() == (uint32_t)__ffsll((long long)active) - 1u);
        if (issue && side_add(lds, par, word, high, weight)) issue = false;
    }
    if (issue) {
        const uint32_t inc = high ? weight << 16 : weight;
        if (HIST == 2) {
            (void)__hip_atomic_fetch_add(&lds.joint[word], inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            const uint32_t old = __hip_atomic_fetch_add(&lds.joint[word], inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t field_old = high ? old >> 16 : old & 0xFFFFu;
            if (field_old + weight > 0xFFFFu) record_wrap(lds, par, word, high ? 0x10000u : 1u, high ? old : (old | 0xFFFFu));
        }
    }
    return true;
}

// 16 pixels of one lane.  HIST selects how wraps of the 16-bit counters are handled:
//   0  returning atomic + test per pixel (serialises on the LDS round trip; kept as the ablation baseline)
//   1  16 returning atomics in flight, one combined wrap test per 16 pixels, flat chunks folded (the exact path)
//   2  non-returning atomics, no test: exact only when no bin can exceed 65535 (first try of the
//      optimistic scheme HIST = 3, see nmi_grid_kernel)
template <bool BG, bool SHIFTED, int HIST, bool FOLD>
__device__ __forceinline__ void add_chunk(Lds &lds, int par, const uint4 &rv, const uint4 &wv, int shift, bool try_flat)
{
    const uint32_t r[4] = {rv.x, rv.y, rv.z, rv.w};
    const uint32_t w[4] = {wv.x, wv.y, wv.z, wv.w};
    if (HIST == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                add_pixel<BG, SHIFTED>(lds, par, (r[q] >> (8 * j)) & 0xFFu, (w[q] >> (8 * j)) & 0xFFu, shift);
        return;
    }
    // Flat chunks (render background over saturated sky or frame border, ...) are folded, see fold_flat_chunk; only the
    // careful loop of histogram_phase asks for it.
    if (FOLD && HIST != 0 && try_flat) {
        if (__builtin_expect(flat_hint(rv, wv), 0)) {
            if (fold_flat_chunk<BG, SHIFTED, HIST>(lds, par, r, w, shift)) return;
        }
    }
    if (HIST == 2 && BG && !SHIFTED) {
        // The hot case, written so that each pixel costs 5 VALU + 1 DS: byte address = d1 * 512 + (d2 & 127) * 4 from one
        // byte-select shift of the render dword and one shift + and-or of the frame dword; increment 1 + 0xFFFF * bit7(d2).
        // hipcc re-derives 7 instructions from the plain C expressions (mask + compare + select for the increment, a
        // separate mask for the render byte), so the five are spelled out: SDWA byte-select shift, shift, and-or, bit-field
        // extract, 24-bit multiply-add.
        char *const base = reinterpret_cast<char *>(lds.joint);
        const uint32_t nine = 9, mask_1fc = 0x1FCu, k_ffff = 0xFFFFu;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t a1, a2, addr, hi, val;
                if (j == 0)
                    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(a1) : "v"(nine), "v"(r[q]));
                else if (j == 1)
                    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(a1) : "v"(nine), "v"(r[q]));
                else if (j == 2)
                    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(a1) : "v"(nine), "v"(r[q]));
                else
                    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(a1) : "v"(nine), "v"(r[q]));
                a2 = j == 0 ? (w[q] << 2) : (w[q] >> (8 * j - 2));
                asm("v_and_or_b32 %0, %1, %3, %2" : "=v"(addr) : "v"(a2), "v"(a1), "s"(mask_1fc));  // VOP3: no literals on gfx9
                hi = __builtin_amdgcn_ubfe(w[q], 8 * j + 7, 1);
                asm("v_mad_u32_u24 %0, %1, %2, 1" : "=v"(val) : "v"(hi), "s"(k_ffff));
                (void)__hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(base + addr), val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        return;
    }
    uint32_t old[16];
    uint32_t any = 0;  // max over pixels of (old | ~field): 0xFFFFFFFF iff some counter wrapped
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t d1 = (r[q] >> (8 * j)) & 0xFFu, d2 = (w[q] >> (8 * j)) & 0xFFu;
            const bool skip = !BG && (d1 == 0 || d2 == 0);  // NMI.cu:85
            if (SHIFTED) {
                d1 >>= shift;
                d2 >>= shift;
            }
            const uint32_t word = joint_word(d1, d2), val = joint_inc(d2);
            if (HIST == 2) {
                if (!skip) (void)__hip_atomic_fetch_add(&lds.joint[word], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                old[q * 4 + j] = 0;
                if (!skip)
                    old[q * 4 + j] = __hip_atomic_fetch_add(&lds.joint[word], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    if (HIST == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t d2 = (w[q] >> (8 * j)) & 0xFFu;
                if (SHIFTED) d2 >>= shift;
                const uint32_t notfield = (d2 & 128u) ? 0x0000FFFFu : 0xFFFF0000u;
                const uint32_t t = old[q * 4 + j] | notfield;
                any = t > any ? t : any;
            }
        if (__builtin_expect(any == 0xFFFFFFFFu, 0)) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint32_t d1 = (r[q] >> (8 * j)) & 0xFFu, d2 = (w[q] >> (8 * j)) & 0xFFu;
                    const bool skip = !BG && (d1 == 0 || d2 == 0);
                    if (SHIFTED) {
                        d1 >>= shift;
                        d2 >>= shift;
                    }
                    const uint32_t val = joint_inc(d2), field = val * 0xFFFFu;
                    if (!skip && (old[q * 4 + j] & field) == field) record_wrap(lds, par, joint_word(d1, d2), val, old[q * 4 + j]);
                }
        }
    }
}

// Histogram phase for one candidate: histogram256Kernel's pixel loop, NMI.cu:79-87.
// NT lanes (tid = 0..NT-1) share the pixels of the candidate -- on the 16-byte path the 16-pixel chunks [c_first, c_end)
// of it (the whole pair: 0, npix / 16; nmi_pix_kernel.hip gives each of a candidate's workgroups a range of its own).
// ROWS (nmi_kernels_rows.hip only): rows that are not whole aligned chunks -- width % 16 != 0 (KITTI's 1241 x 376), or stacks
// that are not 16-byte aligned.  Chunk c = (row y, j-th chunk of the row) starts at byte y * width + 16 j of the frame and at
// ry * width + 16 j of the render (unaligned 16-byte loads); the width % 16 pixels left at the end of every row are added one
// by one after the loop.  c_end is then height * (width / 16).  (Through the byte path below such frames took 4.1x the time per
// pixel: profiles/r04_a/odd_width_time.txt.)
template <bool BG, bool SHIFTED, int HIST, int NT, bool FOLD = true, bool ROWS = false>
__device__ __forceinline__ void histogram_phase(Lds &lds, int par, const GridArgs &a, const uint8_t *__restrict__ render,
                                                const uint8_t *__restrict__ warped, int tid, int c_first, int c_end)
{
    if (ROWS || a.vec_ok) {
        // 16 pixels per lane per step: one 16-byte load from each image (1 KiB per wavefront instruction),
        // the next step's loads issued before this step's atomics.
        const int nchunks = c_end;
        // 32-bit unsigned byte offsets from the (scalar) image bases: one shift per load instead of 64-bit pointer math
        const int row_rem = ROWS ? a.width - (a.chunks_per_row << 4) : 0;  // pixels of a row beyond its whole chunks
        auto ldw = [&](int c) {
            if (ROWS) return *reinterpret_cast<const uint4 *>(warped + (((uint32_t)c << 4) + (uint32_t)__mul24((int)__umulhi((uint32_t)c, a.cpr_magic), row_rem)));
            return *reinterpret_cast<const uint4 *>(warped + ((uint32_t)c << 4));
        };
        // NMI.cu:82: row y of the frame meets row H-1-y of a bottom-up render.  Branch-free for both orientations:
        // render chunk = c + flip_base + y * flip_row with y = c / chunks_per_row (multiply-high by the magic).
        auto ldr = [&](int c) {
            const int y = (int)__umulhi((uint32_t)c, a.cpr_magic);
            if (ROWS) {
                const int ry = a.flip ? a.height - 1 - y : y;
                return *reinterpret_cast<const uint4 *>(render + (((uint32_t)(__mul24(y, a.flip_row) + c + a.flip_base) << 4) + (uint32_t)__mul24(ry, row_rem)));
            }
            return *reinterpret_cast<const uint4 *>(render + ((uint32_t)(__mul24(y, a.flip_row) + c + a.flip_base) << 4));
        };
        // Fast loop.  Software pipeline with two named register sets: the loads of the chunk after next are in flight
        // while the current chunk's 16 atomics issue (a third set measured no faster and costs 8 VGPRs).  Loads are
        // unconditional (index clamped to the last chunk, a valid address) so the code is straight-line and the
        // compiler can wait on exact load counts; only the atomics are predicated on the chunk being in range.
        // The only trace of the flat-region handling in here is flat_hint + a branch that is never taken on textured
        // content: on a hit the wavefront leaves for the careful loop below and stays there for the rest of this
        // candidate (everything the fold needs inside this loop cost 5-11 % of the whole kernel).
        constexpr bool kHint = FOLD && HIST != 0;
        const bool try_flat = kHint && !(a.phase_mask & 4);  // bit 2: ablation switch (careful loop entered, nothing folded)
        const int last = nchunks - 1;
        const int iters = (nchunks - c_first + NT - 1) / NT;  // workgroup-uniform
        int resume = (HIST == 1 && kHint) ? c_first + tid : -1;  // the exact path is cold anyway: careful from the start
        if (resume < 0) {
            int ch = c_first + tid;
            int c0 = min(ch, last);
            uint4 wa = ldw(c0), ra = ldr(c0), wb, rb;
            for (int it = 0; it < iters; it += 2) {
                const int c1 = min(ch + NT, last);
                wb = ldw(c1);
                rb = ldr(c1);
                if (kHint && __builtin_expect(flat_hint(ra, wa), 0)) {
                    resume = ch;
                    break;
                }
                if (ch < nchunks) add_chunk<BG, SHIFTED, HIST, false>(lds, par, ra, wa, a.shift, false);
                const int c2 = min(ch + 2 * NT, last);
                wa = ldw(c2);
                ra = ldr(c2);
                if (kHint && __builtin_expect(flat_hint(rb, wb), 0)) {
                    resume = ch + NT;
                    break;
                }
                if (ch + NT < nchunks) add_chunk<BG, SHIFTED, HIST, false>(lds, par, rb, wb, a.shift, false);
                ch += 2 * NT;
            }
        }
        if (resume >= 0) {
            // Careful loop: same adds, flat chunks folded; one chunk of prefetch.
            int c = min(resume, last);
            uint4 wc = ldw(c), rc = ldr(c);
#pragma unroll 1
            for (int ch = resume; ch < nchunks; ch += NT) {
                const int cn = min(ch + NT, last);
                const uint4 wn = ldw(cn), rn = ldr(cn);
                add_chunk<BG, SHIFTED, HIST, true>(lds, par, rc, wc, a.shift, try_flat);
                wc = wn;
                rc = rn;
            }
        }
        if (ROWS && row_rem > 0) {
            // the last width % 16 pixels of every row
            const int x0 = a.chunks_per_row << 4, n = a.height * row_rem;
            for (int t = tid; t < n; t += NT) {
                const int y = t / row_rem, x = x0 + t - y * row_rem;
                const int ry = a.flip ? (a.height - 1 - y) : y;
                uint32_t d1 = render[ry * a.width + x], d2 = warped[y * a.width + x];
                if (HIST == 2) {
                    if (BG || (d1 != 0 && d2 != 0)) {
                        if (SHIFTED) {
                            d1 >>= a.shift;
                            d2 >>= a.shift;
                        }
                        (void)__hip_atomic_fetch_add(&lds.joint[joint_word(d1, d2)], joint_inc(d2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                } else {
                    add_pixel<BG, SHIFTED>(lds, par, d1, d2, a.shift);
                }
            }
        }
    } else {
        // Any width / alignment: byte loads, position arithmetic as written in NMI.cu:79-83.
        for (int pos = tid; pos < a.npix; pos += NT) {
            const int y = pos / a.width;
            const int x = pos - y * a.width;
            const int ry = a.flip ? (a.height - 1 - y) : y;
            if (HIST == 2) {
                uint32_t d1 = render[ry * a.width + x], d2 = warped[pos];
                if (BG || (d1 != 0 && d2 != 0)) {
                    if (SHIFTED) {
                        d1 >>= a.shift;
                        d2 >>= a.shift;
                    }
                    (void)__hip_atomic_fetch_add(&lds.joint[joint_word(d1, d2)], joint_inc(d2), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            } else {
                add_pixel<BG, SHIFTED>(lds, par, render[ry * a.width + x], warped[pos], a.shift);
            }
        }
    }
}

__device__ __forceinline__ float term(const float *__restrict__ table, uint32_t c)
{
    // ComputeEntropyKernel, NMI.cu:242-263; table[c] = (c/len) * log2f(c/len), table[0] = 0.
    return c ? table[c] : 0.0f;
}
// Same value, served from the LDS copy of the table for the (overwhelmingly common) small counts.
__device__ __forceinline__ float term_lds(const Lds &lds, const float *__restrict__ table, uint32_t c)
{
    float t = lds.table[c < (uint32_t)kLdsTable ? c : 0u];
    if (__builtin_expect(c >= (uint32_t)kLdsTable, 0)) t = table[c];
    return t;
}

// Replays the wrap events of one LDS word onto its two decoded counters.
__device__ __forceinline__ void apply_wraps(const Lds &lds, int par, uint32_t novf, uint32_t word, uint32_t &lo,
                                            uint32_t &hi)
{
    for (uint32_t e = 0; e < novf; ++e) {
        const uint32_t ev = lds.ovf[par][e];
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

// ---- decode phase: counters -> per-bin terms -> row trees (ComputeEntropyKernel + AddvectorParwiseMidKernel) ----
// A wavefront takes 4 joint rows per pass, one per 16-lane DPP row.  Lane i of a row owns the bins
// d2 = i + 16*j (j = 0..15): words i + 16*k (k = 0..7) hold the pairs (d2, d2 + 128).  Every tree step
// n >= 16 of NMI.cu:276-284 then pairs two values of the same lane and the steps n = 8..1 are DPP
// shifts inside the 16-lane row: no LDS traffic besides reading (and clearing) the counters.
// Odd DPP rows start at k = 1 so that the two rows of a 32-lane LDS access group hit disjoint banks.
// ZERO0 (background rule off, NMI.cu:85: a pixel counts only if both intensities are non-zero): the histogram phase
// has counted every pixel -- the skipped ones are exactly row 0 and column 0 of the joint histogram, which are cleared
// here, after they have entered the wrap detector's total.
template <bool ZERO0 = false>
__device__ __forceinline__ void decode_phase(Lds &lds, int par, const GridArgs &a, int wave, int lane)
{
    const uint32_t novf = lds.ovf_n[par] < (uint32_t)kOvfCap ? lds.ovf_n[par] : (uint32_t)kOvfCap;
    const bool side_any = lds.side_key[par][0] != 0u;
    uint32_t wave_total = 0;
    const int i = lane & 15, r = lane >> 4, o = r & 1;
    uint32_t col_lo[8], col_hi[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) col_lo[k] = col_hi[k] = 0;
#pragma unroll 1
    for (int pass = 0; pass < kRowsPerWave / 4; ++pass) {
        const int d1 = wave * kRowsPerWave + pass * 4 + r;
        const uint32_t a0 = d1 * 128 + i + 16 * o;
        uint32_t lo[8], hi[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t idx = k < 7 ? a0 + 16 * k : a0 + 112 - 128 * o;
            const uint32_t wd = lds.joint[idx];
            lds.joint[idx] = 0;  // ready for the next candidate
            lo[k] = wd & 0xFFFFu;
            hi[k] = wd >> 16;
        }
        if (__builtin_expect(novf != 0, 0)) {
#pragma unroll
            for (int k = 0; k < 8; ++k) apply_wraps(lds, par, novf, k < 7 ? a0 + 16 * k : a0 + 112 - 128 * o, lo[k], hi[k]);
        }
        if (__builtin_expect(side_any, 0)) {
            // side counters of flat regions (fold_flat_chunk): entries fill in order, a free one ends the list
            for (int e = 0; e < kSide; ++e) {
                const uint32_t key1 = __builtin_amdgcn_readfirstlane(lds.side_key[par][e]);
                if (key1 == 0u) break;
                const uint32_t sword = (key1 - 1u) >> 1;
                if ((sword >> 9) != (uint32_t)((wave * kRowsPerWave + pass * 4) >> 2)) continue;  // not among this pass's 4 rows
                const uint32_t cnt = lds.side_cnt[par][e];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if ((k < 7 ? a0 + 16 * k : a0 + 112 - 128 * o) == sword) {
                        if ((key1 - 1u) & 1u)
                            hi[k] += cnt;
                        else
                            lo[k] += cnt;
                    }
                }
            }
        }
        uint32_t rsum = 0, cmax = 0;
        if (ZERO0) {
            uint32_t raw = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) raw += lo[k] + hi[k];
            wave_total += row_sum_16(raw);
            if (i == 0) lo[o ? 7 : 0] = 0;  // the bin d2 = 0 of this row
            if (d1 == 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) lo[k] = hi[k] = 0;
            }
        }
        float tl[8], th[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            col_lo[k] += lo[k];
            col_hi[k] += hi[k];
            rsum += lo[k] + hi[k];
            cmax = max(cmax, max(lo[k], hi[k]));
            // straight-line LDS lookups; counts beyond the LDS table are patched below (one branch per pass)
            tl[k] = lds.table[lo[k] & (kLdsTable - 1)];
            th[k] = lds.table[hi[k] & (kLdsTable - 1)];
        }
        if (__builtin_expect(cmax >= (uint32_t)kLdsTable, 0)) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (lo[k] >= (uint32_t)kLdsTable) tl[k] = a.table[lo[k]];
                if (hi[k] >= (uint32_t)kLdsTable) th[k] = a.table[hi[k]];
            }
        }
        rsum = row_sum_16(rsum);
        if (!ZERO0) wave_total += rsum;
        const float x = row_tree_16(lane_tree_16(tl, th));
        if (i == 0) {
            lds.hist_render[d1] = rsum;
            lds.joint_row_sums[d1] = x;
        }
        if (a.dbg_joint) {
            uint32_t *row = a.dbg_joint + d1 * kBins;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int q = (i + 16 * (k + o)) & 127;
                row[q] = lo[k];
                row[q + 128] = hi[k];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int q = (i + 16 * (k + o)) & 127;
        atomicAdd(&lds.hist_warped[q], col_lo[k]);
        atomicAdd(&lds.hist_warped[q + 128], col_hi[k]);
    }
    if (i == 0) atomicAdd(&lds.total[par], wave_total);
}

// Final stage, one wavefront: the three 256-element trees of AddVectorPairwiseKernel (NMI.cu:295-339) run
// side by side in DPP rows 0 (render marginal), 1 (frame marginal), 2 (joint row sums); then the score.
__device__ __forceinline__ void final_phase(Lds &lds, const GridArgs &a, int lane, int p, int w, int s,
                                            unsigned long long &prev_key)
{
    const int i = lane & 15, r = lane >> 4;
    float lo[8], hi[8];
    // All 16 table lookups of a lane are issued back to back and unconditionally (table[0] = 0; rows 2, 3 fetch
    // table[0] and discard it): one memory round trip (1.0 us) instead of one per conditional lookup (2.5 us).  The
    // other wavefronts are already adding the next candidate's pixels and wait for this one at the next barrier.
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
    if ((w == 0 || s == 0) && a.plan) {
        // The search as its own content probe (NMI_OPT_CONTENT_PATH): which bins do the two marginals hold?  16 flags per lane
        // of DPP rows 0 and 1, ORed into the plan's masks (LevelPlan::seen) while the table lookups above are in flight --
        // fire-and-forget device atomics, issued as the search goes, not at its end (8,000 of them from all workgroups' exits
        // into one cache line put 1.5 us on the end of every search).  Only the candidates of the grid's first row and first
        // column do this: between them they show every render and every warp once.
        uint32_t m = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) m |= (cl[k] != 0u ? 1u << k : 0u) | (ch[k] != 0u ? 0x100u << k : 0u);
        if (lane < 32 && m) __hip_atomic_fetch_or(const_cast<uint32_t *>(&a.plan->seen[lane]), m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    if (a.dbg_h1 && lane < 64) {
        for (int t = lane; t < kBins; t += 64) {
            a.dbg_h1[t] = lds.hist_render[t];
            if (a.dbg_h2) a.dbg_h2[t] = lds.hist_warped[t];
        }
    }
    if (lane == 0) commit_score(a, p, w, s, a1, a2, a3, prev_key);
}

// End of a workgroup of nmi_grid_kernel, by all of wavefront 0: publish_winner (nmi_device.h) plus the content probe's share.
// The workgroup that draws the last ticket fetches the plan's bins-seen masks (final_phase ORs them in; it zeroes them for
// the next search) in the same round trip as the final key, posts the winner first and then (nr, nw) to the pinned word
// the context watches.  Nothing waits for the ORs of other workgroups: a straggling OR can cost a bin in this count or add
// one to the next search's -- the count is a hint for the host's choice of kernels, every few-levels launch probes its own
// stacks exactly.
// `expected`: workgroups of the launch that call this (all of them: gridDim.x; nmi_pix_kernel: the candidates' owners).
__device__ __forceinline__ void finish_search(const GridArgs &a, int lane, unsigned long long prev_key, uint32_t expected)
{
    LevelPlan *plan = const_cast<LevelPlan *>(a.plan);
    uint32_t arrived = 0;
    if (lane == 0) {
        const unsigned int one = prev_key == ~0ull ? 2u : 1u;  // always 1; ties the ticket to this workgroup's maxes (publish_winner)
        arrived = __hip_atomic_fetch_add(a.done, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (__builtin_amdgcn_readfirstlane(arrived) != expected - 1) return;
    unsigned long long final_key = 0;
    uint32_t bits = 0;
    unsigned long long *post = nullptr;
    uint32_t state = 0, max_joint = 0;
    if (lane == 0) final_key = __hip_atomic_load(a.key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (plan && lane < 32) bits = __hip_atomic_exchange(&plan->seen[lane], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (plan && lane == 0) post = plan->seen_post, state = plan->seen_state, max_joint = plan->seen_max_joint;  // (one round trip for all of them)
    if (lane == 0) {
        __hip_atomic_store(a.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a.out_key) __hip_atomic_store(a.out_key, final_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (a.mailbox)
            __hip_atomic_store(&a.mailbox->word, final_key | ((unsigned long long)(a.seq & 1u) << 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (plan) {
        const uint32_t n = row_sum_16((uint32_t)__popc(bits));  // lanes 0 / 16: bins seen in the render / frame marginals
        const uint32_t nw = (uint32_t)__shfl((int)n, 16, 64);
        // Only a CHANGE of the verdict goes to the host: a store to pinned host memory holds the end of the kernel back by a trip
        // over PCIe (0.8 us on every search, measured), and the host has no use for a confirmation.
        const uint32_t few = (n > 0u && nw > 0u && n * nw <= max_joint) ? 1u : 0u;
        if (lane == 0 && post && few != state) {
            plan->seen_state = few;
            // the word's upper half only has to differ from the previous post's: the 100 MHz clock serves (no counter to load)
            const uint32_t stamp = 0x80000000u | (uint32_t)wall_clock64();
            __hip_atomic_store(post, ((unsigned long long)stamp << 32) | ((unsigned long long)n << 16) | nw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// One candidate start to finish on the exact path (returning atomics + wrap bookkeeping + flat-region folding), all 16
// wavefronts.  It runs only for candidates with a bin above 65535 hits; the kernels call it from a separate cold loop
// AFTER their hot loop, never inside it: inlined into the hot loop it cost ~10 % there (spills, code size), and as a
// real function call inside the loop ~25 %.  Uses the parity-0 event list / total and leaves them, hist_warped and
// the joint counters zero.
template <bool SHIFTED, bool BG = true, bool ROWS = false>
__device__ __forceinline__ void exact_candidate(Lds &lds, const GridArgs &a, int tid, int p, unsigned long long &prev_key)
{
    const int lane = tid & 63, wave = tid >> 6;
    const int w = p / a.S_local, s = p - w * a.S_local;
    __syncthreads();  // wavefront 0 may still be finishing the previous candidate's final phase (it resets shared state)
    histogram_phase<true, SHIFTED, 1, kBlock, true, ROWS>(lds, 0, a, a.render_stack + (size_t)s * a.npix, a.warp_stack + (size_t)w * a.npix, tid, 0,
                                                          ROWS ? a.height * a.chunks_per_row : a.npix >> 4);
    __syncthreads();
    decode_phase<!BG>(lds, 0, a, wave, lane);
    __syncthreads();
    if (wave == 0) {
        final_phase(lds, a, lane, p, w, s, prev_key);
        for (int t = lane; t < kBins; t += 64) lds.hist_warped[t] = 0;
        if (lane == 0) lds.ovf_n[0] = lds.total[0] = 0;
        if (lane < kSide) lds.side_key[0][lane] = lds.side_cnt[0][lane] = 0;
    }
    __syncthreads();
}

}  // namespace

// One workgroup per candidate (grid-stride over the candidates of this launch).
//
// HIST = 3 (default with BG on): the histogram phase uses non-returning atomics and no wrap test; the
// decode phase sums every decoded counter, and since a wrapped 16-bit field always loses weight
// (a low wrap turns 65536 hits into one carry, a high wrap drops 65536), the sum equals the pixel
// count iff no field wrapped.  A candidate that fails the test is simply histogrammed again with the
// exact wrap bookkeeping (HIST = 1 path); only frames with a bin above 65535 hits ever pay that.
//
// Two barriers per candidate: B1 histogram -> decode, B2 decode -> (wavefront 0: three final trees +
// score + arg-max) || (all other wavefronts: next candidate's histogram phase).  The small per-candidate
// state that the two sides would share is double-buffered by candidate parity.
//
// nmi_kernels_gated.hip compiles this file a second time with NMI_GRID_KERNEL_GATED defined: the same kernel under the
// name nmi_grid_kernel_gated, which returns at once unless the few-levels kernels enqueued before it handed the search
// back (plan->use == 0, nmi_fewlevels_kernel.hip).  A second translation unit rather than a template parameter because
// this kernel sits at its register cap: the mere presence of more instantiations in this unit changed its allocation.
//
// nmi_kernels_stamped.hip compiles it a third time (NMI_GRID_KERNEL_STAMPED) as nmi_grid_kernel_stamped: the same code plus
// wall-clock stamps of every workgroup's first candidate at the phase boundaries (NMI_OPT_STAMPS, tools/grid_stamps.py).
// nmi_pix_kernel.hip includes this file for its device functions only (NMI_KERNELS_DEVICE_ONLY).
#ifndef NMI_KERNELS_DEVICE_ONLY
#if defined(NMI_GRID_KERNEL_GATED)
#define NMI_GRID_KERNEL_NAME nmi_grid_kernel_gated
#elif defined(NMI_GRID_KERNEL_STAMPED)
#define NMI_GRID_KERNEL_NAME nmi_grid_kernel_stamped
#elif defined(NMI_GRID_KERNEL_ROWS)
#define NMI_GRID_KERNEL_NAME nmi_grid_kernel_rows
#else
#define NMI_GRID_KERNEL_NAME nmi_grid_kernel
#endif
#ifdef NMI_GRID_KERNEL_STAMPED
#define NMI_GRID_STAMP(k)                                                                                         \
    do {                                                                                                          \
        if (a.dbg_stamps && tid == 0 && stamp_on) a.dbg_stamps[blockIdx.x * 8 + (k)] = wall_clock64();            \
    } while (0)
#else
#define NMI_GRID_STAMP(k) \
    do {                  \
    } while (0)
#endif
template <bool BG, bool SHIFTED, int HIST>
__global__ __launch_bounds__(NMI_BLOCK_THREADS) void NMI_GRID_KERNEL_NAME(GridArgs a)
{
    __shared__ Lds lds;
#ifdef NMI_GRID_KERNEL_GATED
    if (a.plan->use != 0u) return;
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
#ifdef NMI_GRID_KERNEL_STAMPED
    bool stamp_on = true;  // first candidate of this workgroup only
#endif
    NMI_GRID_STAMP(0);
    constexpr bool kOptimistic = HIST == 3;
    constexpr int kFirst = kOptimistic ? 2 : HIST;
#ifdef NMI_GRID_KERNEL_ROWS
    constexpr bool kRows = true;  // rows that are not whole aligned chunks (histogram_phase)
#define NMI_ALL_CHUNKS (a.height * a.chunks_per_row)
#else
    constexpr bool kRows = false;
#define NMI_ALL_CHUNKS (a.npix >> 4)
#endif
    // Background rule off (NMI.cu:85) on the optimistic path: every pixel is counted, so that the wrap detector knows the
    // expected total (W*H), and decode_phase clears the row and the column of intensity 0 -- the skipped pixels.  Only
    // with 256 bins: the rule looks at the intensity before the shift, and bin 0 of a shifted histogram also holds
    // intensities 1 .. 2^shift - 1 (launch_grid sends BG off + shift to the exact path, HIST = 1).
    constexpr bool kZero0 = !BG && kOptimistic && !SHIFTED;
    constexpr bool kCountAll = BG || kZero0;

    if (blockIdx.x == 0 && tid == 0 && a.reset_key) *a.reset_key = 0ull;  // next launch's slot; idle during this one
    // The LDS copy of the term table is first needed by the first decode phase: fetch it now, park it in
    // registers across the first histogram phase, store it before the first barrier.
    float tab[kLdsTable / kBlock];
#pragma unroll
    for (int k = 0; k < kLdsTable / kBlock; ++k) {
        const int c = tid + k * kBlock;
        tab[k] = a.table[c <= a.npix ? c : 0];
    }
    {
        uint4 *j4 = reinterpret_cast<uint4 *>(lds.joint);
        const uint4 z = {0, 0, 0, 0};
        for (int i = tid; i < kWords / 4; i += kBlock) j4[i] = z;
    }
    if (tid < kBins) lds.hist_warped[tid] = 0;
    if (tid < 2) lds.ovf_n[tid] = lds.total[tid] = 0;
    if (tid < 2 * kSide) (&lds.side_key[0][0])[tid] = (&lds.side_cnt[0][0])[tid] = 0;
    bool table_pending = true;
    int exact_from = -1;  // first ordinal of this workgroup that needs the exact path (workgroup-uniform)
    __syncthreads();
    NMI_GRID_STAMP(1);

    const int total = a.S_local * a.Wn;
    unsigned long long prev_key = 0;
    int par = 0;
    const int slot = slot_in_round(blockIdx.x, gridDim.x);
    int ordinal = slot;  // this workgroup visits ordinals slot, slot + grid, slot + 2 grid, ...
    for (; ordinal < total; ordinal += gridDim.x, par ^= 1) {
        const int p = candidate_at(a, ordinal);
        const int w = p / a.S_local;
        const int s = p - w * a.S_local;
        const uint8_t *render = a.render_stack + (size_t)s * a.npix;
        const uint8_t *warped = a.warp_stack + (size_t)w * a.npix;

        if (a.phase_mask & 1) {
            if (a.phase_mask & 8) {  // ablation: only half of the wavefronts take part in the histogram phase
                if (wave < kWaves / 2) histogram_phase<kCountAll, SHIFTED, kFirst, kBlock / 2, true, kRows>(lds, par, a, render, warped, tid, 0, NMI_ALL_CHUNKS);
            } else
                histogram_phase<kCountAll, SHIFTED, kFirst, kBlock, true, kRows>(lds, par, a, render, warped, tid, 0, NMI_ALL_CHUNKS);
        }
        if (table_pending) {
#pragma unroll
            for (int k = 0; k < kLdsTable / kBlock; ++k) lds.table[tid + k * kBlock] = tab[k];
            table_pending = false;
        }
        NMI_GRID_STAMP(2);  // this wavefront's share of the pixels done
#ifdef NMI_GRID_KERNEL_STAMPED
        // ... and every wavefront's, behind the [workgroups][8] block: [workgroups][16]
        if (a.dbg_stamps && lane == 0 && stamp_on) a.dbg_stamps[gridDim.x * 8 + blockIdx.x * 16 + wave] = wall_clock64();
#endif
        __syncthreads();  // B1
        NMI_GRID_STAMP(3);
        if (a.phase_mask & 2) decode_phase<kZero0>(lds, par, a, wave, lane);
        __syncthreads();  // B2
        NMI_GRID_STAMP(4);
        if (kOptimistic && (a.phase_mask & 3) == 3 && lds.total[par] != (uint32_t)a.npix) {
            // Some counter wrapped (workgroup-uniform, rare).  This candidate and, since the same frame and renders
            // come back, all later ones of this workgroup are scored on the exact path in the cold loop below.
            exact_from = ordinal;
            break;
        }
        if (wave == 0) {
            if (a.phase_mask & 2) final_phase(lds, a, lane, p, w, s, prev_key);
            for (int t = lane; t < kBins; t += 64) lds.hist_warped[t] = 0;
            if (lane == 0) {
                lds.ovf_n[par] = 0;       // consumed by this candidate's decode; next used two candidates on
                lds.total[par ^ 1] = 0;   // read by everyone right after the previous B2; next candidate adds to it
            }
            if (lane < kSide) {  // like ovf_n[par]
                int l = lane;
                asm volatile("" : "+v"(l));  // keep the two addresses out of long-lived (spilled) registers
                lds.side_key[par][l] = lds.side_cnt[par][l] = 0;
            }
        }
        NMI_GRID_STAMP(5);  // score committed (wavefront 0)
#ifdef NMI_GRID_KERNEL_STAMPED
        stamp_on = false;
#endif
    }

    if (kOptimistic && exact_from >= 0) {
        __syncthreads();  // everyone has read the failed total; wavefront 0 is past the previous candidate's final phase
        if (tid < kBins) lds.hist_warped[tid] = 0;
        if (tid < 2) lds.total[tid] = lds.ovf_n[tid] = 0;
        if (tid < 2 * kSide) (&lds.side_key[0][0])[tid] = (&lds.side_cnt[0][0])[tid] = 0;
        __syncthreads();
        for (int o = exact_from; o < total; o += gridDim.x) exact_candidate<SHIFTED, !kZero0, kRows>(lds, a, tid, candidate_at(a, o), prev_key);
    }

    if (wave == 0 && !(a.phase_mask & 16)) {  // bit 4: timing experiment without the protocol (no result)
        finish_search(a, lane, prev_key, gridDim.x);
    }
#ifdef NMI_GRID_KERNEL_STAMPED
    stamp_on = true;
#endif
    NMI_GRID_STAMP(6);
}

#if defined(NMI_GRID_KERNEL_ROWS)
hipError_t launch_grid_rows(const GridArgs &a, int workgroups, bool use_bg, hipStream_t stream)
{
    if (a.width < 32 || (a.hist_variant != 1 && a.hist_variant != 3)) return hipErrorInvalidValue;
    const dim3 grid(workgroups), block(kBlock);
    const bool exact_only = a.hist_variant == 1 || (!use_bg && a.shift != 0);  // (as launch_grid: BG off below 256 bins has no optimistic path)
    if (exact_only) {
        if (use_bg) {
            if (a.shift != 0) hipLaunchKernelGGL((nmi_grid_kernel_rows<true, true, 1>), grid, block, 0, stream, a);
            else hipLaunchKernelGGL((nmi_grid_kernel_rows<true, false, 1>), grid, block, 0, stream, a);
        } else {
            if (a.shift != 0) hipLaunchKernelGGL((nmi_grid_kernel_rows<false, true, 1>), grid, block, 0, stream, a);
            else hipLaunchKernelGGL((nmi_grid_kernel_rows<false, false, 1>), grid, block, 0, stream, a);
        }
    } else if (a.shift != 0) {
        hipLaunchKernelGGL((nmi_grid_kernel_rows<true, true, 3>), grid, block, 0, stream, a);
    } else if (use_bg) {
        hipLaunchKernelGGL((nmi_grid_kernel_rows<true, false, 3>), grid, block, 0, stream, a);
    } else {
        hipLaunchKernelGGL((nmi_grid_kernel_rows<false, false, 3>), grid, block, 0, stream, a);
    }
    return hipGetLastError();
}
#elif defined(NMI_GRID_KERNEL_STAMPED)
// tools only: 256 bins, background rule on, default histogram variant
hipError_t launch_grid_stamped(const GridArgs &a, int workgroups, hipStream_t stream)
{
    if (a.hist_variant != 3 || a.shift != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL((nmi_grid_kernel_stamped<true, false, 3>), dim3(workgroups), dim3(kBlock), 0, stream, a);
    return hipGetLastError();
}
#elif defined(NMI_GRID_KERNEL_GATED)
// The fallback launch behind launch_fewlevels: 256 bins, default histogram variant.
hipError_t launch_grid_gated(const GridArgs &a, int workgroups, bool use_bg, hipStream_t stream)
{
    if (a.hist_variant != 3 || !a.plan || (a.shift != 0 && !use_bg)) return hipErrorInvalidValue;
    const dim3 grid(workgroups), block(kBlock);
    if (a.shift != 0)
        hipLaunchKernelGGL((nmi_grid_kernel_gated<true, true, 3>), grid, block, 0, stream, a);
    else if (use_bg)
        hipLaunchKernelGGL((nmi_grid_kernel_gated<true, false, 3>), grid, block, 0, stream, a);
    else
        hipLaunchKernelGGL((nmi_grid_kernel_gated<false, false, 3>), grid, block, 0, stream, a);
    return hipGetLastError();
}
#else  // everything below belongs to the primary translation unit only

#ifdef NMI_BUILD_ABLATIONS  // experiments kept for tools/ablate.py; not part of the shipped library (profiles/NOTES.md)
// ---------------------------------------------------------------------------------------------------------
// Pipelined ("wavefront-specialised") form of the same computation (NMI_OPT_HIST_VARIANT = 4, experimental).
// Exact, covered by the parity tests, but measured SLOWER than the sequential kernel on MI355X (114 vs 95 us per 729
// candidates): the decode wavefronts take issue slots from the histogram wavefronts, 8 histogram wavefronts add
// pixels 22 % slower than 16, and the drain is a third serial step.  Kept as an ablation; see profiles/NOTES.md.
//
// The histogram phase is bound by the LDS atomic unit, the decode arithmetic by VALU issue and latency; run one
// after the other (kernel above) a CU leaves each unit idle in turn.  Here the 16 wavefronts split into
//   H = wavefronts 0..7   histogram phase of candidate k (non-returning atomics into the packed LDS joint)
//   D = wavefronts 8..15  decode arithmetic of candidate k-1 (terms, trees, marginals) from its drained counters
// with two workgroup barriers per candidate: X (histogram k and arithmetic k-1 done) and Y (all wavefronts have
// drained candidate k's packed counters from LDS to the workgroup's scratch slab in L2 and cleared the LDS words;
// meanwhile wavefront 0 forms the score of k-1).  D then reads candidate k back from the slab while H is already
// adding candidate k+1.  Only the drain (an LDS read + clear sweep with coalesced stores) stays serial with H.
// Counter wraps are detected by the pixel-count test as before; the first wrapped candidate switches the
// workgroup to the sequential exact path (below) for that candidate and all that follow.
// ---------------------------------------------------------------------------------------------------------
namespace {

constexpr int kHalf = kBlock / 2;               // lanes per role
constexpr int kDWaves = kWaves / 2;             // 8 decode wavefronts
constexpr int kDRows = kBins / kDWaves;         // 32 joint rows per decode wavefront
constexpr int kDPasses = kDRows / 4;            // 8 passes of 4 rows (one per 16-lane DPP row)

// Drain, all 16 wavefronts: packed counters LDS -> this workgroup's scratch slab in global memory (it stays in L2),
// clearing the LDS words.  The slab is written in the order the decode wavefronts read it -- [dwave][pass][k][lane],
// 256 contiguous bytes per wavefront instruction -- with the word ownership of decode_phase (lane i of a 16-lane row
// owns words i + 16*k of its joint row).
__device__ __forceinline__ void drain_to_scratch(Lds &lds, uint32_t *__restrict__ slab, int wave, int lane)
{
    const int i = lane & 15, r = lane >> 4, o = r & 1;
    const int dwave = wave >> 1, pass0 = (wave & 1) * (kDPasses / 2);
#pragma unroll
    for (int pp = 0; pp < kDPasses / 2; ++pp) {
        const int pass = pass0 + pp;
        const int d1 = dwave * kDRows + pass * 4 + r;
        const uint32_t a0 = d1 * 128 + i + 16 * o;
        uint32_t wd[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t idx = k < 7 ? a0 + 16 * k : a0 + 112 - 128 * o;
            wd[k] = lds.joint[idx];
            lds.joint[idx] = 0;
        }
        uint32_t *dst = slab + ((dwave * kDPasses + pass) * 8) * 64 + lane;
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[k * 64] = wd[k];
    }
}

// D: per-bin terms, row trees, marginals for the 32 rows of this decode wavefront, read back from the slab
// (agent-scope loads: the slab was written by other wavefronts of this CU through L2, this CU's L1 may hold the
// lines of an earlier candidate).  The next pass's words are in flight while the current pass is reduced.
__device__ __forceinline__ void load_pass(const uint32_t *__restrict__ src, uint32_t (&wd)[8])
{
#pragma unroll
    for (int k = 0; k < 8; ++k) wd[k] = __hip_atomic_load(src + k * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void decode_pass(Lds &lds, const GridArgs &a, int d1, int i, const uint32_t (&wd)[8],
                                            uint32_t (&col_lo)[8], uint32_t (&col_hi)[8], uint32_t &wave_total)
{
    uint32_t lo[8], hi[8], rsum = 0, cmax = 0;
    float tl[8], th[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        lo[k] = wd[k] & 0xFFFFu;
        hi[k] = wd[k] >> 16;
        col_lo[k] += lo[k];
        col_hi[k] += hi[k];
        rsum += lo[k] + hi[k];
        cmax = max(cmax, max(lo[k], hi[k]));
        tl[k] = lds.table[lo[k] & (kLdsTable - 1)];
        th[k] = lds.table[hi[k] & (kLdsTable - 1)];
    }
    if (__builtin_expect(cmax >= (uint32_t)kLdsTable, 0)) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (lo[k] >= (uint32_t)kLdsTable) tl[k] = a.table[lo[k]];
            if (hi[k] >= (uint32_t)kLdsTable) th[k] = a.table[hi[k]];
        }
    }
    rsum = row_sum_16(rsum);
    wave_total += rsum;
    const float x = row_tree_16(lane_tree_16(tl, th));
    if (i == 0) {
        lds.hist_render[d1] = rsum;
        lds.joint_row_sums[d1] = x;
    }
}

__device__ __forceinline__ void decode_from_scratch(Lds &lds, const GridArgs &a, const uint32_t *__restrict__ slab, int dwave,
                                                    int lane)
{
    const int i = lane & 15, r = lane >> 4, o = r & 1;
    uint32_t col_lo[8], col_hi[8], wave_total = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) col_lo[k] = col_hi[k] = 0;
    const uint32_t *src = slab + (dwave * kDPasses * 8) * 64 + lane;
    uint32_t wa[8], wb[8];
    load_pass(src, wa);
#pragma unroll 1
    for (int pass = 0; pass < kDPasses; pass += 2) {
        load_pass(src + (pass + 1) * 8 * 64, wb);
        decode_pass(lds, a, dwave * kDRows + pass * 4 + r, i, wa, col_lo, col_hi, wave_total);
        if (pass + 2 < kDPasses) load_pass(src + (pass + 2) * 8 * 64, wa);
        decode_pass(lds, a, dwave * kDRows + (pass + 1) * 4 + r, i, wb, col_lo, col_hi, wave_total);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int q = (i + 16 * (k + o)) & 127;
        atomicAdd(&lds.hist_warped[q], col_lo[k]);
        atomicAdd(&lds.hist_warped[q + 128], col_hi[k]);
    }
    if (i == 0) atomicAdd(&lds.total[0], wave_total);
}

}  // namespace

template <bool SHIFTED>
__global__ __launch_bounds__(NMI_BLOCK_THREADS) void nmi_grid_kernel_ws(GridArgs a)
{
    __shared__ Lds lds;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const bool is_hist = wave < kDWaves;  // wavefront-uniform role
    const int dwave = wave - kDWaves;

    if (blockIdx.x == 0 && tid == 0 && a.reset_key) *a.reset_key = 0ull;
    {
        uint4 *j4 = reinterpret_cast<uint4 *>(lds.joint);
        const uint4 z = {0, 0, 0, 0};
        for (int i = tid; i < kWords / 4; i += kBlock) j4[i] = z;
    }
    for (int c = tid; c < kLdsTable; c += kBlock) lds.table[c] = a.table[c <= a.npix ? c : 0];
    if (tid < kBins) lds.hist_warped[tid] = 0;
    if (tid < 2) lds.ovf_n[tid] = lds.total[tid] = 0;
    if (tid < 2 * kSide) (&lds.side_key[0][0])[tid] = (&lds.side_cnt[0][0])[tid] = 0;
    if (tid == 0) lds.fallback = lds.redo_n = 0;
    __syncthreads();

    const int total = a.S_local * a.Wn;
    const int slot = slot_in_round(blockIdx.x, gridDim.x);
    const int n = slot < total ? (total - slot + (int)gridDim.x - 1) / (int)gridDim.x : 0;  // candidates of this workgroup
    unsigned long long prev_key = 0;
    // two slabs per workgroup, alternating by candidate: D reads slab (k-1)&1 while the drain of candidate k fills slab k&1
    uint32_t *const slab0 = a.scratch + (size_t)blockIdx.x * 2 * kWords;

    int k = 0;        // stage: H works on candidate k, D on candidate k-1
    bool bail = false;
    for (; k <= n && !bail; ++k) {
        const int p = k < n ? candidate_at(a, slot + k * (int)gridDim.x) : 0;
        if (is_hist) {
            if (k < n && (a.phase_mask & 1)) {
                const int w = p / a.S_local, s = p - w * a.S_local;
                histogram_phase<true, SHIFTED, 2, kHalf, false>(lds, 0, a, a.render_stack + (size_t)s * a.npix,
                                                         a.warp_stack + (size_t)w * a.npix, tid, 0, a.npix >> 4);
            }
        } else if (k > 0 && (a.phase_mask & 2)) {
            decode_from_scratch(lds, a, slab0 + ((k - 1) & 1) * kWords, dwave, lane);
        }
        __syncthreads();  // X
        if (k < n && !(a.phase_mask & 4)) drain_to_scratch(lds, slab0 + (k & 1) * kWords, wave, lane);
        if (is_hist) {
            if (wave == 0 && k > 0) {
                const int pp = candidate_at(a, slot + (k - 1) * (int)gridDim.x), w = pp / a.S_local, s = pp - w * a.S_local;
                if ((a.phase_mask & 7) == 3 && lds.total[0] != (uint32_t)a.npix) {
                    // a 16-bit counter wrapped in candidate k-1: hand it (and everything after it) to the exact path
                    if (lane == 0) {
                        lds.fallback = 1;
                        lds.redo[lds.redo_n++] = (uint32_t)(k - 1);
                    }
                } else {
                    final_phase(lds, a, lane, pp, w, s, prev_key);
                }
                for (int t = lane; t < kBins; t += 64) lds.hist_warped[t] = 0;
                if (lane == 0) lds.total[0] = 0;
            }
        }
        __builtin_amdgcn_s_waitcnt(0);  // the slab stores of this wavefront have reached L2 before anyone is released
        __syncthreads();  // Y
        bail = lds.fallback != 0;
    }

    if (bail) {
        // Stage k-1 detected the wrap; candidate k-1 (if any) has already been drained into the D registers.
        const int kd = k - 1;  // ordinal of the drained candidate
        if (kd < n) {
            if (!is_hist) decode_from_scratch(lds, a, slab0 + (kd & 1) * kWords, dwave, lane);
            __syncthreads();
            if (wave == 0) {
                const int pp = candidate_at(a, slot + kd * (int)gridDim.x), w = pp / a.S_local, s = pp - w * a.S_local;
                if (lds.total[0] != (uint32_t)a.npix) {
                    if (lane == 0) lds.redo[lds.redo_n++] = (uint32_t)kd;
                } else {
                    final_phase(lds, a, lane, pp, w, s, prev_key);
                }
                for (int t = lane; t < kBins; t += 64) lds.hist_warped[t] = 0;
                if (lane == 0) lds.total[0] = 0;
            }
            __syncthreads();
        }
        const int nredo = (int)lds.redo_n;
        for (int e = 0; e < nredo; ++e)
            exact_candidate<SHIFTED>(lds, a, tid, candidate_at(a, slot + (int)lds.redo[e] * (int)gridDim.x), prev_key);
        for (int kk = kd + 1; kk < n; ++kk) exact_candidate<SHIFTED>(lds, a, tid, candidate_at(a, slot + kk * (int)gridDim.x), prev_key);
    }

    if (tid == 0) publish_winner(a, prev_key);
}

#endif  // NMI_BUILD_ABLATIONS

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

template <int HIST>
static void launch_hist(const GridArgs &a, dim3 grid, dim3 block, bool use_bg, hipStream_t stream)
{
    const bool shifted = a.shift != 0;
    if (use_bg) {
        if (shifted)
            hipLaunchKernelGGL((nmi_grid_kernel<true, true, HIST>), grid, block, 0, stream, a);
        else
            hipLaunchKernelGGL((nmi_grid_kernel<true, false, HIST>), grid, block, 0, stream, a);
    } else {
        if (shifted)
            hipLaunchKernelGGL((nmi_grid_kernel<false, true, HIST>), grid, block, 0, stream, a);
        else
            hipLaunchKernelGGL((nmi_grid_kernel<false, false, HIST>), grid, block, 0, stream, a);
    }
}

hipError_t launch_grid(const GridArgs &a, int workgroups, bool use_bg, hipStream_t stream)
{
    dim3 grid(workgroups), block(kBlock);
    // rows that are not whole aligned 16-byte chunks: the unaligned-row form instead of the byte path (frames under 32 pixels
    // of width keep the byte path)
    if (!a.vec_ok && a.width >= 32 && (a.hist_variant == 1 || a.hist_variant == 3) && !(a.phase_mask & 8)) return launch_grid_rows(a, workgroups, use_bg, stream);
    switch (a.hist_variant) {
    case 1: launch_hist<1>(a, grid, block, use_bg, stream); break;
    case 3:
        if (a.dbg_stamps && use_bg && a.shift == 0) return launch_grid_stamped(a, workgroups, stream);  // tools/grid_stamps.py
        // The wrap detector of HIST = 3 needs the expected pixel count: W*H with BG on, and with BG off at 256 bins (the
        // kernel then counts every pixel and clears row / column 0 afterwards); BG off with fewer bins takes the exact path.
        if (use_bg || a.shift == 0)
            launch_hist<3>(a, grid, block, use_bg, stream);
        else
            launch_hist<1>(a, grid, block, use_bg, stream);
        break;
#ifdef NMI_BUILD_ABLATIONS
    case 0: launch_hist<0>(a, grid, block, use_bg, stream); break;
    case 2: launch_hist<2>(a, grid, block, use_bg, stream); break;
    case 4:
        // pipelined kernel (experimental).  It has no debug exports and needs BG on.
        if (use_bg && !a.dbg_joint && !a.dbg_h1 && !a.dbg_h2 && !a.dbg_sums) {
            if (a.shift != 0)
                hipLaunchKernelGGL((nmi_grid_kernel_ws<true>), grid, block, 0, stream, a);
            else
                hipLaunchKernelGGL((nmi_grid_kernel_ws<false>), grid, block, 0, stream, a);
        } else if (use_bg) {
            launch_hist<3>(a, grid, block, use_bg, stream);
        } else {
            launch_hist<1>(a, grid, block, use_bg, stream);
        }
        break;
#endif
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

bool ablation_variants_built()
{
#ifdef NMI_BUILD_ABLATIONS
    return true;
#else
    return false;
#endif
}

int grid_kernel_lds_bytes() { return (int)sizeof(Lds); }
size_t grid_kernel_scratch_bytes(int workgroups) { return (size_t)workgroups * 2 * kWords * sizeof(uint32_t); }

#endif  // primary translation unit
#endif  // !NMI_KERNELS_DEVICE_ONLY

}  // namespace nmi
