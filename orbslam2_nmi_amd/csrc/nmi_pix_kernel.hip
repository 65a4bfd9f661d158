// nmi_pix_kernel.hip -- the scoring path for MID-SIZE grids (33 ... 128 candidates on 256 compute units): P workgroups per
// candidate, each adding a PIXEL RANGE of the pair into a whole packed joint histogram of its own.
//
// Why: the live search seeds its grid from the drift and collapses every axis whose step falls under the minimum
// (src/Tracking.cc:2014-2043, Thirdparty/Localization/nmiSearchKernel.cpp:124-141): 27 / 81 / 243-candidate grids are its
// typical levels, and a rank's share of a sharded 729-candidate grid is 91.  nmi_grid_kernel gives a candidate to one
// workgroup = one CU: 81 candidates take as long as 256 (35 us) with two thirds of the chip idle.  The row split of
// nmi_split_kernel.hip does not help there: a part issues one LDS atomic instruction per 64 pixels of the WHOLE pair
// however few of its lanes own them, and the time of nmi_grid_kernel's histogram phase is its LDS atomic instructions.
//
// How: workgroup (candidate p, range q) runs nmi_grid_kernel's own histogram phase (NMI.cu:79-87; nmi_kernels.hip:
// packed 16-bit counters, 128 KiB of LDS, one non-returning atomic per pixel, flat regions folded) over chunks
// of its own.  Range 0 is the candidate's OWNER; ranges 1 .. P-1 are HELPERS: a helper writes the 16-byte units of its packed
// histogram that hold a count (+ its flat-region side counters) to its block in memory with write-through stores, every wave
// drains its stores, and after the workgroup's barrier the launch's tag goes out with the masks that say which units came
// (MI355X_MICROARCH.md "Valid forms", first row of the table: sc1 stores, drained, signalled after the barrier; the owner's
// waves poll with sc1 loads and read every byte with 16-byte sc1 loads).  The owner adds the helpers' words to its own LDS
// words -- packed fields add like the counters they are -- and from there on is nmi_grid_kernel:
// decode_phase (ComputeEntropyKernel + AddvectorParwiseMidKernel, NMI.cu:230-287), the wrap detector, final_phase
// (AddVectorPairwiseKernel, NMI.cu:290-363), rating store, arg-max, completion.  Results are bit-identical.
//   * Counter wraps: a wrapped 16-bit field always LOSES weight, in a helper, in the owner or in the merge, so the sum of all
//     decoded counters still equals W*H iff nothing wrapped; a candidate that fails is redone by its owner alone on the exact
//     path (exact_candidate), as in nmi_grid_kernel.
//   * Liveness: helpers never wait, and they are the FIRST total * (P - 1) workgroups of the launch; workgroups are dispatched
//     in index order, so by the time an owner runs, its helpers run or have finished, whatever else occupies the chip -- no
//     residence condition (the row-split kernel has one), which is why this form may also be used by calls that only enqueue.
//     Should a flag not arrive within 2 ms all the same, the owner scores the candidate alone (exact path) and counts the
//     event in *timeouts: the launch heals itself, the host has nothing to redo.
//   * Tags: 0x80000000 | (host epoch + replay word) mod 2^31; the replay word lives in device memory for launches that are
//     replayed from a captured graph with frozen arguments (bumped by the graph's first node), absent otherwise.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NMI_KERNELS_DEVICE_ONLY 1
#include "nmi_kernels.hip"  // Lds, histogram_phase, decode_phase, final_phase, finish_search, exact_candidate

namespace nmi {

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// One helper's block in memory.  The counters travel as 8,192 16-byte UNITS -- but only the units that hold a count: most of
// a natural pair's joint histogram is empty, and the launch's hand-offs together (20 MB at 81 candidates x 3 ranges) otherwise
// run at the rate the memory system takes write-through stores.  A unit is what ONE LANE OF THE OWNER'S DECODE needs at once:
// unit (wave, pass, half) of lane l = the four packed words k = 4 half .. 4 half + 3 that decode_phase's lane l of that
// wave reads in that pass (nmi_kernels.hip: rows 16 wave + 4 pass + l / 16, words i + 16 k of the row), so the owner never
// reshuffles anything.  Which units came is said by 64-bit masks, one per (wave, k = 2 pass + half), and the masks double as
// the flags: each travels as two 8-byte granules {half of the mask, launch tag}, stored after the drain and the barrier, so
// a wave of the owner that finds the tag in its 16 granules has its masks AND knows the units are in memory.
constexpr int kUnits = kWords / 4;             // 8192
constexpr int kUnitsPerLane = kUnits / kBlock;  // 8
constexpr int kGranules = kWaves * kUnitsPerLane * 2;  // 256 per helper
struct PixHeader {
    unsigned long long granule[kGranules];  // [(wave * 8 + k) * 2 + half]: {mask half, tag}
    uint32_t side_key[kSide];               // the helper's flat-region side counters (fold_flat_chunk), 0 = free
    uint32_t side_cnt[kSide];
    uint32_t pad[16];
};
static_assert(sizeof(PixHeader) == 2048 + 128 && sizeof(PixHeader) % 128 == 0, "whole lines");

constexpr size_t kPixBlockBytes = sizeof(PixHeader) + (size_t)kWords * sizeof(uint32_t);
constexpr int kAuxSc1 = 16;  // cache-policy bits of the raw buffer intrinsics: sc1 (write-through store / L1-bypassing load)
constexpr int kMaxRanges = 5;  // a wave's 64 lanes poll 16 granules of each of at most 4 helpers

constexpr unsigned long long kPixTimeoutTicks = 200000ull;  // 2 ms of the 100 MHz clock; a hand-off takes microseconds

// Which pixels a workgroup adds.  The pair is cut into PIECES of 64 chunks of 16 pixels (what one wavefront loads at once: 1.6
// rows of a 640-pixel-wide frame) and the pieces are DEALT to the candidate's workgroups rather than cut into P contiguous
// ranges: flat regions (render background, the warped frame's border) and busy ones cost different time per pixel and sit in
// different parts of the frame -- with contiguous thirds the range at the bottom of the benchmark's frames took 7.9 us against
// 6.2 for the middle one, up to 10.4, and a candidate is as slow as its slowest helper.  Dealing with period L = own + (P - 1) * hlp
// pieces: the owner takes the first `own` pieces of every period, helper h the `hlp` pieces from own + (h - 1) * hlp on; own / L
// is the owner's share (the host's choice, NMI_OPT_PIX_OWNER_BIAS).  A workgroup's i-th piece is piece
// (i / cnt) * L + off + i % cnt of the frame; wavefront w takes i = 16 * iteration + w: scalar arithmetic only.
struct Deal {
    int L, off, cnt;      // period, this workgroup's first piece in a period, its pieces per period
    uint32_t magic;       // ceil(2^32 / cnt): i / cnt = umulhi(i, magic) (exact far beyond the 2^14 pieces of a 2^24-pixel frame)
    int n;                // pieces of this workgroup in the whole frame
};
// The dealing pattern of a launch, made by the host (launch_pix): the kernel does no division.
struct DealArgs {
    int own, hlp;                    // pieces per period of the owner / of each helper
    uint32_t own_magic, hlp_magic;   // ceil(2^32 / own), ceil(2^32 / hlp) (unused when the count is 1)
    int periods, rest;               // pieces of the frame = periods * L + rest, rest < L
    uint32_t total_magic;            // ceil(2^32 / candidates) (0 for one candidate): block -> (range, candidate)
};
__device__ __forceinline__ Deal make_deal(const DealArgs &g, int P, int q)
{
    Deal d;
    d.L = g.own + (P - 1) * g.hlp;
    d.off = q == 0 ? 0 : g.own + (q - 1) * g.hlp;
    d.cnt = q == 0 ? g.own : g.hlp;
    d.magic = q == 0 ? g.own_magic : g.hlp_magic;
    d.n = g.periods * d.cnt + min(max(g.rest - d.off, 0), d.cnt);
    return d;
}

template <bool SHIFTED>
__device__ __forceinline__ void histogram_dealt(Lds &lds, const GridArgs &a, const uint8_t *__restrict__ render, const uint8_t *__restrict__ warped,
                                                int wave, int lane, const Deal &d)
{
    // chunk c = the j-th 16-byte chunk of row y: byte y * width + 16 j of the frame, ry * width + 16 j of the render -- rows need not
    // be whole aligned chunks (histogram_phase's ROWS form, nmi_kernels.hip; row_rem = width % 16 pixels per row are left for
    // add_row_tails below)
    const int nchunks = a.height * a.chunks_per_row, last = nchunks - 1, row_rem = a.width - (a.chunks_per_row << 4);
    auto ldw = [&](int c) {
        c = min(c, last);
        return *reinterpret_cast<const uint4 *>(warped + (((uint32_t)c << 4) + (uint32_t)__mul24((int)__umulhi((uint32_t)c, a.cpr_magic), row_rem)));
    };
    auto ldr = [&](int c) {  // NMI.cu:82: row y of the frame meets row H-1-y of a bottom-up render
        c = min(c, last);
        const int y = (int)__umulhi((uint32_t)c, a.cpr_magic);
        const int ry = a.flip ? a.height - 1 - y : y;
        return *reinterpret_cast<const uint4 *>(render + (((uint32_t)(__mul24(y, a.flip_row) + c + a.flip_base) << 4) + (uint32_t)__mul24(ry, row_rem)));
    };
    // chunk of this lane in the workgroup's iteration `it`; beyond the workgroup's pieces: some chunk >= nchunks (not added)
    auto chunk_of = [&](int it) {
        const int i = it * kWaves + wave;  // wavefront-uniform
        const int g = d.cnt > 1 ? (int)__umulhi((uint32_t)i, d.magic) : i;
        const int t = g * d.L + d.off + (i - g * d.cnt);
        return i < d.n ? (t << 6) + lane : 0x7FFFFFC0;
    };
    const int iters = (d.n + kWaves - 1) / kWaves;  // workgroup-uniform
    if (d.off == 0 && row_rem > 0) {
        // the owner also adds the last width % 16 pixels of every row
        const int x0 = a.chunks_per_row << 4, n = a.height * row_rem;
        for (int t = wave * 64 + lane; t < n; t += kBlock) {
            const int y = t / row_rem, x = x0 + t - y * row_rem;
            uint32_t d1 = render[(a.flip ? a.height - 1 - y : y) * a.width + x], d2 = warped[y * a.width + x];
            if (SHIFTED) {
                d1 >>= a.shift;
                d2 >>= a.shift;
            }
            (void)__hip_atomic_fetch_add(&lds.joint[joint_word(d1, d2)], joint_inc(d2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (iters <= 0) return;
    const bool try_flat = !(a.phase_mask & 4);
    int resume = -1;
    int c = chunk_of(0);
    uint4 wa = ldw(c), ra = ldr(c), wb, rb;
    for (int it = 0; it < iters; it += 2) {
        const int cb = chunk_of(it + 1);
        wb = ldw(cb);
        rb = ldr(cb);
        if (__builtin_expect(flat_hint(ra, wa), 0)) {
            resume = it;
            break;
        }
        if (c < nchunks) add_chunk<true, SHIFTED, 2, false>(lds, 0, ra, wa, a.shift, false);
        c = chunk_of(it + 2);
        wa = ldw(c);
        ra = ldr(c);
        if (__builtin_expect(flat_hint(rb, wb), 0)) {
            resume = it + 1;
            break;
        }
        if (cb < nchunks) add_chunk<true, SHIFTED, 2, false>(lds, 0, rb, wb, a.shift, false);
    }
    if (resume >= 0) {
        // careful loop: same adds, flat chunks folded (fold_flat_chunk); one chunk of prefetch
        int cc = chunk_of(resume);
        uint4 wc = ldw(cc), rc = ldr(cc);
#pragma unroll 1
        for (int it = resume; it < iters; ++it) {
            const int cn = chunk_of(it + 1);
            const uint4 wn = ldw(cn), rn = ldr(cn);
            if (cc < nchunks) add_chunk<true, SHIFTED, 2, true>(lds, 0, rc, wc, a.shift, try_flat);
            wc = wn;
            rc = rn;
            cc = cn;
        }
    }
}

// LDS word k (0..7) of decode lane (i, r = DPP row, o = r & 1) in joint row d1: decode_phase's ownership (nmi_kernels.hip)
__device__ __forceinline__ uint32_t decode_word(int d1, int i, int o, int k)
{
    const uint32_t a0 = d1 * 128 + i + 16 * o;
    return k < 7 ? a0 + 16 * k : a0 + 112 - 128 * o;
}
__device__ __forceinline__ int unit_offset(int wave, int kk, int lane) { return (int)sizeof(PixHeader) + ((wave * kUnitsPerLane + kk) * 64 + lane) * 16; }

// The owner's decode: decode_phase (ComputeEntropyKernel + AddvectorParwiseMidKernel, NMI.cu:230-287) over its own packed
// counters PLUS the helpers' (acc: their units of this lane, already summed field by field), with two differences: counters
// are not cleared (the workgroup scores one candidate) and there are no wrap events to replay (nobody used returning atomics).
template <bool ZERO0>
__device__ __forceinline__ void decode_merged(Lds &lds, const GridArgs &a, int wave, int lane, const u32x4 (&acc)[kUnitsPerLane])
{
    const bool side_any = lds.side_key[0][0] != 0u;
    uint32_t wave_total = 0;
    const int i = lane & 15, r = lane >> 4, o = r & 1;
    uint32_t col_lo[8], col_hi[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) col_lo[k] = col_hi[k] = 0;
#pragma unroll
    for (int pass = 0; pass < kRowsPerWave / 4; ++pass) {
        const int d1 = wave * kRowsPerWave + pass * 4 + r;
        uint32_t lo[8], hi[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t wd = lds.joint[decode_word(d1, i, o, k)];
            const uint32_t ad = acc[pass * 2 + (k >> 2)][k & 3];
            lo[k] = (wd & 0xFFFFu) + (ad & 0xFFFFu);
            hi[k] = (wd >> 16) + (ad >> 16);
        }
        if (__builtin_expect(side_any, 0)) {
            // side counters of flat regions (fold_flat_chunk): entries fill in order, a free one ends the list
            for (int e = 0; e < kSide; ++e) {
                const uint32_t key1 = __builtin_amdgcn_readfirstlane(lds.side_key[0][e]);
                if (key1 == 0u) break;
                const uint32_t sword = (key1 - 1u) >> 1;
                if ((sword >> 9) != (uint32_t)((wave * kRowsPerWave + pass * 4) >> 2)) continue;  // not among this pass's 4 rows
                const uint32_t cnt = lds.side_cnt[0][e];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (decode_word(d1, i, o, k) == sword) {
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
            tl[k] = lds.table[lo[k] & (kLdsTable - 1)];
            th[k] = lds.table[hi[k] & (kLdsTable - 1)];
        }
        if (__builtin_expect(cmax >= (uint32_t)kLdsTable, 0)) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (lo[k] >= (uint32_t)kLdsTable) tl[k] = a.table[min(lo[k], (uint32_t)a.npix)];  // (a wrapped helper field can read high; the detector rejects the candidate)
                if (hi[k] >= (uint32_t)kLdsTable) th[k] = a.table[min(hi[k], (uint32_t)a.npix)];
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
    if (i == 0) atomicAdd(&lds.total[0], wave_total);
}

// final_phase (nmi_kernels.hip: the three 256-element trees of AddVectorPairwiseKernel, NMI.cu:295-339, side by side in DPP rows
// 0..2, then the score) with one difference: the marginal counts' terms come from the LDS copy of the table where the count is
// below its 4096 entries (most of a 640x480 frame's 256 marginal bins are) and from memory only above -- the owner scores ONE
// candidate, so the memory round trip of the lookups is on every launch's critical path instead of hidden behind the next
// candidate's pixels.  Same values (the LDS table is a copy), same order.
__device__ __forceinline__ void final_phase_owner(Lds &lds, const GridArgs &a, int lane, int p, int w, int s, unsigned long long &prev_key)
{
    const int i = lane & 15, r = lane >> 4;
    float lo[8], hi[8];
    const uint32_t *h = r == 0 ? lds.hist_render : lds.hist_warped;
    uint32_t cl[8], ch[8], cmax = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        cl[k] = r < 2 ? h[i + 16 * k] : 0u;
        ch[k] = r < 2 ? h[i + 16 * k + 128] : 0u;
        cmax = max(cmax, max(cl[k], ch[k]));
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        lo[k] = lds.table[cl[k] & (kLdsTable - 1)];
        hi[k] = lds.table[ch[k] & (kLdsTable - 1)];
    }
    if (cmax >= (uint32_t)kLdsTable) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (cl[k] >= (uint32_t)kLdsTable) lo[k] = a.table[cl[k]];
            if (ch[k] >= (uint32_t)kLdsTable) hi[k] = a.table[ch[k]];
        }
    }
    if ((w == 0 || s == 0) && a.plan) {  // the search as its own content probe (final_phase)
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

}  // namespace

size_t pix_block_bytes(int candidates, int pix_parts) { return (size_t)candidates * (size_t)(pix_parts - 1) * kPixBlockBytes; }
int pix_max_ranges() { return kMaxRanges; }

template <bool ZERO0, bool SHIFTED>
__global__ __launch_bounds__(NMI_BLOCK_THREADS) void nmi_pix_kernel(GridArgs a, int P, DealArgs dealing, const uint32_t *replay, uint32_t *timeouts)
{
    __shared__ Lds lds;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    auto stamp = [&](int k) {  // tools/pix_stamps.py: where a workgroup's time goes (100 MHz clock)
        if (a.dbg_stamps && tid == 0) a.dbg_stamps[blockIdx.x * 8 + k] = wall_clock64();
    };
    stamp(0);

    // unit of this workgroup: helpers first (see Liveness above)
    const int total = a.S_local * a.Wn;
    const int helpers = total * (P - 1);
    const int b = (int)blockIdx.x;
    const int q = b < helpers ? 1 + (total > 1 ? (int)__umulhi((uint32_t)b, dealing.total_magic) : b) : 0;
    const int p = b < helpers ? b - (q - 1) * total : b - helpers;
    const bool owner = q == 0;
    const int w = p / a.S_local, s = p - w * a.S_local;
    const uint8_t *render = a.render_stack + (size_t)s * a.npix;
    const uint8_t *warped = a.warp_stack + (size_t)w * a.npix;
    // never 0 (the state of fresh memory); the replay word counts the replays of a captured graph, whose arguments are frozen
    const uint32_t tag = 0x80000000u | ((a.epoch + (replay ? __hip_atomic_load(replay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u)) & 0x7FFFFFFFu);

    if (b == 0 && tid == 0 && a.reset_key) *a.reset_key = 0ull;  // next launch's slot; idle during this one
    float tab[kLdsTable / kBlock];
    if (owner) {
#pragma unroll
        for (int k = 0; k < kLdsTable / kBlock; ++k) {
            const int c = tid + k * kBlock;
            tab[k] = a.table[c <= a.npix ? c : 0];
        }
    }
    {
        uint4 *j4 = reinterpret_cast<uint4 *>(lds.joint);
        const uint4 z = {0, 0, 0, 0};
        for (int i = tid; i < kWords / 4; i += kBlock) j4[i] = z;
    }
    if (tid < kBins) lds.hist_warped[tid] = 0;
    if (tid < 2) lds.ovf_n[tid] = lds.total[tid] = 0;
    if (tid < 2 * kSide) (&lds.side_key[0][0])[tid] = (&lds.side_cnt[0][0])[tid] = 0;
    if (tid == 0) lds.fallback = 0;
    // This workgroup's pixels (frames of at least 32 pixels of width).  The owner's share is the larger one: its
    // helpers' counters need a few microseconds to reach it, which it spends adding pixels.
    const Deal deal = make_deal(dealing, P, q);
    __syncthreads();
    stamp(1);
    histogram_dealt<SHIFTED>(lds, a, render, warped, wave, lane, deal);

    char *const blocks = reinterpret_cast<char *>(a.blocks) + (size_t)p * (size_t)(P - 1) * kPixBlockBytes;
    if (!owner) {
        // ---- helper: units that hold a count -> memory, write-through; drain; barrier; tagged masks ----
        __syncthreads();
        stamp(2);
        char *const blk = blocks + (size_t)(q - 1) * kPixBlockBytes;
        PixHeader *const hdr = reinterpret_cast<PixHeader *>(blk);
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(blk, 0, (int)kPixBlockBytes, 0x00020000);
        unsigned long long mask[kUnitsPerLane];
        {
            const int i = lane & 15, r = lane >> 4, o = r & 1;
#pragma unroll
            for (int kk = 0; kk < kUnitsPerLane; ++kk) {
                const int d1 = wave * kRowsPerWave + (kk >> 1) * 4 + r;
                u32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = lds.joint[decode_word(d1, i, o, (kk & 1) * 4 + j)];
                const bool on = (v.x | v.y | v.z | v.w) != 0u;
                mask[kk] = __ballot(on);
                if (on) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, unit_offset(wave, kk, lane), 0, kAuxSc1);
            }
        }
        if (tid < kSide) {
            __hip_atomic_store(&hdr->side_key[tid], lds.side_key[0][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&hdr->side_cnt[tid], lds.side_cnt[0][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        uint32_t half = 0;
#pragma unroll
        for (int g = 0; g < 2 * kUnitsPerLane; ++g)
            if (lane == g) half = (uint32_t)(mask[g >> 1] >> (32 * (g & 1)));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave, before the barrier the granules' lanes wait at
        __syncthreads();
        // (phase mask bit 9, tests only: helper 1 keeps its masks to itself, so its owner's wait must time out)
        if (lane < 2 * kUnitsPerLane && !((a.phase_mask & 512) && q == 1))
            __hip_atomic_store(&hdr->granule[wave * 2 * kUnitsPerLane + lane], ((unsigned long long)tag << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stamp(3);
        return;
    }

    // ---- owner ----
#pragma unroll
    for (int k = 0; k < kLdsTable / kBlock; ++k) lds.table[tid + k * kBlock] = tab[k];
    stamp(2);
    // every wave polls for itself: lane 16 h + g the granule g of helper h + 1 that belongs to this wave's units
    unsigned long long gv = 0;
    bool seen = true;
    if (lane < 16 * (P - 1)) {
        const unsigned long long *g = reinterpret_cast<const PixHeader *>(blocks + (size_t)(lane >> 4) * kPixBlockBytes)->granule + wave * 16 + (lane & 15);
        unsigned long long t0 = 0;
        int tries = 0;
        while ((uint32_t)((gv = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != tag) {
            __builtin_amdgcn_s_sleep(4);
            if ((++tries & 15) == 1) {
                const unsigned long long now = wall_clock64();
                if (tries == 1) t0 = now;
                if (now - t0 > kPixTimeoutTicks || tries > (1 << 20)) {
                    seen = false;
                    break;
                }
            }
        }
    }
    seen = __all(seen);  // wave-uniform
    if (!seen && lane == 0) lds.fallback = 1;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // (no instruction: keeps the loads below behind the poll)
    // This lane's units of every helper, summed field by field as packed words (a field that wraps in the sum loses weight like
    // any other wrap).  Issued before the barrier: they arrive while the slower wavefronts finish their pixels.
    const uint32_t gh = (uint32_t)gv;  // this lane's mask half
    u32x4 acc[kUnitsPerLane];
#pragma unroll
    for (int kk = 0; kk < kUnitsPerLane; ++kk) acc[kk] = u32x4{0, 0, 0, 0};
    if (seen) {
        for (int h = 0; h < P - 1; ++h) {
            const char *blk = blocks + (size_t)h * kPixBlockBytes;
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(blk), 0, (int)kPixBlockBytes, 0x00020000);
            u32x4 v[kUnitsPerLane];
#pragma unroll
            for (int kk = 0; kk < kUnitsPerLane; ++kk) {
                const uint32_t lo = __builtin_amdgcn_readlane(gh, h * 16 + 2 * kk), hi = __builtin_amdgcn_readlane(gh, h * 16 + 2 * kk + 1);
                v[kk] = u32x4{0, 0, 0, 0};
                if ((((((unsigned long long)hi << 32) | lo) >> lane) & 1ull) != 0ull) v[kk] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, unit_offset(wave, kk, lane), 0, kAuxSc1);
            }
            if (wave == 0 && lane < kSide) {
                // a helper's side counter: into a side counter of the owner's (atomics: other wavefronts may still be folding flat
                // chunks of their own), or, when those are taken, onto the packed field (which may wrap it: see above)
                const uint32_t skey = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)offsetof(PixHeader, side_key) + lane * 4, 0, kAuxSc1);
                const uint32_t scnt = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)offsetof(PixHeader, side_cnt) + lane * 4, 0, kAuxSc1);
                if (skey != 0u) {
                    const uint32_t word = (skey - 1u) >> 1, high = (skey - 1u) & 1u;
                    if (!side_add(lds, 0, word, high, scnt)) atomicAdd(&lds.joint[word], high ? scnt << 16 : scnt);
                }
            }
#pragma unroll
            for (int kk = 0; kk < kUnitsPerLane; ++kk) acc[kk] += v[kk];
        }
    }
    __syncthreads();  // B1: every wavefront's pixels are in the counters, every helper's side counters in the list
    stamp(3);
    unsigned long long prev_key = 0;
    bool alone = lds.fallback != 0;  // some wave gave up on a helper (workgroup-uniform)
    if (!alone) {
        decode_merged<ZERO0>(lds, a, wave, lane, acc);
        __syncthreads();
        stamp(5);
        alone = lds.total[0] != (uint32_t)a.npix;  // some 16-bit field wrapped (workgroup-uniform, rare)
        if (!alone && wave == 0) final_phase_owner(lds, a, lane, p, w, s, prev_key);
    } else if (tid == 0 && timeouts) {
        __hip_atomic_fetch_add(timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (alone) {
        // cold: this candidate once more, by this workgroup alone, on the exact path
        __syncthreads();
        {
            uint4 *j4 = reinterpret_cast<uint4 *>(lds.joint);
            const uint4 z = {0, 0, 0, 0};
            for (int i = tid; i < kWords / 4; i += kBlock) j4[i] = z;
        }
        if (tid < kBins) lds.hist_warped[tid] = 0;
        if (tid < 2) lds.total[tid] = lds.ovf_n[tid] = 0;
        if (tid < 2 * kSide) (&lds.side_key[0][0])[tid] = (&lds.side_cnt[0][0])[tid] = 0;
        __syncthreads();
        exact_candidate<SHIFTED, !ZERO0>(lds, a, tid, p, prev_key);
    }
    stamp(6);
    if (wave == 0) finish_search(a, lane, prev_key, (uint32_t)total);
    stamp(7);
}

// One launch of total * pix_parts workgroups.  Needs the 16-byte path (a.vec_ok), 256 bins or the background rule on, and
// a.blocks of pix_block_bytes(total, pix_parts), zero when allocated.  owner_share: the fraction of the pair's pixels the
// owner adds itself (1 / pix_parts: equal shares).
hipError_t launch_pix(const GridArgs &a, int pix_parts, double owner_share, bool use_bg, const uint32_t *replay, uint32_t *timeouts, hipStream_t stream)
{
    const long long total = (long long)a.S_local * a.Wn;
    if (pix_parts < 2 || pix_parts > kMaxRanges || total <= 0 || total * pix_parts > 0x7FFFFFFFll) return hipErrorInvalidValue;
    if (a.width < 32 || !a.blocks || a.hist_variant != 3 || (a.shift != 0 && !use_bg) || a.order) return hipErrorInvalidValue;
    // the dealing pattern: own : hlp pieces per period, the closest to the wanted share among periods of at most 48 pieces
    int own = 1, hlp = 1;
    {
        const double f = owner_share < 0.02 ? 0.02 : (owner_share > 0.98 ? 0.98 : owner_share);
        double best = 2.0;
        for (int b = 1; b <= 12; ++b) {
            int o = (int)(f / (1.0 - f) * (pix_parts - 1) * b + 0.5);
            o = o < 1 ? 1 : o;
            if (o + (pix_parts - 1) * b > 48) break;
            const double err = fabs((double)o / (o + (pix_parts - 1) * b) - f);
            constexpr double kCloser = 0.03;
            if (err < best - kCloser) best = err, own = o, hlp = b;  // a longer period has to be clearly closer: dealt in runs of 5 pieces
                                                                      // (8 rows) one helper was 16 % slower than the other on the benchmark's frames
        }
    }
    auto magic = [](int d) { return d > 1 ? (uint32_t)((0x100000000ull + (uint32_t)d - 1) / (uint32_t)d) : 0u; };
    const int pieces = (a.height * a.chunks_per_row + 63) >> 6, L = own + (pix_parts - 1) * hlp;
    const DealArgs g{own, hlp, magic(own), magic(hlp), pieces / L, pieces % L, magic((int)total)};
    const dim3 grid((unsigned)(total * pix_parts)), block(kBlock);
    if (a.shift != 0)
        hipLaunchKernelGGL((nmi_pix_kernel<false, true>), grid, block, 0, stream, a, pix_parts, g, replay, timeouts);
    else if (use_bg)
        hipLaunchKernelGGL((nmi_pix_kernel<false, false>), grid, block, 0, stream, a, pix_parts, g, replay, timeouts);
    else
        hipLaunchKernelGGL((nmi_pix_kernel<true, false>), grid, block, 0, stream, a, pix_parts, g, replay, timeouts);
    return hipGetLastError();
}

}  // namespace nmi
