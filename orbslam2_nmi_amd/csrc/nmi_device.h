// nmi_device.h -- device-side helpers shared by the scoring kernels (nmi_kernels.hip: one workgroup per candidate;
// nmi_split_kernel.hip: K workgroups per candidate): the reference's stride-halving trees as DPP / in-lane adds, the
// score formulas, the packed arg-max key and the launch-completion protocol.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nmi_kernels.h"

namespace nmi {
namespace {

// ---- cross-lane helpers (DPP within a row of 16 lanes: lane i receives lane i + N) -------------------
template <int N>
__device__ __forceinline__ float row_shl(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x100 + N, 0xF, 0xF, true));
}
template <int N>
__device__ __forceinline__ uint32_t row_shl(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x100 + N, 0xF, 0xF, true);
}
// Tree steps n = 8,4,2,1 over one value per lane of a 16-lane row: lane t < n takes a[t] += a[t + n];
// the sum ends in lane 0 of the row.
__device__ __forceinline__ float row_tree_16(float x)
{
    x = x + row_shl<8>(x);
    x = x + row_shl<4>(x);
    x = x + row_shl<2>(x);
    x = x + row_shl<1>(x);
    return x;
}
__device__ __forceinline__ uint32_t row_sum_16(uint32_t x)
{
    x += row_shl<8>(x);
    x += row_shl<4>(x);
    x += row_shl<2>(x);
    x += row_shl<1>(x);
    return x;
}
// In-lane part of the 256-element stride-halving tree for a lane that owns elements
// t = i + 16*j (j = 0..15, any rotation of j): steps n = 128, 64, 32, 16 pair j with j + n/16.
__device__ __forceinline__ float lane_tree_16(const float (&lo)[8], const float (&hi)[8])
{
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = lo[k] + hi[k];  // n = 128
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = s[k] + s[k + 4];  // n = 64
    s[0] = s[0] + s[2];                                  // n = 32
    s[1] = s[1] + s[3];
    return s[0] + s[1];                                  // n = 16
}

// Cheap necessary condition for a flat chunk, wavefront-uniform: first dword == last dword in both images for every
// active lane (2 VALU compares + scalar work); the full test runs only where this holds.
__device__ __forceinline__ bool flat_hint(const uint4 &rv, const uint4 &wv)
{
    // spelled out: from "__all(...)" hipcc builds compare, select 0/1, compare again, two scalar tests
    unsigned long long m;
    asm volatile("v_cmp_eq_u32 vcc, %1, %2\n\tv_cmp_eq_u32 %0, %3, %4\n\ts_and_b64 %0, %0, vcc"
                 : "=s"(m)
                 : "v"(rv.x), "v"(rv.w), "v"(wv.x), "v"(wv.w)
                 : "vcc");
    return m == __builtin_amdgcn_read_exec();
}

// Score from the three completed (<= 0) entropy sums, NMI.cu:342-362 (the reference reads them across blocks without
// synchronisation, NMI.cu:340-342; this is the intended value), then: rating store, debug sums, and the arg-max update.
// find_max_elements, helperFunctions.cpp:52-101: max starts at 0, strict '>', first cell equal to the max wins.
// Non-negative floats order like their bit patterns, so one 64-bit max of (score bits, inverted global index)
// reproduces it; negative / NaN scores contribute nothing.  One lane calls this per candidate.
__device__ __forceinline__ void commit_score(const GridArgs &a, int p, int w, int s, float a1, float a2, float a3,
                                             unsigned long long &prev_key)
{
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
    if (a.score_post)  // nmi_eval_pair: the score itself (any sign, NaN included) + the call's sequence number, one store
        __hip_atomic_store(a.score_post, (unsigned long long)__float_as_uint(score) | ((unsigned long long)a.seq << 32),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (score >= 0.0f) {
        const uint32_t bits = score == 0.0f ? 0u : __float_as_uint(score);
        const uint32_t gidx = (uint32_t)(a.w_offset + w) * (uint32_t)a.S_total + (uint32_t)(a.s_offset + s);
        const unsigned long long key = ((unsigned long long)bits << 32) | (unsigned long long)(0xFFFFFFFFu - gidx);
        // returning form: the value is not needed, but its arrival (awaited once, at kernel end) proves the
        // max was performed, which the completion protocol below builds on
        prev_key = __hip_atomic_fetch_max(a.key, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Completion, called by lane 0 of each workgroup when it has scored its last candidate: the last workgroup to finish
// publishes the winner.  Every atomicMax of final_phase and the counter below are device-scope read-modify-writes
// performed at the memory side.  The counter increment carries a data dependency on the values returned by this
// workgroup's maxes, so it is issued after they were performed; the workgroup that draws the last ticket therefore
// reads the final key.  Nothing here needs a cache write-back: the key travels in atomics, and the mailbox is one
// 8-byte store (key in bits 0..62, launch parity in bit 63 -- scores are non-negative floats, bit 63 is free).
// Returns true in the workgroup that drew the last ticket.
__device__ __forceinline__ bool publish_winner(const GridArgs &a, unsigned long long prev_key)
{
    const unsigned int one = prev_key == ~0ull ? 2u : 1u;  // always 1 (a key never has all bits set)
    const unsigned int arrived = __hip_atomic_fetch_add(a.done, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (arrived == gridDim.x - 1) {
        const unsigned long long final_key = __hip_atomic_load(a.key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a.out_key)  // may be pinned host memory that a caller polls (nmi_level_run)
            __hip_atomic_store(a.out_key, final_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (a.mailbox)
            __hip_atomic_store(&a.mailbox->word, final_key | ((unsigned long long)(a.seq & 1u) << 63), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        return true;
    }
    return false;
}

}  // namespace
}  // namespace nmi
