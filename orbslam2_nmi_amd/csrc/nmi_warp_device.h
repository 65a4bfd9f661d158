// nmi_warp_device.h -- device code of the warp-stack producer's LDS-staged form, shared by its own kernel
// (nmi_producers.hip) and by the render kernels that carry the warp blocks of a captured level along (nmi_producers.hip:
// point splat; nmi_mesh.hip: triangle binning) -- a level's graph is then one chain of kernels, without the fork and the
// join (~15 us of hand-overs between queues) a warp kernel on a branch of its own costs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nmi {
namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

// 1.0f / d, correctly rounded, in 4 instructions instead of the 11 of the compiler's division: the hardware's approximation
// (1 ulp) and one Newton step in two FMAs.  Checked against the division for EVERY float on gfx950 (tests/test_render.py,
// test_gpu_reciprocal_exhaustive): identical for all d with exponent field in [3, 252] (warp_rcp_ok); everything else --
// zero, denormal, huge, inf, NaN -- takes the division.
__device__ __forceinline__ bool warp_rcp_ok(float d) { return ((__float_as_uint(d) >> 23) & 0xFFu) - 3u <= 249u; }
__device__ __forceinline__ float warp_rcp_fast(float d)
{
    const float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float warp_rcp(float d) { return __builtin_expect(warp_rcp_ok(d), 1) ? warp_rcp_fast(d) : 1.0f / d; }

__device__ __forceinline__ float warp_tap(const uint8_t *__restrict__ src, int w, int h, int x, int y)
{
    return (x >= 0 && x < w && y >= 0 && y < h) ? (float)src[y * w + x] : 0.0f;  // BORDER_CONSTANT, value 0
}

// One output pixel with its taps taken from global memory (the arithmetic of nmi_warp_kernel, shared with the fallback of
// the staged kernel below).
__device__ __forceinline__ uint32_t warp_pixel_global(const uint8_t *__restrict__ frame, const float *__restrict__ c, int width, int height,
                                                      int x, int y)
{
    const float fx = (float)x, fy = (float)y;
    const float coeff = 1.0f / (c[6] * fx + c[7] * fy + c[8]);
    const float xs = coeff * (c[0] * fx + c[1] * fy + c[2]);
    const float ys = coeff * (c[3] * fx + c[4] * fy + c[5]);
    float acc = 0.0f;
    if (xs > -2.0f && xs < (float)(width + 1) && ys > -2.0f && ys < (float)(height + 1)) {
        const int x1 = (int)floorf(xs), y1 = (int)floorf(ys);
        const int x2 = x1 + 1, y2 = y1 + 1;
        const float t11 = warp_tap(frame, width, height, x1, y1), t21 = warp_tap(frame, width, height, x2, y1);
        const float t12 = warp_tap(frame, width, height, x1, y2), t22 = warp_tap(frame, width, height, x2, y2);
        acc = acc + t11 * (((float)x2 - xs) * ((float)y2 - ys));
        acc = acc + t21 * ((xs - (float)x1) * ((float)y2 - ys));
        acc = acc + t12 * (((float)x2 - xs) * (ys - (float)y1));
        acc = acc + t22 * ((xs - (float)x1) * (ys - (float)y1));
    }
    const float r = rintf(acc);
    return r <= 0.0f ? 0u : (r >= 255.0f ? 255u : (uint32_t)r);
}

// The same warp with the source patch of each block staged in LDS.  nmi_warp_kernel spends its time on scattered tap
// loads and the border tests around them (two 2-byte global loads per pixel with 64 different addresses per wave
// instruction, four-way branching per tap near the border: 38 us for 27 warps at 848x480 = 0.58 TB/s of algorithmic
// traffic).  A block's 128 x 32 output pixels take their taps from the image of that rectangle under the (inverse)
// homography -- a convex quadrilateral, hence inside the bounding box of its four warped corners; for the grid's
// rotations a patch of ~160 x 50 pixels.  The block fetches that patch once with coalesced 16-byte loads (frame rows
// start on 16-byte boundaries: the width is a multiple of 16 here) INCLUDING a border of zeros wherever the box (grown by
// the 2 pixels a tap can reach beyond the frame) sticks out of the frame, so that BORDER_CONSTANT(0) needs no test: every
// tap of every pixel is one unconditional LDS byte read, and neighbouring lanes read neighbouring bytes of one dword,
// which the LDS serves as a broadcast.  Arithmetic and its order are exactly nmi_warp_kernel's (the same fp32 twin
// tests both).  Blocks whose patch exceeds the LDS budget, whose corners are not finite or whose homogeneous
// coordinate is not positive at every corner take the global-tap form wholesale.
// 16 KiB, not more: a kernel's LDS is reserved for EVERY workgroup of a launch, also for the splat / triangle workgroups that ride
// in the same launch and never touch it, and with 24 KiB a CU held 6 workgroups instead of the 8 its wave slots allow -- five of
// them the long-lived splat workgroups waiting on their atomics, one left for everything else (front kernel of the e2e bench's
// level: 59.1 -> 56.7 us with 8).  The grid's rotations need ~7 KiB; a larger patch takes the global-tap form.
constexpr int kWarpPatchBytes = 16 * 1024 - 64;
constexpr int kWarpRowsPerThread = 4;  // a block covers 128 x 32 output pixels: one patch fetch per 4096 pixels

// One block of 256 lanes (tx = 0..31, ty = 0..7) = 128 x 32 output pixels of warp wi: block column bx, block row by.
__device__ __forceinline__ void warp_lds_block(const uint8_t *__restrict__ frame, const float *__restrict__ coeffs, uint8_t *__restrict__ out,
                                               int width, int height, int quads_per_row, int bx, int by, int wi, int tx, int ty)
{
    __shared__ __attribute__((aligned(16))) uint8_t patch[kWarpPatchBytes + 16];  // + slack for the 8-byte tap windows
    __shared__ int box[4];  // patch origin x (multiple of 16, may be -16), origin y (may be negative), pitch in bytes (0 = no patch), rows
    constexpr int kBlockRows = 8 * kWarpRowsPerThread;
    const int tid = ty * 32 + tx;
    const int q = bx * 32 + tx;
    const float *c = coeffs + wi * 9;
    if (tid < 64) {
        // lanes 0..3: the four corners of this block's pixel rectangle, through the same fp32 expressions as the pixels
        const int bx0 = bx * 128, by0 = by * kBlockRows;
        const int bx1 = min(bx0 + 127, width - 1), by1 = min(by0 + kBlockRows - 1, height - 1);
        const float fx = (float)((tid & 1) ? bx1 : bx0), fy = (float)((tid & 2) ? by1 : by0);
        const float den = c[6] * fx + c[7] * fy + c[8];
        const float coeff = 1.0f / den;
        const float xs = coeff * (c[0] * fx + c[1] * fy + c[2]);
        const float ys = coeff * (c[3] * fx + c[4] * fy + c[5]);
        const bool good = den > 0.0f && fabsf(xs) < 1e8f && fabsf(ys) < 1e8f;  // also false for NaN
        float xlo = xs, xhi = xs, ylo = ys, yhi = ys;
        bool all_good = good;
#pragma unroll
        for (int off = 1; off < 4; off <<= 1) {
            xlo = fminf(xlo, __shfl_xor(xlo, off, 64));
            xhi = fmaxf(xhi, __shfl_xor(xhi, off, 64));
            ylo = fminf(ylo, __shfl_xor(ylo, off, 64));
            yhi = fmaxf(yhi, __shfl_xor(yhi, off, 64));
            all_good = all_good && __shfl_xor((int)all_good, off, 64) != 0;
        }
        if (tid == 0) {
            int pitch = 0, rows = 0, px0 = 0, py0 = 0;
            if (all_good) {
                // taps of pixels that pass the range test lie in [-2, width + 1] x [-2, height + 1]; 2 pixels of margin
                // around the corners' box absorb floor / +1 and the rounding of the transform
                const int x_lo = max((int)floorf(xlo) - 2, -2), x_hi = min((int)floorf(xhi) + 3, width + 1);
                const int y_lo = max((int)floorf(ylo) - 2, -2), y_hi = min((int)floorf(yhi) + 3, height + 1);
                if (x_lo <= x_hi && y_lo <= y_hi) {
                    px0 = x_lo < 0 ? -16 : (x_lo & ~15);
                    py0 = y_lo;
                    pitch = ((x_hi - px0 + 1) + 15) & ~15;
                    rows = y_hi - y_lo + 1;
                    if (pitch * rows > kWarpPatchBytes) pitch = 0, rows = 0;  // too large: global taps
                } else {
                    px0 = -16, py0 = -2, pitch = 16, rows = 2;  // nothing in reach: two border rows of zeros serve every (clamped) tap
                }
            }
            box[0] = px0, box[1] = py0, box[2] = pitch, box[3] = rows;
        }
    }
    __syncthreads();
    const int px0 = box[0], py0 = box[1], pitch = box[2], rows = box[3];
    if (pitch == 0) {  // block-uniform fallback
        if (q >= quads_per_row) return;
        for (int rr = 0; rr < kWarpRowsPerThread; ++rr) {
            const int y = by * kBlockRows + rr * 8 + ty;
            if (y >= height) break;
            uint32_t packed = 0;
            for (int k = 0; k < 4; ++k) packed |= warp_pixel_global(frame, c, width, height, q * 4 + k, y) << (8 * k);
            *reinterpret_cast<uint32_t *>(out + ((size_t)wi * height + y) * width + q * 4) = packed;
        }
        return;
    }
    {
        const int units_per_row = pitch >> 4, units = units_per_row * rows;
        const uint4 zero = {0, 0, 0, 0};
        for (int u = tid; u < units; u += 256) {
            const int r = u / units_per_row, cx = u - r * units_per_row;
            const int fy = py0 + r, fx = px0 + cx * 16;  // a 16-byte unit is wholly inside or wholly outside the frame
            uint4 v = zero;
            if (fy >= 0 && fy < height && fx >= 0 && fx < width) v = *reinterpret_cast<const uint4 *>(frame + (size_t)fy * width + fx);
            *reinterpret_cast<uint4 *>(patch + r * pitch + cx * 16) = v;
        }
    }
    __syncthreads();
    if (q >= quads_per_row) return;
    // The loop below is bound by vector instruction issue (a wave instruction occupies its SIMD for 4 cycles: 0.6 T wave
    // instructions/s for the whole chip; 68 instructions per pixel made this kernel 19 us for 11 M pixels).  Hence: pixels go
    // in pairs through the packed fp32 instructions (v_pk_mul / v_pk_add: IEEE results per element, and never fused -- this
    // file is built with -ffp-contract=off), and 1 / den comes from warp_rcp instead of the 11-instruction division.
    const v2f c0 = {c[0], c[0]}, c3 = {c[3], c[3]}, c6 = {c[6], c[6]};
    const float c1 = c[1], c2 = c[2], c4 = c[4], c5 = c[5], c7 = c[7], c8 = c[8];
    const float xmax = (float)(width + 1), ymax = (float)(height + 1);
    const v2f fxa = {(float)(q * 4), (float)(q * 4 + 1)}, fxb = {(float)(q * 4 + 2), (float)(q * 4 + 3)};
    const v2f c0xa = c0 * fxa, c0xb = c0 * fxb, c3xa = c3 * fxa, c3xb = c3 * fxb, c6xa = c6 * fxa, c6xb = c6 * fxb;  // the same for every row
    const uint32_t lx_max = (uint32_t)(pitch - 2), ly_max = (uint32_t)(rows - 2);
#pragma unroll 1
    for (int rr = 0; rr < kWarpRowsPerThread; ++rr) {
        const int y = by * kBlockRows + rr * 8 + ty;
        if (y >= height) break;
        const float fy = (float)y;
        const float c1y = c1 * fy, c4y = c4 * fy, c7y = c7 * fy;
        uint32_t packed = 0;
        // coeff = 1 / ((c6 x + c7 y) + c8) of the four pixels: one range test (and one branch) for all of them
        const v2f den_a = (c6xa + c7y) + c8, den_b = (c6xb + c7y) + c8;
        v2f coeff_a, coeff_b;
        if (__builtin_expect(warp_rcp_ok(den_a.x) & warp_rcp_ok(den_a.y) & warp_rcp_ok(den_b.x) & warp_rcp_ok(den_b.y), 1)) {
            coeff_a = v2f{warp_rcp_fast(den_a.x), warp_rcp_fast(den_a.y)}, coeff_b = v2f{warp_rcp_fast(den_b.x), warp_rcp_fast(den_b.y)};
        } else {
            coeff_a = v2f{1.0f / den_a.x, 1.0f / den_a.y}, coeff_b = v2f{1.0f / den_b.x, 1.0f / den_b.y};
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // pixel pair (4q + 2 half, 4q + 2 half + 1): xs = coeff ((c0 x + c1 y) + c2), ys = coeff ((c3 x + c4 y) + c5)
            const v2f coeff = half ? coeff_b : coeff_a;
            const v2f xs = coeff * (((half ? c0xb : c0xa) + c1y) + c2);
            const v2f ys = coeff * (((half ? c3xb : c3xa) + c4y) + c5);
            // nmi_warp_kernel gives 0 to pixels whose source lies outside (-2, width + 1) x (-2, height + 1).  Clamping the
            // source coordinate into that closed range does the same without a test: a clamped coordinate has both of its
            // taps (or its whole 2 x 2 window) in the zero border, and it stays inside this block's patch because the patch
            // is the corners' bounding box clipped to the very same range.
            const v2f xsc = {__builtin_amdgcn_fmed3f(xs.x, -2.0f, xmax), __builtin_amdgcn_fmed3f(xs.y, -2.0f, xmax)};
            const v2f ysc = {__builtin_amdgcn_fmed3f(ys.x, -2.0f, ymax), __builtin_amdgcn_fmed3f(ys.y, -2.0f, ymax)};
            const v2f x1f = {floorf(xsc.x), floorf(xsc.y)}, y1f = {floorf(ysc.x), floorf(ysc.y)};
            v2f t11, t21, t12, t22;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                // (unsigned min: a negative offset -- impossible while the bounding-box argument holds -- also ends up inside)
                const uint32_t lx = min((uint32_t)((int)x1f[e] - px0), lx_max), ly = min((uint32_t)((int)y1f[e] - py0), ly_max);
                // The two taps of a row are bytes o, o + 1 of the patch with o of any alignment.  An unaligned 2-byte LDS read
                // costs ~200 cycles of issue stall (measured: SQ_WAIT_INST_LDS 55 units per ds_read_u16, the whole kernel 40 us
                // however its taps were fetched), so the 8 aligned bytes around them are read and shifted instead.
                const uint32_t o = ly * (uint32_t)pitch + lx;
                const uint32_t *w0 = reinterpret_cast<const uint32_t *>(patch + (o & ~3u));
                const uint32_t *w1 = reinterpret_cast<const uint32_t *>(patch + (o & ~3u) + pitch);
                const uint32_t top = __builtin_amdgcn_alignbyte(w0[1], w0[0], o & 3u);
                const uint32_t bot = __builtin_amdgcn_alignbyte(w1[1], w1[0], o & 3u);
                t11[e] = (float)(top & 0xFFu), t21[e] = (float)((top >> 8) & 0xFFu);
                t12[e] = (float)(bot & 0xFFu), t22[e] = (float)((bot >> 8) & 0xFFu);
            }
            const v2f x2f = x1f + 1.0f, y2f = y1f + 1.0f;  // exact: small integers
            const v2f ax = x2f - xsc, bx = xsc - x1f, ay = y2f - ysc, bw = ysc - y1f;
            // 0 + t11 w11 + t21 w21 + t12 w12 + t22 w22 in nmi_warp_kernel's order; its leading "0 +" changes nothing here
            // (taps and weights are >= +0, so the first product is never -0)
            v2f acc = t11 * (ax * ay);
            acc = acc + t21 * (bx * ay);
            acc = acc + t12 * (ax * bw);
            acc = acc + t22 * (bx * bw);
            // saturate_cast<uchar>: round to nearest even, clamp; acc >= 0, and the pack instruction saturates at 255
            packed = __builtin_amdgcn_cvt_pk_u8_f32(rintf(acc.x), 2 * half, packed);
            packed = __builtin_amdgcn_cvt_pk_u8_f32(rintf(acc.y), 2 * half + 1, packed);
        }
        *reinterpret_cast<uint32_t *>(out + ((size_t)wi * height + y) * width + q * 4) = packed;
    }
}


// Grid of warp blocks for a frame: blocks per row, per column (x Wn warps).
__host__ __device__ inline int warp_blocks_x(int width) { return ((width + 3) / 4 + 31) / 32; }
__host__ __device__ inline int warp_blocks_y(int height) { return (height + 8 * kWarpRowsPerThread - 1) / (8 * kWarpRowsPerThread); }
// The staged form needs 16-byte-aligned frame rows and dword stores.
inline bool warp_lds_eligible(const void *frame, const void *out, int width) { return (width & 15) == 0 && ((uintptr_t)frame & 15) == 0 && ((uintptr_t)out & 3) == 0; }

// Block `b` of a fused launch whose first warp_blocks_x * warp_blocks_y * Wn blocks are warp blocks (256 lanes each).
__device__ __forceinline__ void warp_lds_block_linear(const uint8_t *__restrict__ frame, const float *__restrict__ coeffs, uint8_t *__restrict__ out,
                                                      int width, int height, int b, int tid)
{
    const int nbx = warp_blocks_x(width), nby = warp_blocks_y(height);
    const int wi = b / (nbx * nby), rem = b - wi * nbx * nby, by = rem / nbx, bx = rem - by * nbx;
    warp_lds_block(frame, coeffs, out, width, height, (width + 3) / 4, bx, by, wi, tid & 31, tid >> 5);
}

}  // namespace
}  // namespace nmi
