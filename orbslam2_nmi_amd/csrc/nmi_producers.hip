// nmi_producers.hip -- gfx950 kernels of the stack producers that feed the scoring path:
// warp stack (homography warps of the camera frame), point-cloud and textured-mesh render stacks (no OpenGL).
// The scoring kernels themselves are in nmi_kernels.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nmi_kernels.h"

namespace nmi {

// ---------------------------------------------------------------------------------------------------------
// Warp-stack producer: Image::calculateWarping, Thirdparty/Localization/image.cpp:115-128 --
// Wn calls of cv::cuda::warpPerspective(frame, warped[w], K*R*K^-1, size) with the defaults INTER_LINEAR,
// BORDER_CONSTANT(0), forward matrix.  OpenCV 3.4.0 is not vendored in the reference and absent here, so the
// arithmetic below follows OpenCV's published device path (inverse matrix as 9 floats; coeff = 1/(c6*x+c7*y+c8),
// source coordinate coeff*(c0*x+c1*y+c2) in fp32; bilinear LinearFilter with floor(), the four taps accumulated in the
// order (y1,x1) (y1,x2) (y2,x1) (y2,x2); saturate_cast<uchar> = round to nearest even) -- parity unpinned.
// One thread produces 4 horizontally adjacent pixels of one warp and stores them as one dword.
__device__ __forceinline__ float warp_tap(const uint8_t *__restrict__ src, int w, int h, int x, int y)
{
    return (x >= 0 && x < w && y >= 0 && y < h) ? (float)src[y * w + x] : 0.0f;  // BORDER_CONSTANT, value 0
}

__global__ __launch_bounds__(256) void nmi_warp_kernel(const uint8_t *__restrict__ frame, const float *__restrict__ coeffs,
                                                       uint8_t *__restrict__ out, int width, int height, int quads_per_row)
{
    // a block is 32 quads x 8 rows = a 128 x 8 pixel patch of one warp: its taps fall into a compact patch of the frame
    // (a whole row per block spread them over up to 50 frame rows at the largest rotation of the grid)
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int wi = blockIdx.z;
    if (q >= quads_per_row || y >= height) return;
    const float *c = coeffs + wi * 9;
    uint32_t packed = 0;
    uint8_t px[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = q * 4 + k;
        uint32_t v = 0;
        if (x < width) {
            const float fx = (float)x, fy = (float)y;
            // OpenCV's device transform: one reciprocal of the homogeneous coordinate, two multiplies
            const float coeff = 1.0f / (c[6] * fx + c[7] * fy + c[8]);
            const float xs = coeff * (c[0] * fx + c[1] * fy + c[2]);
            const float ys = coeff * (c[3] * fx + c[4] * fy + c[5]);
            // coordinates far outside the frame (or non-finite) see only the constant border
            float acc = 0.0f;
            if (xs > -2.0f && xs < (float)(width + 1) && ys > -2.0f && ys < (float)(height + 1)) {
                const int x1 = (int)floorf(xs), y1 = (int)floorf(ys);
                const int x2 = x1 + 1, y2 = y1 + 1;
                float t11, t21, t12, t22;
                if (x1 >= 0 && x2 < width && y1 >= 0 && y2 < height) {  // all four taps inside: no per-tap border test
                    // two (possibly odd-addressed) 16-bit loads instead of four byte loads: the kernel is bound by the
                    // number of scattered load instructions, not by bytes (-12 %; fetching the sixteen taps of a lane's
                    // four pixels as two 8-byte loads when they share a frame row was slower again: more tests than loads)
                    const uint8_t *p = frame + y1 * width + x1;
                    unsigned short top, bot;
                    __builtin_memcpy(&top, p, 2);
                    __builtin_memcpy(&bot, p + width, 2);
                    t11 = (float)(top & 0xFFu), t21 = (float)(top >> 8), t12 = (float)(bot & 0xFFu), t22 = (float)(bot >> 8);
                } else {
                    t11 = warp_tap(frame, width, height, x1, y1), t21 = warp_tap(frame, width, height, x2, y1);
                    t12 = warp_tap(frame, width, height, x1, y2), t22 = warp_tap(frame, width, height, x2, y2);
                }
                acc = acc + t11 * (((float)x2 - xs) * ((float)y2 - ys));
                acc = acc + t21 * ((xs - (float)x1) * ((float)y2 - ys));
                acc = acc + t12 * (((float)x2 - xs) * (ys - (float)y1));
                acc = acc + t22 * ((xs - (float)x1) * (ys - (float)y1));
            }
            const float r = rintf(acc);
            v = r <= 0.0f ? 0u : (r >= 255.0f ? 255u : (uint32_t)r);
        }
        px[k] = (uint8_t)v;
        packed |= v << (8 * k);
    }
    uint8_t *dst = out + ((size_t)wi * height + y) * width + q * 4;
    if ((width & 3) == 0 && ((uintptr_t)out & 3) == 0) {
        *reinterpret_cast<uint32_t *>(dst) = packed;
    } else {
        for (int k = 0; k < 4 && q * 4 + k < width; ++k) dst[k] = px[k];
    }
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
constexpr int kWarpPatchBytes = 24 * 1024;
constexpr int kWarpRowsPerThread = 4;  // a block covers 128 x 32 output pixels: one patch fetch per 4096 pixels

__global__ __launch_bounds__(256) void nmi_warp_lds_kernel(const uint8_t *__restrict__ frame, const float *__restrict__ coeffs,
                                                           uint8_t *__restrict__ out, int width, int height, int quads_per_row)
{
    __shared__ __attribute__((aligned(16))) uint8_t patch[kWarpPatchBytes + 16];  // + slack for the 8-byte tap windows
    __shared__ int box[4];  // patch origin x (multiple of 16, may be -16), origin y (may be negative), pitch in bytes (0 = no patch), rows
    constexpr int kBlockRows = 8 * kWarpRowsPerThread;
    const int tid = threadIdx.y * 32 + threadIdx.x;
    const int q = blockIdx.x * 32 + threadIdx.x;
    const int wi = blockIdx.z;
    const float *c = coeffs + wi * 9;
    if (tid < 64) {
        // lanes 0..3: the four corners of this block's pixel rectangle, through the same fp32 expressions as the pixels
        const int bx0 = blockIdx.x * 128, by0 = blockIdx.y * kBlockRows;
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
            const int y = blockIdx.y * kBlockRows + rr * 8 + threadIdx.y;
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
    const float c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], c4 = c[4], c5 = c[5], c6 = c[6], c7 = c[7], c8 = c[8];
    const float xmax = (float)(width + 1), ymax = (float)(height + 1);
#pragma unroll 1
    for (int rr = 0; rr < kWarpRowsPerThread; ++rr) {
        const int y = blockIdx.y * kBlockRows + rr * 8 + threadIdx.y;
        if (y >= height) break;
        const float fy = (float)y;
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float fx = (float)(q * 4 + k);
            const float coeff = 1.0f / (c6 * fx + c7 * fy + c8);
            const float xs = coeff * (c0 * fx + c1 * fy + c2);
            const float ys = coeff * (c3 * fx + c4 * fy + c5);
            // nmi_warp_kernel gives 0 to pixels whose source lies outside (-2, width + 1) x (-2, height + 1).  Clamping the
            // source coordinate into that closed range does the same without a test: a clamped coordinate has both of its
            // taps (or its whole 2 x 2 window) in the zero border, and it stays inside this block's patch because the patch
            // is the corners' bounding box clipped to the very same range.
            const float xsc = __builtin_amdgcn_fmed3f(xs, -2.0f, xmax), ysc = __builtin_amdgcn_fmed3f(ys, -2.0f, ymax);
            const float x1f = floorf(xsc), y1f = floorf(ysc);
            // (unsigned min: a negative offset -- impossible while the bounding-box argument holds -- also ends up inside)
            const uint32_t lx = min((uint32_t)((int)x1f - px0), (uint32_t)(pitch - 2)), ly = min((uint32_t)((int)y1f - py0), (uint32_t)(rows - 2));
            // The two taps of a row are bytes o, o + 1 of the patch with o of any alignment.  An unaligned 2-byte LDS read
            // costs ~200 cycles of issue stall (measured: SQ_WAIT_INST_LDS 55 units per ds_read_u16, the whole kernel 40 us
            // however its taps were fetched), so the 8 aligned bytes around them are read and shifted instead.
            const uint32_t o = ly * (uint32_t)pitch + lx;
            const uint32_t *w0 = reinterpret_cast<const uint32_t *>(patch + (o & ~3u));
            const uint32_t *w1 = reinterpret_cast<const uint32_t *>(patch + (o & ~3u) + pitch);
            const uint32_t top = __builtin_amdgcn_alignbyte(w0[1], w0[0], o & 3u);
            const uint32_t bot = __builtin_amdgcn_alignbyte(w1[1], w1[0], o & 3u);
            const float t11 = (float)(top & 0xFFu), t21 = (float)((top >> 8) & 0xFFu);
            const float t12 = (float)(bot & 0xFFu), t22 = (float)((bot >> 8) & 0xFFu);
            const float x2f = x1f + 1.0f, y2f = y1f + 1.0f;  // exact: small integers
            float acc = 0.0f;
            acc = acc + t11 * ((x2f - xsc) * (y2f - ysc));
            acc = acc + t21 * ((xsc - x1f) * (y2f - ysc));
            acc = acc + t12 * ((x2f - xsc) * (ysc - y1f));
            acc = acc + t22 * ((xsc - x1f) * (ysc - y1f));
            // saturate_cast<uchar>: round to nearest even, clamp; acc >= 0, and the pack instruction saturates at 255
            packed = __builtin_amdgcn_cvt_pk_u8_f32(rintf(acc), k, packed);
        }
        *reinterpret_cast<uint32_t *>(out + ((size_t)wi * height + y) * width + q * 4) = packed;
    }
}

hipError_t launch_warp(const uint8_t *frame, const float *coeffs, uint8_t *out, int width, int height, int Wn,
                       hipStream_t stream)
{
    const int quads = (width + 3) / 4;
    dim3 block(32, 8), grid((quads + 31) / 32, (height + 7) / 8, Wn);
    // the staged form needs 16-byte-aligned frame rows and dword stores
    if ((width & 15) == 0 && ((uintptr_t)frame & 15) == 0 && ((uintptr_t)out & 3) == 0) {
        const int block_rows = 8 * kWarpRowsPerThread;
        hipLaunchKernelGGL(nmi_warp_lds_kernel, dim3(grid.x, (height + block_rows - 1) / block_rows, Wn), block, 0, stream, frame, coeffs,
                           out, width, height, quads);
    } else
        hipLaunchKernelGGL(nmi_warp_kernel, grid, block, 0, stream, frame, coeffs, out, width, height, quads);
    return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------------------
// Render-stack producer for coloured point clouds: Rendering<4>::renderToTextureOnGPU,
// Thirdparty/Localization/rendering.hpp:530-630 with shaders/ShadingWithColor.{vertex,fragment}shader --
// glClearColor(1,1,1) (:533), gl_Position = MVP * vec4(p,1), GL_POINTS of glPointSize(PointSize) (:307), GL_DEPTH_TEST
// with GL_LESS (:294-297), colour = vertex colour, only the red channel kept (GL_RED texture, :347).
// The OpenGL rasteriser is not part of the reference tree; the rules below are the OpenGL 3.3 specification's for
// non-antialiased points (centre clipped against the view volume; size rounded to an integer >= 1; odd sizes centred on
// floor(x)+0.5, even sizes on floor(x+0.5); 24-bit depth) evaluated in fp32 -- parity with a GL driver is unpinned.
// Depth test + colour write are one 32-bit atomicMin on (depth24 << 8 | red8); equal depths resolve to the darker
// fragment (GL: the first drawn).  Rows are written bottom-up like a GL texture (what NMI.cu:82 flips back).
__global__ __launch_bounds__(256) void nmi_zbuf_clear_kernel(uint32_t *zbuf, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) zbuf[i] = 0xFFFFFFFFu;
}

// First node of a captured search level (nmi_level_*): what would otherwise be two parameter uploads, a key reset
// and the buffer clear -- four graph nodes with a hand-over each -- as one kernel.  The parameters are read straight
// from the caller's pinned (device-mapped) buffers.
__global__ __launch_bounds__(256) void nmi_level_prep_kernel(const float *__restrict__ h_mvps, float *__restrict__ d_mvps, int n_mvps,
                                                             const float *__restrict__ h_coeffs, float *__restrict__ d_coeffs, int n_coeffs,
                                                             unsigned long long *key, uint32_t *zbuf, size_t nz)
{
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < n_mvps; i += blockDim.x) d_mvps[i] = h_mvps[i];
        for (int i = threadIdx.x; i < n_coeffs; i += blockDim.x) d_coeffs[i] = h_coeffs[i];
        if (threadIdx.x == 0) *key = 0ull;
    }
    uint4 *z4 = reinterpret_cast<uint4 *>(zbuf);
    const uint4 ones = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    for (size_t i = t; i < nz / 4; i += step) z4[i] = ones;
    for (size_t i = (nz & ~(size_t)3) + t; i < nz; i += step) zbuf[i] = 0xFFFFFFFFu;
}

hipError_t launch_level_prep(const float *h_mvps, float *d_mvps, int n_mvps, const float *h_coeffs, float *d_coeffs, int n_coeffs,
                             unsigned long long *key, uint32_t *zbuf, size_t nz, hipStream_t stream)
{
    hipLaunchKernelGGL(nmi_level_prep_kernel, dim3(2048), dim3(256), 0, stream, h_mvps, d_mvps, n_mvps, h_coeffs, d_coeffs, n_coeffs, key,
                       zbuf, nz);
    return hipGetLastError();
}

// View-frustum culling per block of 256 lanes.  Each lane brings the bounding box of its own primitive (a point, the
// three corners of a triangle; +inf / -inf for a lane without one); the primitives of a block are neighbours in the
// map's own order, so their common box is small.  A view whose clip planes put all eight corners of that box beyond ONE
// plane -- by a margin that covers the rounding of both this test and the per-primitive test that follows -- cannot
// receive anything from the block, which then skips that view's 256 transforms.  (A map seen from inside has most of
// itself outside any one view.)  The clip tests are affine in the position, so the box test is exact-conservative:
// results do not change.  On return (after a barrier) beyond[s] != 0 means "skip view s".  m_all must be loaded and
// beyond[s] preset to 0x3F by the caller, both before its own barrier... which this function provides.
__device__ __forceinline__ void block_frustum_cull(const float *m_all, int views, const float (&lo_in)[3], const float (&hi_in)[3],
                                                   float (*wave_box)[6], uint32_t *beyond)
{
    float lo[3] = {lo_in[0], lo_in[1], lo_in[2]}, hi[3] = {hi_in[0], hi_in[1], hi_in[2]};
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off, 64));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off, 64));
        }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) wave_box[threadIdx.x >> 6][k] = lo[k], wave_box[threadIdx.x >> 6][3 + k] = hi[k];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < views * 8; t += blockDim.x) {
        const int s = t >> 3, c = t & 7;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(fminf(wave_box[0][k], wave_box[1][k]), fminf(wave_box[2][k], wave_box[3][k]));
            hi[k] = fmaxf(fmaxf(wave_box[0][3 + k], wave_box[1][3 + k]), fmaxf(wave_box[2][3 + k], wave_box[3][3 + k]));
        }
        const float bx = (c & 1) ? hi[0] : lo[0], by = (c & 2) ? hi[1] : lo[1], bz = (c & 4) ? hi[2] : lo[2];
        const float ax = fmaxf(fabsf(lo[0]), fabsf(hi[0])), ay = fmaxf(fabsf(lo[1]), fabsf(hi[1])), az = fmaxf(fabsf(lo[2]), fabsf(hi[2]));
        const float *m = m_all + s * 16;
        const float cx = (m[0] * bx + m[4] * by) + (m[8] * bz + m[12]);
        const float cy = (m[1] * bx + m[5] * by) + (m[9] * bz + m[13]);
        const float cz = (m[2] * bx + m[6] * by) + (m[10] * bz + m[14]);
        const float cw = (m[3] * bx + m[7] * by) + (m[11] * bz + m[15]);
        // magnitude of the terms anywhere in the box (rounding of a 4-term fp32 sum is below 3e-7 of it; margin 1e-5)
        const float mw = fabsf(m[3]) * ax + fabsf(m[7]) * ay + fabsf(m[11]) * az + fabsf(m[15]);
        const float ex = 1e-5f * (fabsf(m[0]) * ax + fabsf(m[4]) * ay + fabsf(m[8]) * az + fabsf(m[12]) + mw);
        const float ey = 1e-5f * (fabsf(m[1]) * ax + fabsf(m[5]) * ay + fabsf(m[9]) * az + fabsf(m[13]) + mw);
        const float ez = 1e-5f * (fabsf(m[2]) * ax + fabsf(m[6]) * ay + fabsf(m[10]) * az + fabsf(m[14]) + mw);
        uint32_t code = 0;  // bit p: this corner is beyond clip plane p (comparisons with NaN / inf operands are false)
        code |= (cx + cw < -ex) ? 1u : 0u;   // cx < -cw
        code |= (cw - cx < -ex) ? 2u : 0u;   // cx >  cw
        code |= (cy + cw < -ey) ? 4u : 0u;
        code |= (cw - cy < -ey) ? 8u : 0u;
        code |= (cz + cw < -ez) ? 16u : 0u;
        code |= (cw - cz < -ez) ? 32u : 0u;
        atomicAnd(&beyond[s], code);
    }
    __syncthreads();
}

// One lane per point, looping over the S views: the cloud is read once, not once per view (27 views of a 3 M-point
// cloud would otherwise stream 1.3 GB per level).
//
// Culling is per WAVEFRONT (64 points that are neighbours in the map's own order) and costs no LDS and no barrier, so
// wavefronts do not wait for each other: the box of the 64 points (DPP / permute reductions), then lane v tests view v --
// for each of the six clip planes the box corner farthest along the plane's normal, with a margin that covers the
// rounding of this test and of the per-point test below; a view with that corner outside one plane cannot receive
// anything from these points.  The clip tests are affine in the position, so the test is exact-conservative: results do
// not change.  The surviving views come back as one 64-bit ballot; the loop over them is scalar (s_ff1), so each view's
// matrix arrives through scalar loads instead of 16 LDS reads per view and wavefront.  (The first version culled per
// 256-point block through LDS with two barriers and kept the matrices in LDS: 73 us for 3 M points x 27 views; this one
// 62 us = 15 us loading the cloud + 16 us culling + 32 us in the view loop, the atomics being 5 of those -- ablations
// in DESIGN.md section 7b.)
constexpr int kMaxViewsPerLaunch = 64;

template <typename T>
__device__ __forceinline__ T wave_min(T v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}


__device__ __forceinline__ void splat_point(const float *__restrict__ m, float x, float y, float z, uint32_t colour, uint32_t *__restrict__ zbuf,
                                            int s, int width, int height, int size, int stride)
{
    // glm mat4 * vec4: (m0*x + m1*y) + (m2*z + m3*1), component-wise
    const float cx = (m[0] * x + m[4] * y) + (m[8] * z + m[12]);
    const float cy = (m[1] * x + m[5] * y) + (m[9] * z + m[13]);
    const float cz = (m[2] * x + m[6] * y) + (m[10] * z + m[14]);
    const float cw = (m[3] * x + m[7] * y) + (m[11] * z + m[15]);
    if (!(cw > 0.0f) || cx < -cw || cx > cw || cy < -cw || cy > cw || cz < -cw || cz > cw) return;  // point clipping
    const float xw = (cx / cw * 0.5f + 0.5f) * (float)width;
    const float yw = (cy / cw * 0.5f + 0.5f) * (float)height;
    const float zw = cz / cw * 0.5f + 0.5f;
    const uint32_t depth = (uint32_t)(zw * 16777215.0f + 0.5f);
    const uint32_t frag = (depth << 8) | colour;
    int x0, y0;
    if (size & 1) {
        x0 = (int)floorf(xw) - (size - 1) / 2;
        y0 = (int)floorf(yw) - (size - 1) / 2;
    } else {
        x0 = (int)floorf(xw + 0.5f) - size / 2;
        y0 = (int)floorf(yw + 0.5f) - size / 2;
    }
    // Only the sprite's anchor (its lowest-left pixel) is written here: one atomic per point and view instead of
    // size^2.  Two points with the same anchor have the same footprint, so the farther one would lose on every
    // pixel anyway; the resolve pass below takes, for each pixel, the minimum over the size^2 anchors whose
    // sprites cover it -- exactly the depth-tested sprites.  The buffer is padded by size-1 so that sprites
    // straddling the left / bottom edge keep their anchor.
    const int ax = x0 + size - 1, ay = y0 + size - 1, wp = width + size - 1, hp = height + size - 1;
    if (ax < 0 || ax >= wp || ay < 0 || ay >= hp) return;
    atomicMin(&zbuf[((size_t)s * hp + ay) * stride + ax], frag);
}

// P = points per lane.  Two (a wavefront takes 128 consecutive points: half the culling per point, two loads in flight)
// measured slower than one: 107 vs 93 us for clear + splat + resolve of 3 M points x 27 views.
template <int P>
__global__ __launch_bounds__(256) void nmi_splat_kernel(const float *__restrict__ xyz, const float *__restrict__ red, long long npoints,
                                                        const float *__restrict__ mvps, int views, uint32_t *__restrict__ zbuf,
                                                        int width, int height, int size, int stride)
{
    const int lane = (int)(threadIdx.x & 63);
    // lane v's view matrix for the culling test below: asked for first, it is needed last
    const float4 *mv = reinterpret_cast<const float4 *>(mvps + (size_t)(lane < views ? lane : 0) * 16);  // column-major like glm: m[c*4 + r]
    const float4 c0 = mv[0], c1 = mv[1], c2 = mv[2], c3 = mv[3];
    const long long wave = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    bool valid[P];
    float x[P], y[P], z[P], r[P];
#pragma unroll
    for (int q = 0; q < P; ++q) {
        const long long i = (wave * P + q) * 64 + lane;
        valid[q] = i < npoints;
        x[q] = y[q] = z[q] = r[q] = 0.0f;
        if (valid[q]) x[q] = xyz[3 * i], y[q] = xyz[3 * i + 1], z[q] = xyz[3 * i + 2], r[q] = red[i];
    }

    // ---- which views can these points reach? ----
    const float inf = __builtin_huge_valf();
    float lx = inf, ly = inf, lz = inf, hx = -inf, hy = -inf, hz = -inf;
#pragma unroll
    for (int q = 0; q < P; ++q) {
        lx = fminf(lx, valid[q] ? x[q] : inf), ly = fminf(ly, valid[q] ? y[q] : inf), lz = fminf(lz, valid[q] ? z[q] : inf);
        hx = fmaxf(hx, valid[q] ? x[q] : -inf), hy = fmaxf(hy, valid[q] ? y[q] : -inf), hz = fmaxf(hz, valid[q] ? z[q] : -inf);
    }
    const float lox = wave_min(lx), loy = wave_min(ly), loz = wave_min(lz);
    const float hix = wave_max(hx), hiy = wave_max(hy), hiz = wave_max(hz);
    const float ax = fmaxf(fabsf(lox), fabsf(hix)), ay = fmaxf(fabsf(loy), fabsf(hiy)), az = fmaxf(fabsf(loz), fabsf(hiz));
    bool outside = lane >= views;
    {
        const float row[4][4] = {{c0.x, c1.x, c2.x, c3.x}, {c0.y, c1.y, c2.y, c3.y}, {c0.z, c1.z, c2.z, c3.z}, {c0.w, c1.w, c2.w, c3.w}};
        // magnitude of the terms of cw anywhere in the box (rounding of a 4-term fp32 sum is below 3e-7 of it; margin 1e-5)
        const float mw = fabsf(row[3][0]) * ax + fabsf(row[3][1]) * ay + fabsf(row[3][2]) * az + fabsf(row[3][3]);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float e = 1e-5f * (fabsf(row[j][0]) * ax + fabsf(row[j][1]) * ay + fabsf(row[j][2]) * az + fabsf(row[j][3]) + mw);
#pragma unroll
            for (int sgn = 0; sgn < 2; ++sgn) {
                // plane  cw + c_j >= 0  (sgn 0: c_j >= -cw)   or   cw - c_j >= 0  (sgn 1: c_j <= cw)
                const float a = sgn ? row[3][0] - row[j][0] : row[3][0] + row[j][0];
                const float b = sgn ? row[3][1] - row[j][1] : row[3][1] + row[j][1];
                const float c = sgn ? row[3][2] - row[j][2] : row[3][2] + row[j][2];
                const float d = sgn ? row[3][3] - row[j][3] : row[3][3] + row[j][3];
                // the largest value the plane function takes in the box (comparisons with NaN / inf operands are false)
                const float best = (a * (a >= 0.0f ? hix : lox) + b * (b >= 0.0f ? hiy : loy)) + (c * (c >= 0.0f ? hiz : loz) + d);
                // the coefficients themselves carry one rounding each: covered by the same margin (twice)
                outside = outside || best < -2.0f * e;
            }
        }
    }
    unsigned long long todo = ~__ballot(outside);
    if (views < 64) todo &= (1ull << views) - 1ull;

    uint32_t colour[P];
#pragma unroll
    for (int q = 0; q < P; ++q) colour[q] = (uint32_t)(fminf(fmaxf(r[q], 0.0f), 1.0f) * 255.0f + 0.5f);
    while (todo) {  // wavefront-uniform
        const int s = __builtin_ctzll(todo);
        todo &= todo - 1ull;
        const float *m = mvps + (size_t)s * 16;  // uniform address: scalar loads
#pragma unroll
        for (int q = 0; q < P; ++q)
            if (valid[q]) splat_point(m, x[q], y[q], z[q], colour[q], zbuf, s, width, height, size, stride);
    }
}

// Resolve.  Four horizontally adjacent output pixels per lane: the size x (size+3) anchor window is read once and the
// four results leave as one dword.  Rows of the anchor buffer are `stride` words apart (zbuf_stride: padded so that a
// lane can fetch its window as two aligned 16-byte loads per row).
template <int SIZE>
__global__ __launch_bounds__(256) void nmi_zbuf_resolve_fast_kernel(const uint32_t *__restrict__ zbuf, uint8_t *__restrict__ out, int views,
                                                                    int width, int height, int stride)
{
    // requires width % 4 == 0, SIZE <= 5, stride % 4 == 0 and (SIZE == 1 or stride >= width + 4)
    const int hp = height + SIZE - 1;
    const int quads = width >> 2;
    const size_t n = (size_t)views * height * quads;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % quads), py = (int)((i / quads) % height), s = (int)(i / ((size_t)quads * height));
        const uint32_t *base = zbuf + ((size_t)s * hp + py) * stride + q * 4;
        uint32_t best[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
#pragma unroll
        for (int dy = 0; dy < SIZE; ++dy) {
            const uint4 lo = *reinterpret_cast<const uint4 *>(base + (size_t)dy * stride);
            uint32_t v[8] = {lo.x, lo.y, lo.z, lo.w, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            if (SIZE > 1) {
                const uint4 hi = *reinterpret_cast<const uint4 *>(base + (size_t)dy * stride + 4);
                v[4] = hi.x, v[5] = hi.y, v[6] = hi.z, v[7] = hi.w;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int d = 0; d < SIZE; ++d) best[k] = min(best[k], v[k + d]);
        }
        uint8_t *dst = out + ((size_t)s * height + py) * width + q * 4;
        *reinterpret_cast<uint32_t *>(dst) = (best[0] & 0xFFu) | ((best[1] & 0xFFu) << 8) | ((best[2] & 0xFFu) << 16) | (best[3] << 24);
    }
}

// Any size / width.
__global__ __launch_bounds__(256) void nmi_zbuf_resolve_kernel(const uint32_t *__restrict__ zbuf, uint8_t *__restrict__ out, int views,
                                                               int width, int height, int size, int stride)
{
    const int wp = width + size - 1, hp = height + size - 1;
    const int quads = (width + 3) / 4;
    const size_t n = (size_t)views * height * quads;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % quads), py = (int)((i / quads) % height), s = (int)(i / ((size_t)quads * height));
        const int px = q * 4;
        const uint32_t *base = zbuf + ((size_t)s * hp + py) * stride + px;
        uint32_t best[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        for (int dy = 0; dy < size; ++dy) {
            const uint32_t *row = base + (size_t)dy * stride;
            for (int dx = 0; dx < size + 3; ++dx) {
                if (px + dx >= wp) break;
                const uint32_t v = row[dx];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (dx >= k && dx - k < size) best[k] = min(best[k], v);
            }
        }
        uint8_t *dst = out + ((size_t)s * height + py) * width + px;
        if (px + 3 < width && (width & 3) == 0) {
            *reinterpret_cast<uint32_t *>(dst) = (best[0] & 0xFFu) | ((best[1] & 0xFFu) << 8) | ((best[2] & 0xFFu) << 16) | (best[3] << 24);
        } else {
            for (int k = 0; k < 4 && px + k < width; ++k) dst[k] = (uint8_t)(best[k] & 0xFFu);  // untouched pixels keep 255
        }
    }
}

static void launch_resolve(const uint32_t *zbuf, uint8_t *out, int S, int width, int height, int size, hipStream_t stream)
{
    const int stride = zbuf_stride(width, size);
    const size_t nq = (size_t)S * height * ((width + 3) / 4);
    const dim3 grid((unsigned)((nq + 255) / 256 < 8192 ? (nq + 255) / 256 : 8192)), block(256);
    const bool fast = (width & 3) == 0 && size <= 5 && (((uintptr_t)zbuf & 15) == 0) && (((uintptr_t)out & 3) == 0);
    if (!fast) {
        hipLaunchKernelGGL(nmi_zbuf_resolve_kernel, grid, block, 0, stream, zbuf, out, S, width, height, size, stride);
        return;
    }
    switch (size) {
    case 1: hipLaunchKernelGGL(nmi_zbuf_resolve_fast_kernel<1>, grid, block, 0, stream, zbuf, out, S, width, height, stride); break;
    case 2: hipLaunchKernelGGL(nmi_zbuf_resolve_fast_kernel<2>, grid, block, 0, stream, zbuf, out, S, width, height, stride); break;
    case 3: hipLaunchKernelGGL(nmi_zbuf_resolve_fast_kernel<3>, grid, block, 0, stream, zbuf, out, S, width, height, stride); break;
    case 4: hipLaunchKernelGGL(nmi_zbuf_resolve_fast_kernel<4>, grid, block, 0, stream, zbuf, out, S, width, height, stride); break;
    default: hipLaunchKernelGGL(nmi_zbuf_resolve_fast_kernel<5>, grid, block, 0, stream, zbuf, out, S, width, height, stride); break;
    }
}

size_t render_zbuf_words(int S, int width, int height, int size) { return (size_t)S * zbuf_stride(width, size) * (height + size - 1); }

// ---------------------------------------------------------------------------------------------------------
// Render-stack producer for textured meshes (nmi_prop_RENDER 1): Rendering<1>::renderToTextureOnGPU,
// rendering.hpp:530-630 with shaders/ShadingWithTexture.* -- glDrawArrays(GL_TRIANGLES) of the OBJ's expanded
// vertex / uv arrays (objloader.cpp:140-224), GL_CULL_FACE (back faces, counter-clockwise front; rendering.hpp:300),
// depth test GL_LESS, fragment colour = 0.299 r + 0.587 g + 0.114 b of the texture sample (fragment shader :16) with
// GL_REPEAT wrap, GL_LINEAR magnification and GL_LINEAR_MIPMAP_LINEAR minification (texture.cpp:88-92).
// What follows is the OpenGL 3.3 pipeline in fp32 as the specification words it (pixel centres at +0.5, top-left
// fill rule, perspective-correct interpolation, isotropic level of detail from the per-pixel uv derivatives); a real
// driver rasterises in fixed point and is free in its LOD approximation, so parity with one is unpinned.
// Clipping: triangles are clipped against the NEAR plane (z_clip >= -w_clip) in clip space, as the GL pipeline does before
// the perspective divide (1 or 2 output triangles, clip coordinates and uv interpolated linearly along the cut edges,
// always from the inside vertex towards the outside one so that two triangles sharing an edge cut it at the same
// point).  A ground plane passing under the camera, a wall the camera stands next to -- the common case for a UAV over a
// terrain mesh or a camera inside a city model -- therefore keeps its visible part.  The other five planes need no
// geometric clipping: the pixel bounding box is clamped to the window and fragments beyond the far plane fail the
// per-pixel depth-range test.
// The texture arrives as per-level fp32 luma (host: nmi_texture_create), since every filter here is linear.
struct MeshTexture {
    const float *luma;   // all levels, level l at luma + off[l], row-major, row 0 = v 0
    int levels;
    int w[16], h[16];
    long long off[16];
};

__device__ __forceinline__ float tex_bilinear(const MeshTexture &t, int l, float u, float v)
{
    const int w = t.w[l], h = t.h[l];
    const float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    const float xf = floorf(x), yf = floorf(y);
    const float fx = x - xf, fy = y - yf;
    int i0 = (int)xf % w, j0 = (int)yf % h;  // GL_REPEAT
    if (i0 < 0) i0 += w;
    if (j0 < 0) j0 += h;
    const int i1 = i0 + 1 == w ? 0 : i0 + 1, j1 = j0 + 1 == h ? 0 : j0 + 1;
    const float *p = t.luma + t.off[l];
    const float t00 = p[(size_t)j0 * w + i0], t10 = p[(size_t)j0 * w + i1], t01 = p[(size_t)j1 * w + i0], t11 = p[(size_t)j1 * w + i1];
    const float a = t00 + (t10 - t00) * fx, b = t01 + (t11 - t01) * fx;
    return a + (b - a) * fy;
}

__device__ __forceinline__ bool edge_owner(float ex, float ey)
{
    // top-left rule for a counter-clockwise triangle in y-up window coordinates: an edge owns the pixels exactly on it
    // when it is a left edge (going down) or a top edge (horizontal, going left)
    return ey < 0.0f || (ey == 0.0f && ex < 0.0f);
}

// One triangle seen by one view: everything the per-pixel work needs.  Both kernels below build it with the same
// arithmetic, so which of them shades a pixel does not change its value.
struct TriView {
    float xw[3], yw[3], zw[3], iw[3];  // window x, y, depth, 1/w of the corners
    float ex[3], ey[3];                // edge k is opposite vertex k: from vertex (k+1)%3 to vertex (k+2)%3
    bool own[3];
    float inv_area;
    int x_lo, x_hi, y_lo, y_hi;        // pixel bounding box, clamped to the window
};

// A triangle in clip space after near-plane clipping: 3 or 4 corners in the original winding order with their uv.
struct ClipPoly {
    float cx[4], cy[4], cz[4], cw[4], u[4], v[4];
    int n;  // 0 (nothing left), 3 or 4
};

// Clip coordinates of the three corners and their signed distances d = z + w to the near plane (inside iff >= 0).
// Returns the number of corners inside.
__device__ __forceinline__ int tri_clip_coords(const float *__restrict__ m, const float (&px)[3], const float (&py)[3], const float (&pz)[3],
                                               float (&cx)[3], float (&cy)[3], float (&cz)[3], float (&cw)[3], float (&d)[3])
{
    int n_in = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        cx[k] = (m[0] * px[k] + m[4] * py[k]) + (m[8] * pz[k] + m[12]);
        cy[k] = (m[1] * px[k] + m[5] * py[k]) + (m[9] * pz[k] + m[13]);
        cz[k] = (m[2] * px[k] + m[6] * py[k]) + (m[10] * pz[k] + m[14]);
        cw[k] = (m[3] * px[k] + m[7] * py[k]) + (m[11] * pz[k] + m[15]);
        d[k] = cz[k] + cw[k];
        n_in += d[k] >= 0.0f ? 1 : 0;
    }
    return n_in;
}

// The rare case (1 or 2 corners inside): Sutherland-Hodgman against the near plane; corners are appended in winding
// order (3 or 4 of them).  Only the clip and tile kernels contain this code: nmi_mesh_kernel hands such triangles over.
__device__ __forceinline__ void tri_clip_poly(const float (&cx)[3], const float (&cy)[3], const float (&cz)[3], const float (&cw)[3],
                                           const float (&d)[3], const float (&tu)[3], const float (&tv)[3], ClipPoly &P)
{
    int n = 0;
    auto push = [&](float x, float y, float z, float w, float uu, float vv) {
        // n is 0..3 here; written as selects so that the arrays stay in registers
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j == n) P.cx[j] = x, P.cy[j] = y, P.cz[j] = z, P.cw[j] = w, P.u[j] = uu, P.v[j] = vv;
        ++n;
    };
    P.cx[3] = P.cy[3] = P.cz[3] = P.cw[3] = P.u[3] = P.v[3] = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int b = (k + 1) % 3;
        const bool in_a = d[k] >= 0.0f, in_b = d[b] >= 0.0f;
        if (in_a) push(cx[k], cy[k], cz[k], cw[k], tu[k], tv[k]);
        if (in_a != in_b) {
            const int i = in_a ? k : b, o = in_a ? b : k;  // from the inside corner towards the outside one
            const float t = d[i] / (d[i] - d[o]);
            const float w = cw[i] + (cw[o] - cw[i]) * t;
            push(cx[i] + (cx[o] - cx[i]) * t, cy[i] + (cy[o] - cy[i]) * t, -w /* on the near plane */, w, tu[i] + (tu[o] - tu[i]) * t,
                 tv[i] + (tv[o] - tv[i]) * t);
        }
    }
    P.n = n;
}

// Corners (0, sub + 1, sub + 2) of a clipped polygon.
__device__ __forceinline__ void poly_corners(const ClipPoly &P, int sub, float (&cx)[3], float (&cy)[3], float (&cz)[3], float (&cw)[3],
                                             float (&su)[3], float (&sv)[3])
{
    cx[0] = P.cx[0], cy[0] = P.cy[0], cz[0] = P.cz[0], cw[0] = P.cw[0], su[0] = P.u[0], sv[0] = P.v[0];
    cx[1] = sub ? P.cx[2] : P.cx[1], cy[1] = sub ? P.cy[2] : P.cy[1], cz[1] = sub ? P.cz[2] : P.cz[1], cw[1] = sub ? P.cw[2] : P.cw[1];
    su[1] = sub ? P.u[2] : P.u[1], sv[1] = sub ? P.v[2] : P.v[1];
    cx[2] = sub ? P.cx[3] : P.cx[2], cy[2] = sub ? P.cy[3] : P.cy[2], cz[2] = sub ? P.cz[3] : P.cz[2], cw[2] = sub ? P.cw[3] : P.cw[2];
    su[2] = sub ? P.u[3] : P.u[2], sv[2] = sub ? P.v[3] : P.v[2];
}

// One triangle given by the clip coordinates of its corners, seen through the window transform.  Returns false if it
// cannot produce a fragment.
__device__ __forceinline__ bool tri_setup(const float (&cx)[3], const float (&cy)[3], const float (&cz)[3], const float (&cw)[3], int width,
                                          int height, TriView &t)
{
    if (!(cw[0] > 0.0f) || !(cw[1] > 0.0f) || !(cw[2] > 0.0f)) return false;  // (a corner on or behind the eye plane survives near clipping only with a degenerate matrix)
    if ((cx[0] < -cw[0] && cx[1] < -cw[1] && cx[2] < -cw[2]) || (cx[0] > cw[0] && cx[1] > cw[1] && cx[2] > cw[2]) ||
        (cy[0] < -cw[0] && cy[1] < -cw[1] && cy[2] < -cw[2]) || (cy[0] > cw[0] && cy[1] > cw[1] && cy[2] > cw[2]) ||
        (cz[0] < -cw[0] && cz[1] < -cw[1] && cz[2] < -cw[2]) || (cz[0] > cw[0] && cz[1] > cw[1] && cz[2] > cw[2]))
        return false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        t.xw[k] = (cx[k] / cw[k] * 0.5f + 0.5f) * (float)width;
        t.yw[k] = (cy[k] / cw[k] * 0.5f + 0.5f) * (float)height;
        t.zw[k] = cz[k] / cw[k] * 0.5f + 0.5f;
        t.iw[k] = 1.0f / cw[k];
    }
    const float area = (t.xw[1] - t.xw[0]) * (t.yw[2] - t.yw[0]) - (t.xw[2] - t.xw[0]) * (t.yw[1] - t.yw[0]);
    if (!(area > 0.0f)) return false;  // back face (or degenerate): GL_CULL_FACE, front = counter-clockwise
    const float minx = fminf(t.xw[0], fminf(t.xw[1], t.xw[2])), maxx = fmaxf(t.xw[0], fmaxf(t.xw[1], t.xw[2]));
    const float miny = fminf(t.yw[0], fminf(t.yw[1], t.yw[2])), maxy = fmaxf(t.yw[0], fmaxf(t.yw[1], t.yw[2]));
    t.x_lo = max(0, (int)ceilf(minx - 0.5f)), t.x_hi = min(width - 1, (int)floorf(maxx - 0.5f));
    t.y_lo = max(0, (int)ceilf(miny - 0.5f)), t.y_hi = min(height - 1, (int)floorf(maxy - 0.5f));
    if (t.x_lo > t.x_hi || t.y_lo > t.y_hi) return false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int a = (k + 1) % 3, b = (k + 2) % 3;
        t.ex[k] = t.xw[b] - t.xw[a];
        t.ey[k] = t.yw[b] - t.yw[a];
        t.own[k] = edge_owner(t.ex[k], t.ey[k]);
    }
    t.inv_area = 1.0f / area;
    return true;
}

__device__ __forceinline__ void tri_attributes(const TriView &t, const float (&tu)[3], const float (&tv)[3], float fxp, float fyp, float &u,
                                               float &v, float &z, float (&bary)[3])
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int a = (k + 1) % 3;
        bary[k] = (t.ex[k] * (fyp - t.yw[a]) - t.ey[k] * (fxp - t.xw[a])) * t.inv_area;
    }
    z = bary[0] * t.zw[0] + bary[1] * t.zw[1] + bary[2] * t.zw[2];
    const float q = bary[0] * t.iw[0] + bary[1] * t.iw[1] + bary[2] * t.iw[2];
    u = (bary[0] * tu[0] * t.iw[0] + bary[1] * tu[1] * t.iw[1] + bary[2] * tu[2] * t.iw[2]) / q;
    v = (bary[0] * tv[0] * t.iw[0] + bary[1] * tv[1] * t.iw[1] + bary[2] * tv[2] * t.iw[2]) / q;
}

// Coverage test + depth + texture + depth-tested write of pixel (xx, yy).
__device__ __forceinline__ void tri_shade(const TriView &t, const float (&tu)[3], const float (&tv)[3], const MeshTexture &tex, int xx, int yy,
                                          uint32_t *__restrict__ img, int width)
{
    const float fxp = (float)xx + 0.5f, fyp = (float)yy + 0.5f;
    float bary[3], u, v, z;
    tri_attributes(t, tu, tv, fxp, fyp, u, v, z, bary);
    bool inside = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) inside = inside && (bary[k] > 0.0f || (bary[k] == 0.0f && t.own[k]));
    if (!inside) return;
    if (!(z >= 0.0f && z <= 1.0f)) return;  // depth clipping
    float b2[3], ux, vx, uy, vy, zz;
    tri_attributes(t, tu, tv, fxp + 1.0f, fyp, ux, vx, zz, b2);
    tri_attributes(t, tu, tv, fxp, fyp + 1.0f, uy, vy, zz, b2);
    const float tw = (float)tex.w[0], th = (float)tex.h[0];
    const float dudx = (ux - u) * tw, dvdx = (vx - v) * th, dudy = (uy - u) * tw, dvdy = (vy - v) * th;
    const float rho = fmaxf(sqrtf(dudx * dudx + dvdx * dvdx), sqrtf(dudy * dudy + dvdy * dvdy));
    float luma;
    const float lambda = log2f(rho);
    if (!(lambda > 0.0f)) {
        luma = tex_bilinear(tex, 0, u, v);  // magnification: GL_LINEAR on the base level
    } else {
        const float lc = fminf(lambda, (float)(tex.levels - 1));
        const int l0 = (int)floorf(lc), l1 = min(l0 + 1, tex.levels - 1);
        const float f = lc - (float)l0;
        const float s0 = tex_bilinear(tex, l0, u, v), s1 = tex_bilinear(tex, l1, u, v);
        luma = s0 + (s1 - s0) * f;   // GL_LINEAR_MIPMAP_LINEAR
    }
    const uint32_t colour = (uint32_t)(fminf(fmaxf(luma, 0.0f), 1.0f) * 255.0f + 0.5f);
    const uint32_t depth = (uint32_t)(z * 16777215.0f + 0.5f);
    atomicMin(&img[(size_t)yy * width + xx], (depth << 8) | colour);
}

// Rasterisation in three kernels (the middle one only has work when triangles cross the near plane).
//  nmi_mesh_kernel       one lane per triangle, looping over the views (the mesh is read once).  A triangle whose pixel
//                        bounding box in a view is at most kSmallBox pixels is shaded right there by its lane; a larger one
//                        is cut into 64 x 64 pixel screen tiles and each (triangle, view, tile) goes into a work queue.
//  nmi_mesh_clip_kernel  one lane per (triangle, view) that crosses the near plane -- handed over by nmi_mesh_kernel through a
//                        second, small queue so that the clipping code (a 4-corner polygon in registers) stays out of the
//                        kernel every triangle goes through: it clips, then shades small pieces itself and queues the
//                        tiles of large ones (a ground plane under the camera is two huge triangles).  If that queue
//                        overflows, this kernel finds the crossing triangles again by itself (a second pass over the mesh).
//  nmi_mesh_tile_kernel  one wavefront per queue entry: it rebuilds the TriView (the same arithmetic) and sweeps the
//                        tile's part of the bounding box, 64 pixels of a row at a time.
// A lane walking a 2,000-pixel triangle alone (facades, floors: the meshes this renderer is for) kept its wavefront busy
// 30 times longer than the other 63 lanes needed: 38 ms for 27 views of a 4,800-triangle plane, against 0.5 ms for the
// same plane in 1.9 M triangles.  If the queue is full the lane shades the triangle itself (slow, still exact).
constexpr int kSmallBox = 16;
constexpr int kTile = 64;

struct ClipItem {
    unsigned long long tri;
    uint32_t view, pad;
};

struct TileItem {
    uint32_t tri;
    uint32_t where;  // view (7 bits; at most 64 per launch) | sub-triangle << 7 | tile x << 8 | tile y << 20
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void nmi_mesh_kernel(const float *__restrict__ xyz, const float *__restrict__ uv, long long ntri,
                                                       const float *__restrict__ mvps, int views, uint32_t *__restrict__ zbuf,
                                                       int width, int height, MeshTexture tex, TileItem *__restrict__ queue,
                                                       unsigned long long *__restrict__ queue_state, unsigned long long queue_cap,
                                                       ClipItem *__restrict__ clipq, unsigned long long *__restrict__ clip_state,
                                                       unsigned long long clip_cap)
{
    // queue_state[0]: entries claimed so far.  Claims are contiguous, so at most one claim straddles the capacity and
    // every later one lies beyond it: the entries actually written are [0, start of the first claim that did not fit),
    // and queue_state[1] holds the bitwise NOT of that start (atomicMax from 0, so one memset resets both words).
    __shared__ float m_all[kMaxViewsPerLaunch * 16];
    __shared__ float wave_box[4][6];
    __shared__ uint32_t beyond[kMaxViewsPerLaunch];
    for (int t = threadIdx.x; t < views * 16; t += blockDim.x) m_all[t] = mvps[t];
    for (int t = threadIdx.x; t < views; t += blockDim.x) beyond[t] = 0x3Fu;
    const long long tri = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const bool valid = tri < ntri;
    float px[3] = {0, 0, 0}, py[3] = {0, 0, 0}, pz[3] = {0, 0, 0}, tu[3] = {0, 0, 0}, tv[3] = {0, 0, 0};
    if (valid) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            px[k] = xyz[(tri * 3 + k) * 3], py[k] = xyz[(tri * 3 + k) * 3 + 1], pz[k] = xyz[(tri * 3 + k) * 3 + 2];
            tu[k] = uv[(tri * 3 + k) * 2], tv[k] = uv[(tri * 3 + k) * 2 + 1];
        }
    }
    {
        // a triangle all of whose corners are beyond one clip plane is rejected by tri_setup; the block's box decides
        // that for its 256 triangles at once (block_frustum_cull)
        const float inf = __builtin_huge_valf();
        const float lo[3] = {valid ? fminf(px[0], fminf(px[1], px[2])) : inf, valid ? fminf(py[0], fminf(py[1], py[2])) : inf,
                             valid ? fminf(pz[0], fminf(pz[1], pz[2])) : inf};
        const float hi[3] = {valid ? fmaxf(px[0], fmaxf(px[1], px[2])) : -inf, valid ? fmaxf(py[0], fmaxf(py[1], py[2])) : -inf,
                             valid ? fmaxf(pz[0], fmaxf(pz[1], pz[2])) : -inf};
        block_frustum_cull(m_all, views, lo, hi, wave_box, beyond);
    }
    if (!valid) return;
    for (int s = 0; s < views; ++s) {
        if (beyond[s]) continue;  // block-uniform
        float cx[3], cy[3], cz[3], cw[3], d[3];
        const int n_in = tri_clip_coords(m_all + s * 16, px, py, pz, cx, cy, cz, cw, d);
        if (n_in < 3) {
            if (n_in > 0) {  // crosses the near plane: nmi_mesh_clip_kernel's business
                const unsigned long long at = atomicAdd(&clip_state[0], 1ull);
                if (at < clip_cap) clipq[at] = ClipItem{(unsigned long long)tri, (uint32_t)s, 0u};
            }
            continue;  // (n_in == 0: wholly in front of the near plane)
        }
        TriView t;
        if (!tri_setup(cx, cy, cz, cw, width, height, t)) continue;
        const int bw = t.x_hi - t.x_lo + 1, bh = t.y_hi - t.y_lo + 1;
        if (bw * bh > kSmallBox && queue != nullptr && tri <= 0xFFFFFFFFll) {
            const int tx0 = t.x_lo / kTile, tx1 = t.x_hi / kTile, ty0 = t.y_lo / kTile, ty1 = t.y_hi / kTile;
            const unsigned long long n = (unsigned long long)((tx1 - tx0 + 1) * (ty1 - ty0 + 1));
            const unsigned long long at = atomicAdd(&queue_state[0], n);
            if (at + n <= queue_cap) {
                unsigned long long k = at;
                for (int ty = ty0; ty <= ty1; ++ty)
                    for (int tx = tx0; tx <= tx1; ++tx) queue[k++] = TileItem{(uint32_t)tri, (uint32_t)s | ((uint32_t)tx << 8) | ((uint32_t)ty << 20)};
                continue;
            }
            atomicMax(&queue_state[1], ~at);  // queue full: this lane does the work itself (below)
        }
        uint32_t *img = zbuf + (size_t)s * width * height;
        for (int yy = t.y_lo; yy <= t.y_hi; ++yy)
            for (int xx = t.x_lo; xx <= t.x_hi; ++xx) tri_shade(t, tu, tv, tex, xx, yy, img, width);
    }
}

// One (triangle, view) that crosses the near plane: clip, then each of the 1 or 2 pieces goes the way of any triangle.
__device__ __forceinline__ void clip_and_raster(const float *__restrict__ xyz, const float *__restrict__ uv, long long tri, int s,
                                                const float *__restrict__ m, uint32_t *__restrict__ zbuf, int width, int height,
                                                const MeshTexture &tex, TileItem *__restrict__ queue, unsigned long long *__restrict__ queue_state,
                                                unsigned long long queue_cap)
{
    float px[3], py[3], pz[3], tu[3], tv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        px[k] = xyz[(tri * 3 + k) * 3], py[k] = xyz[(tri * 3 + k) * 3 + 1], pz[k] = xyz[(tri * 3 + k) * 3 + 2];
        tu[k] = uv[(tri * 3 + k) * 2], tv[k] = uv[(tri * 3 + k) * 2 + 1];
    }
    float cx[3], cy[3], cz[3], cw[3], d[3];
    const int n_in = tri_clip_coords(m, px, py, pz, cx, cy, cz, cw, d);
    if (n_in == 0 || n_in == 3) return;  // not this kernel's (the rescan visits every triangle)
    ClipPoly P;
    tri_clip_poly(cx, cy, cz, cw, d, tu, tv, P);
    for (int sub = 0; sub + 3 <= P.n; ++sub) {
        TriView t;
        float su[3], sv[3];
        poly_corners(P, sub, cx, cy, cz, cw, su, sv);
        if (!tri_setup(cx, cy, cz, cw, width, height, t)) continue;
        const int bw = t.x_hi - t.x_lo + 1, bh = t.y_hi - t.y_lo + 1;
        if (bw * bh > kSmallBox && queue != nullptr && tri <= 0xFFFFFFFFll) {
            const int tx0 = t.x_lo / kTile, tx1 = t.x_hi / kTile, ty0 = t.y_lo / kTile, ty1 = t.y_hi / kTile;
            const unsigned long long n = (unsigned long long)((tx1 - tx0 + 1) * (ty1 - ty0 + 1));
            const unsigned long long at = atomicAdd(&queue_state[0], n);
            if (at + n <= queue_cap) {
                unsigned long long k = at;
                for (int ty = ty0; ty <= ty1; ++ty)
                    for (int tx = tx0; tx <= tx1; ++tx)
                        queue[k++] = TileItem{(uint32_t)tri, (uint32_t)s | ((uint32_t)sub << 7) | ((uint32_t)tx << 8) | ((uint32_t)ty << 20)};
                continue;
            }
            atomicMax(&queue_state[1], ~at);
        }
        uint32_t *img = zbuf + (size_t)s * width * height;
        for (int yy = t.y_lo; yy <= t.y_hi; ++yy)
            for (int xx = t.x_lo; xx <= t.x_hi; ++xx) tri_shade(t, su, sv, tex, xx, yy, img, width);
    }
}

__global__ __launch_bounds__(256) void nmi_mesh_clip_kernel(const float *__restrict__ xyz, const float *__restrict__ uv, long long ntri,
                                                            const float *__restrict__ mvps, int views, uint32_t *__restrict__ zbuf, int width,
                                                            int height, MeshTexture tex, TileItem *__restrict__ queue,
                                                            unsigned long long *__restrict__ queue_state, unsigned long long queue_cap,
                                                            const ClipItem *__restrict__ clipq, const unsigned long long *__restrict__ clip_state,
                                                            unsigned long long clip_cap)
{
    const unsigned long long claimed = clip_state[0];
    const unsigned long long gid = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x, stride = (unsigned long long)gridDim.x * blockDim.x;
    if (claimed <= clip_cap) {
        for (unsigned long long i = gid; i < claimed; i += stride)
            clip_and_raster(xyz, uv, (long long)clipq[i].tri, (int)clipq[i].view, mvps + clipq[i].view * 16, zbuf, width, height, tex, queue,
                            queue_state, queue_cap);
    } else {
        // more crossing triangles than the queue holds: look at every (triangle, view) again
        const unsigned long long all = (unsigned long long)ntri * (unsigned long long)views;
        for (unsigned long long i = gid; i < all; i += stride) {
            const long long tri = (long long)(i / (unsigned long long)views);
            const int s = (int)(i % (unsigned long long)views);
            clip_and_raster(xyz, uv, tri, s, mvps + s * 16, zbuf, width, height, tex, queue, queue_state, queue_cap);
        }
    }
}

__global__ __launch_bounds__(64) void nmi_mesh_tile_kernel(const float *__restrict__ xyz, const float *__restrict__ uv,
                                                           const float *__restrict__ mvps, uint32_t *__restrict__ zbuf, int width, int height,
                                                           MeshTexture tex, const TileItem *__restrict__ queue,
                                                           const unsigned long long *__restrict__ queue_state)
{
    const unsigned long long claimed = queue_state[0], first_unfit = ~queue_state[1];
    const unsigned long long count = claimed < first_unfit ? claimed : first_unfit;  // the written prefix (see nmi_mesh_kernel)
    const int lane = threadIdx.x;
    for (unsigned long long i = blockIdx.x; i < count; i += gridDim.x) {
        const TileItem it = queue[i];
        const long long tri = it.tri;
        const int s = (int)(it.where & 0x7Fu), sub = (int)((it.where >> 7) & 1u), tx = (int)((it.where >> 8) & 0xFFFu), ty = (int)(it.where >> 20);
        float px[3], py[3], pz[3], tu[3], tv[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            px[k] = xyz[(tri * 3 + k) * 3], py[k] = xyz[(tri * 3 + k) * 3 + 1], pz[k] = xyz[(tri * 3 + k) * 3 + 2];
            tu[k] = uv[(tri * 3 + k) * 2], tv[k] = uv[(tri * 3 + k) * 2 + 1];
        }
        TriView t;
        float cx[3], cy[3], cz[3], cw[3], d[3];
        float su[3] = {tu[0], tu[1], tu[2]}, sv[3] = {tv[0], tv[1], tv[2]};
        const int n_in = tri_clip_coords(mvps + s * 16, px, py, pz, cx, cy, cz, cw, d);
        if (n_in < 3) {
            ClipPoly P;
            tri_clip_poly(cx, cy, cz, cw, d, tu, tv, P);
            if (n_in == 0 || sub + 3 > P.n) continue;
            poly_corners(P, sub, cx, cy, cz, cw, su, sv);
        }
        if (!tri_setup(cx, cy, cz, cw, width, height, t)) continue;  // cannot happen: the same test passed before
        const int x0 = max(t.x_lo, tx * kTile), x1 = min(t.x_hi, tx * kTile + kTile - 1);
        const int y0 = max(t.y_lo, ty * kTile), y1 = min(t.y_hi, ty * kTile + kTile - 1);
        uint32_t *img = zbuf + (size_t)s * width * height;
        // the wavefront is an 8 x 8 pixel stamp (a 9 x 8 box is two steps, not eight rows of nine lanes)
        const int lx = lane & 7, ly = lane >> 3;
        for (int by = y0; by <= y1; by += 8)
            for (int bx = x0; bx <= x1; bx += 8) {
                const int xx = bx + lx, yy = by + ly;
                if (xx <= x1 && yy <= y1) tri_shade(t, su, sv, tex, xx, yy, img, width);
            }
    }
}

hipError_t launch_render_mesh(const float *xyz, const float *uv, long long ntri, const float *luma, int levels, const int *lw,
                              const int *lh, const long long *loff, const float *mvps, int S, uint32_t *zbuf, uint8_t *out,
                              int width, int height, void *tile_queue, unsigned long long tile_queue_cap, unsigned long long *queue_state,
                              void *clip_queue, unsigned long long clip_queue_cap, hipStream_t stream, bool clear_first)
{
    MeshTexture tex{};
    tex.luma = luma;
    tex.levels = levels;
    for (int l = 0; l < levels && l < 16; ++l) tex.w[l] = lw[l], tex.h[l] = lh[l], tex.off[l] = loff[l];
    const size_t nz = (size_t)S * width * height;
    if (clear_first)
        hipLaunchKernelGGL(nmi_zbuf_clear_kernel, dim3((unsigned)((nz + 255) / 256 < 4096 ? (nz + 255) / 256 : 4096)), dim3(256), 0, stream, zbuf, nz);
    if (ntri > 0) {
        if (!queue_state || !clip_queue) return hipErrorInvalidValue;
        TileItem *queue = tile_queue_cap > 0 ? static_cast<TileItem *>(tile_queue) : nullptr;
        for (int s0 = 0; s0 < S; s0 += kMaxViewsPerLaunch) {
            const int views = S - s0 < kMaxViewsPerLaunch ? S - s0 : kMaxViewsPerLaunch;
            // queue_state: [0] tile items claimed, [1] ~(first claim that did not fit), [2] clip items claimed, [3] unused
            const hipError_t e = hipMemsetAsync(queue_state, 0, 4 * sizeof(unsigned long long), stream);
            if (e != hipSuccess) return e;
            ClipItem *clipq = static_cast<ClipItem *>(clip_queue);
            hipLaunchKernelGGL(nmi_mesh_kernel, dim3((unsigned)((ntri + 255) / 256)), dim3(256), 0, stream, xyz, uv, ntri,
                               mvps + (size_t)s0 * 16, views, zbuf + (size_t)s0 * width * height, width, height, tex, queue, queue_state,
                               tile_queue_cap, clipq, queue_state + 2, clip_queue_cap);
            hipLaunchKernelGGL(nmi_mesh_clip_kernel, dim3(512), dim3(256), 0, stream, xyz, uv, ntri, mvps + (size_t)s0 * 16, views,
                               zbuf + (size_t)s0 * width * height, width, height, tex, queue, queue_state, tile_queue_cap, clipq,
                               queue_state + 2, clip_queue_cap);
            if (queue)
                hipLaunchKernelGGL(nmi_mesh_tile_kernel, dim3(16384), dim3(64), 0, stream, xyz, uv, mvps + (size_t)s0 * 16,
                                   zbuf + (size_t)s0 * width * height, width, height, tex, queue, queue_state);
        }
    }
    launch_resolve(zbuf, out, S, width, height, 1, stream);
    return hipGetLastError();
}

size_t mesh_tile_item_bytes() { return sizeof(TileItem); }
size_t mesh_clip_item_bytes() { return sizeof(ClipItem); }

hipError_t launch_render_points(const float *xyz, const float *red, long long npoints, const float *mvps, int S, uint32_t *zbuf,
                                uint8_t *out, int width, int height, int size, hipStream_t stream, bool clear_first)
{
    const size_t nz = render_zbuf_words(S, width, height, size);
    const size_t n = (size_t)S * width * height;
    if (clear_first)
        hipLaunchKernelGGL(nmi_zbuf_clear_kernel, dim3((unsigned)((nz + 255) / 256 < 4096 ? (nz + 255) / 256 : 4096)), dim3(256), 0, stream, zbuf, nz);
    if (npoints > 0) {
        const int stride = zbuf_stride(width, size);
        const size_t per_view = (size_t)stride * (height + size - 1);
        for (int s0 = 0; s0 < S; s0 += kMaxViewsPerLaunch) {
            const int views = S - s0 < kMaxViewsPerLaunch ? S - s0 : kMaxViewsPerLaunch;
            hipLaunchKernelGGL(nmi_splat_kernel<1>, dim3((unsigned)((npoints + 255) / 256)), dim3(256), 0, stream, xyz, red, npoints,
                               mvps + (size_t)s0 * 16, views, zbuf + (size_t)s0 * per_view, width, height, size, stride);
        }
    }
    launch_resolve(zbuf, out, S, width, height, size, stream);
    (void)n;
    return hipGetLastError();
}

}  // namespace nmi
