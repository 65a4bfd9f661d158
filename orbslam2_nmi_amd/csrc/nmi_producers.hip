// nmi_producers.hip -- gfx950 kernels of the stack producers that feed the scoring path:
// warp stack (homography warps of the camera frame) and point-cloud render stacks (no OpenGL).  The textured-mesh
// renderer is in nmi_mesh.hip, the scoring kernels themselves in nmi_kernels.hip.
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
