// nmi_producers.hip -- gfx950 kernels of the stack producers that feed the scoring path:
// warp stack (homography warps of the camera frame) and point-cloud render stacks (no OpenGL).  The textured-mesh
// renderer is in nmi_mesh.hip, the scoring kernels themselves in nmi_kernels.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>

#include "nmi_kernels.h"
#include "nmi_warp_device.h"
#include "nmi_cloud_device.h"

namespace nmi {

// ---------------------------------------------------------------------------------------------------------
// Warp-stack producer: Image::calculateWarping, Thirdparty/Localization/image.cpp:115-128 --
// Wn calls of cv::cuda::warpPerspective(frame, warped[w], K*R*K^-1, size) with the defaults INTER_LINEAR,
// BORDER_CONSTANT(0), forward matrix.  OpenCV 3.4.0 is not vendored in the reference and absent here, so the
// arithmetic below follows OpenCV's published device path (inverse matrix as 9 floats; coeff = 1/(c6*x+c7*y+c8),
// source coordinate coeff*(c0*x+c1*y+c2) in fp32; bilinear LinearFilter with floor(), the four taps accumulated in the
// order (y1,x1) (y1,x2) (y2,x1) (y2,x2); saturate_cast<uchar> = round to nearest even) -- parity unpinned.
// One thread produces 4 horizontally adjacent pixels of one warp and stores them as one dword.
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

__global__ __launch_bounds__(256) void nmi_warp_lds_kernel(const uint8_t *__restrict__ frame, const float *__restrict__ coeffs,
                                                           uint8_t *__restrict__ out, int width, int height, int quads_per_row)
{
    warp_lds_block(frame, coeffs, out, width, height, quads_per_row, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)threadIdx.x, (int)threadIdx.y);
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
// With `kept` (a point-cloud level's fused form) the launch has more workgroups: workgroup b >= 1 tests the boxes of wavefronts
// 256 (b - 1) .. of the packed cloud -- ONE LANE PER BOX -- against the six planes around all the views (read straight from the
// host's pinned buffer: `h_bound` = 24 floats + the replay's parity word) and appends the numbers of those that any view may
// see to the compact list kept[], one counter update per wavefront.  The front kernel then runs a bounded number of worker
// workgroups over that list instead of one wavefront per 64 points of the whole map.  Two counters, used in turn (parity):
// workgroup 0 zeroes the one the NEXT replay will count into.
__global__ __launch_bounds__(256) void nmi_level_prep_kernel(const float *__restrict__ h_mvps, float *__restrict__ d_mvps, int n_mvps,
                                                             const float *__restrict__ h_coeffs, float *__restrict__ d_coeffs, int n_coeffs,
                                                             unsigned long long *key, uint32_t *zbuf, size_t nz, uint32_t *epoch,
                                                             const float4 *__restrict__ boxes, long long nwaves, const float *__restrict__ h_bound,
                                                             uint32_t *__restrict__ kept, uint32_t *__restrict__ kept_count)
{
    if (kept && blockIdx.x > 0) {
        const uint32_t parity = __float_as_uint(h_bound[24]) & 1u;   // (wavefront-uniform addresses in host memory: scalar loads, one PCIe trip)
        const long long w = ((long long)blockIdx.x - 1) * 256 + threadIdx.x;
        bool in = false;
        if (w < nwaves) in = !box_outside_bound(h_bound, boxes[2 * w], boxes[2 * w + 1]);
        const unsigned long long mask = __ballot(in);
        const int lane = (int)(threadIdx.x & 63);
        uint32_t base = 0;
        if (lane == 0 && mask) base = atomicAdd(&kept_count[parity], (uint32_t)__popcll(mask));
        base = (uint32_t)__shfl((int)base, 0, 64);
        if (in) kept[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = (uint32_t)w;
        return;
    }
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    if (blockIdx.x == 0) {
        if (kept && threadIdx.x == 64) kept_count[(__float_as_uint(h_bound[24]) & 1u) ^ 1u] = 0u;
        // Every read of the host's buffers is a round trip over PCIe (about 2 us): all of them are asked for before the first
        // one is used, so the kernel costs one round trip, not one per loop iteration (7 us -> 4 us on the level's serial path).
        const int bd = (int)blockDim.x, i0 = (int)threadIdx.x;
        float m[4], c[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) m[k] = i0 + k * bd < n_mvps ? __builtin_nontemporal_load(h_mvps + i0 + k * bd) : 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) c[k] = i0 + k * bd < n_coeffs ? __builtin_nontemporal_load(h_coeffs + i0 + k * bd) : 0.0f;
        const uint32_t e = (threadIdx.x == 0 && epoch) ? *epoch : 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i0 + k * bd < n_mvps) d_mvps[i0 + k * bd] = m[k];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i0 + k * bd < n_coeffs) d_coeffs[i0 + k * bd] = c[k];
        for (int i = i0 + 4 * bd; i < n_mvps; i += bd) d_mvps[i] = h_mvps[i];
        for (int i = i0 + 4 * bd; i < n_coeffs; i += bd) d_coeffs[i] = h_coeffs[i];
        if (threadIdx.x == 0) *key = 0ull;
        if (threadIdx.x == 0 && epoch) *epoch = e + 1u;  // this replay's anchor buffer: zbuf pair [epoch & 1] (nmi_level_front_kernel)
    }
    uint4 *z4 = reinterpret_cast<uint4 *>(zbuf);
    const uint4 ones = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    for (size_t i = t; i < nz / 4; i += step) z4[i] = ones;
    for (size_t i = (nz & ~(size_t)3) + t; i < nz; i += step) zbuf[i] = 0xFFFFFFFFu;
}

hipError_t launch_level_prep(const float *h_mvps, float *d_mvps, int n_mvps, const float *h_coeffs, float *d_coeffs, int n_coeffs,
                             unsigned long long *key, uint32_t *zbuf, size_t nz, hipStream_t stream, uint32_t *epoch, const void *packed,
                             long long npoints, const float *h_bound, uint32_t *kept, uint32_t *kept_count)
{
    const float4 *boxes = nullptr;
    long long nwaves = 0;
    if (kept) {
        if (nz || !packed || !h_bound || !kept_count) return hipErrorInvalidValue;
        size_t off = 0;
        (void)cloud_pack_bytes(npoints, &off);
        boxes = reinterpret_cast<const float4 *>(static_cast<const char *>(packed) + off);
        nwaves = (npoints + 63) / 64;
    }
    const unsigned blocks = kept ? 1u + (unsigned)((nwaves + 255) / 256) : (nz ? 2048u : 1u);
    hipLaunchKernelGGL(nmi_level_prep_kernel, dim3(blocks), dim3(256), 0, stream, h_mvps, d_mvps, n_mvps, h_coeffs, d_coeffs, n_coeffs, key, zbuf, nz,
                       epoch, boxes, nwaves, h_bound, kept, kept_count);
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
// in profiles/NOTES.md.)
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
    int ax, ay;
    uint32_t frag;
    if (!splat_anchor(m, x, y, z, colour, width, height, size, ax, ay, frag)) return;
    uint32_t *zview = zbuf + (size_t)s * (height + size - 1) * stride;   // wavefront-uniform in the callers' view loops: scalar arithmetic
    atomicMin(&zview[ay * stride + ax], frag);
}

__global__ __launch_bounds__(256) void nmi_cloud_pack_kernel(const float *__restrict__ xyz, const float *__restrict__ red, long long npoints,
                                                             float4 *__restrict__ points, float4 *__restrict__ boxes)
{
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const bool valid = i < npoints;
    const float inf = __builtin_huge_valf();
    float x = 0.0f, y = 0.0f, z = 0.0f, r = 0.0f;
    if (valid) {
        x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2], r = red[i];
        points[i] = make_float4(x, y, z, r);
    }
    const float lox = wave_min(valid ? x : inf), loy = wave_min(valid ? y : inf), loz = wave_min(valid ? z : inf);
    const float hix = wave_max(valid ? x : -inf), hiy = wave_max(valid ? y : -inf), hiz = wave_max(valid ? z : -inf);
    if ((threadIdx.x & 63) == 0 && valid) {
        boxes[2 * (i >> 6)] = make_float4(lox, loy, loz, 0.0f);
        boxes[2 * (i >> 6) + 1] = make_float4(hix, hiy, hiz, 0.0f);
    }
}

// The 64 points of wavefront `wave` of the launch (lane = this lane) into the views' anchor buffers.
// (One point per lane: two -- a wavefront taking 128 consecutive points, half the culling per point, two loads in flight --
// measured slower, 107 vs 93 us for clear + splat + resolve of 3 M points x 27 views.)
template <bool PACKED>
__device__ __forceinline__ void splat_wave(const float *__restrict__ xyz, const float *__restrict__ red, PackedCloud pc, long long npoints,
                                           const float *__restrict__ mvps, int views, uint32_t *__restrict__ zbuf, int width, int height,
                                           int size, int stride, long long wave, int lane)
{
    // lane v's view matrix for the culling test below: asked for first, it is needed last
    const float4 *mv = reinterpret_cast<const float4 *>(mvps + (size_t)(lane < views ? lane : 0) * 16);  // column-major like glm: m[c*4 + r]
    const float4 c0 = mv[0], c1 = mv[1], c2 = mv[2], c3 = mv[3];
    const long long i = wave * 64 + lane;
    const bool valid = i < npoints;
    float x = 0.0f, y = 0.0f, z = 0.0f, r = 0.0f;
    float lox, loy, loz, hix, hiy, hiz;
    if (PACKED) {
        if (valid) {
            const float4 p = pc.points[i];
            x = p.x, y = p.y, z = p.z, r = p.w;
        }
        const float4 lo = pc.boxes[2 * wave], hi = pc.boxes[2 * wave + 1];  // the same address for the whole wavefront
        lox = lo.x, loy = lo.y, loz = lo.z, hix = hi.x, hiy = hi.y, hiz = hi.z;
    } else {
        if (valid) x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2], r = red[i];
        const float inf = __builtin_huge_valf();
        lox = wave_min(valid ? x : inf), loy = wave_min(valid ? y : inf), loz = wave_min(valid ? z : inf);
        hix = wave_max(valid ? x : -inf), hiy = wave_max(valid ? y : -inf), hiz = wave_max(valid ? z : -inf);
    }

    // ---- which views can these points reach? ----  (lane v tests view v: nmi_cloud_device.h)
    const bool outside = lane >= views || box_outside_view(c0, c1, c2, c3, lox, loy, loz, hix, hiy, hiz);
    unsigned long long todo = ~__ballot(outside);
    if (views < 64) todo &= (1ull << views) - 1ull;
    const uint32_t colour = (uint32_t)(fminf(fmaxf(r, 0.0f), 1.0f) * 255.0f + 0.5f);
    while (todo) {  // wavefront-uniform
        const int s = __builtin_ctzll(todo);
        todo &= todo - 1ull;
        // uniform address: scalar loads.  (Taking the matrix from the registers of lane s, where the culling test already has it,
        // with 16 lane reads instead: 72 vs 64 us for the level's front kernel -- the scalar cache serves these loads.)
        const float *m = mvps + (size_t)s * 16;
        if (valid) splat_point(m, x, y, z, colour, zbuf, s, width, height, size, stride);
    }
}

__global__ __launch_bounds__(256) void nmi_splat_kernel(const float *__restrict__ xyz, const float *__restrict__ red, long long npoints,
                                                        const float *__restrict__ mvps, int views, uint32_t *__restrict__ zbuf,
                                                        int width, int height, int size, int stride)
{
    splat_wave<false>(xyz, red, PackedCloud{}, npoints, mvps, views, zbuf, width, height, size, stride,
                      (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6, (int)(threadIdx.x & 63));
}

// The front kernel of a captured point-cloud level: its first `warp_blocks` workgroups make the warp stack
// (nmi_warp_device.h), the others splat the packed cloud -- one kernel node instead of a fork and a join in the graph.
// The anchors are double-buffered by replay parity (*epoch, bumped by the prep node): this replay splats into buffer
// epoch & 1 -- cleared by the PREVIOUS replay's front kernel -- and its last `clear_blocks` workgroups clear the other one for
// the next replay.  The 44 MB of stores ride under a kernel that leaves the HBM idle (it is bound by per-wavefront latency
// chains) instead of making a 9 us clear pass at the head of every level.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void nmi_level_front_kernel(PackedCloud pc, long long npoints, const float *__restrict__ mvps, int views,
                                                              uint32_t *__restrict__ zbuf, size_t pair_words, const uint32_t *__restrict__ epoch,
                                                              int width, int height, int size, int stride,
                                                              const uint8_t *__restrict__ frame, const float *__restrict__ coeffs,
                                                              uint8_t *__restrict__ warps, int warp_blocks, int splat_blocks, int clear_blocks,
                                                              const uint32_t *__restrict__ kept, const uint32_t *__restrict__ kept_count)
{
    // Order of the launch: splat workers, warp blocks, clear blocks.  The splat's atomics run at the memory side's rate whatever
    // else the chip does, so they start at once; warp and clear blocks are arithmetic and stores that fit beside them.
    // The splat workers are a BOUNDED number of workgroups (launch_level_front_points: three per CU) that share out the list of
    // wavefronts the prep kernel found in reach of a view: a wavefront of the splat spends its life waiting for its turn at the
    // CU's atomic path, and one per 64 visible points -- 20 per CU on the benchmark's cloud -- held the CU's wave slots while the
    // warp blocks queued (plane of the e2e bench: splat alone 41 us, warp alone 25, together 57; and the 8 of 9 wavefronts
    // that see nothing no longer exist as wavefronts at all).
    if ((long long)blockIdx.x >= splat_blocks && (long long)blockIdx.x < splat_blocks + warp_blocks) {
        warp_lds_block_linear(frame, coeffs, warps, width, height, (int)(blockIdx.x - splat_blocks), (int)threadIdx.x);
        return;
    }
    const uint32_t parity = *epoch & 1u;
    const long long b = (long long)blockIdx.x < splat_blocks ? (long long)blockIdx.x : (long long)blockIdx.x - warp_blocks;
    if (b < splat_blocks) {
        const uint32_t n = kept_count[__float_as_uint(mvps[(size_t)views * 16 + 24]) & 1u];   // (the parity the prep kernel counted under)
        const uint32_t first = (uint32_t)b * 4u + (threadIdx.x >> 6), step = (uint32_t)splat_blocks * 4u;
        for (uint32_t e = first; e < n; e += step) {
            const long long wave = (long long)__builtin_amdgcn_readfirstlane((int)kept[e]);
            splat_wave<true>(nullptr, nullptr, pc, npoints, mvps, views, zbuf + (size_t)parity * pair_words, width, height, size, stride, wave,
                             (int)(threadIdx.x & 63));
        }
        return;
    }
    uint4 *other = reinterpret_cast<uint4 *>(zbuf + (size_t)(parity ^ 1u) * pair_words);  // (pair_words is a multiple of 4)
    const uint4 ones = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    const size_t n4 = pair_words / 4, first = (size_t)(b - splat_blocks) * 256 + threadIdx.x, step = (size_t)clear_blocks * 256;
    for (size_t i = first; i < n4; i += step) other[i] = ones;
}

// Resolve.  A lane takes a strip of four horizontally adjacent output pixels x kResolveRows rows and slides down it: every
// anchor row is fetched once per strip (two aligned 16-byte loads: 8 anchors), reduced to the four pixels' horizontal minima,
// and kept for the SIZE output rows it belongs to -- one row of loads per output row instead of SIZE of them (the first form
// read its SIZE x (SIZE + 3) window afresh for every row: 19 us for 27 views at 848x480, L2-bandwidth-bound).  The four
// results of a row leave as one dword.  Rows of the anchor buffer are `stride` words apart (zbuf_stride: padded so that the
// window of any quad is covered by two aligned 16-byte loads).
constexpr int kResolveRows = 8;

template <int SIZE>
__global__ __launch_bounds__(256) void nmi_zbuf_resolve_fast_kernel(const uint32_t *__restrict__ zbuf, uint8_t *__restrict__ out, int views,
                                                                    int width, int height, int stride, const uint32_t *epoch, size_t pair_words)
{
    // requires width % 4 == 0, SIZE <= 5, stride % 4 == 0 and (SIZE == 1 or stride >= width + 4)
    if (epoch) zbuf += (size_t)(*epoch & 1u) * pair_words;  // a level's double-buffered anchors: the buffer this replay splatted into
    const int hp = height + SIZE - 1;
    const int quads = width >> 2, strips = (height + kResolveRows - 1) / kResolveRows;
    const size_t n = (size_t)views * strips * quads;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % quads), st = (int)((i / quads) % strips), s = (int)(i / ((size_t)quads * strips));
        const int py0 = st * kResolveRows;
        const uint32_t *base = zbuf + ((size_t)s * hp + py0) * stride + q * 4;
        uint8_t *dst = out + ((size_t)s * height + py0) * width + q * 4;
        uint32_t h[SIZE][4];  // horizontal minima of the last SIZE anchor rows (ring, indices static after unrolling)
        auto fetch = [&](int row, uint32_t (&m)[4]) {
            const uint4 lo = *reinterpret_cast<const uint4 *>(base + (size_t)row * stride);
            uint32_t v[8] = {lo.x, lo.y, lo.z, lo.w, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            if (SIZE > 1) {
                const uint4 hi = *reinterpret_cast<const uint4 *>(base + (size_t)row * stride + 4);
                v[4] = hi.x, v[5] = hi.y, v[6] = hi.z, v[7] = hi.w;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                m[k] = v[k];
#pragma unroll
                for (int d = 1; d < SIZE; ++d) m[k] = min(m[k], v[k + d]);
            }
        };
#pragma unroll
        for (int r = 0; r < SIZE - 1; ++r) fetch(r, h[r]);  // (anchor rows py0 .. py0 + SIZE - 2 exist: hp = height + SIZE - 1)
#pragma unroll
        for (int r = 0; r < kResolveRows; ++r) {
            if (py0 + r >= height) break;
            fetch(r + SIZE - 1, h[(r + SIZE - 1) % SIZE]);
            uint32_t best[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                best[k] = h[0][k];
#pragma unroll
                for (int j = 1; j < SIZE; ++j) best[k] = min(best[k], h[j][k]);
            }
            *reinterpret_cast<uint32_t *>(dst + (size_t)r * width) = (best[0] & 0xFFu) | ((best[1] & 0xFFu) << 8) | ((best[2] & 0xFFu) << 16) | (best[3] << 24);
        }
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

static void launch_resolve(const uint32_t *zbuf, uint8_t *out, int S, int width, int height, int size, hipStream_t stream,
                           const uint32_t *epoch = nullptr, size_t pair_words = 0)
{
    const int stride = zbuf_stride(width, size);
    const size_t nq = (size_t)S * height * ((width + 3) / 4);
    dim3 grid((unsigned)((nq + 255) / 256 < 8192 ? (nq + 255) / 256 : 8192)), block(256);
    const bool fast = (width & 3) == 0 && size <= 5 && (((uintptr_t)zbuf & 15) == 0) && (((uintptr_t)out & 3) == 0);
    if (!fast) {  // (never with an epoch: launch_level_front_points checks resolve_fast_eligible first)
        hipLaunchKernelGGL(nmi_zbuf_resolve_kernel, grid, block, 0, stream, zbuf, out, S, width, height, size, stride);
        return;
    }
    const size_t nstrips = (size_t)S * ((height + kResolveRows - 1) / kResolveRows) * (width / 4);  // one lane per strip
    grid = dim3((unsigned)((nstrips + 255) / 256 < 8192 ? (nstrips + 255) / 256 : 8192));
    switch (size) {
    case 1: hipLaunchKernelGGL(nmi_zbuf_resolve_fast_kernel<1>, grid, block, 0, stream, zbuf, out, S, width, height, stride, epoch, pair_words); break;
    case 2: hipLaunchKernelGGL(nmi_zbuf_resolve_fast_kernel<2>, grid, block, 0, stream, zbuf, out, S, width, height, stride, epoch, pair_words); break;
    case 3: hipLaunchKernelGGL(nmi_zbuf_resolve_fast_kernel<3>, grid, block, 0, stream, zbuf, out, S, width, height, stride, epoch, pair_words); break;
    case 4: hipLaunchKernelGGL(nmi_zbuf_resolve_fast_kernel<4>, grid, block, 0, stream, zbuf, out, S, width, height, stride, epoch, pair_words); break;
    default: hipLaunchKernelGGL(nmi_zbuf_resolve_fast_kernel<5>, grid, block, 0, stream, zbuf, out, S, width, height, stride, epoch, pair_words); break;
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
            hipLaunchKernelGGL(nmi_splat_kernel, dim3((unsigned)((npoints + 255) / 256)), dim3(256), 0, stream, xyz, red, npoints,
                               mvps + (size_t)s0 * 16, views, zbuf + (size_t)s0 * per_view, width, height, size, stride);
        }
    }
    launch_resolve(zbuf, out, S, width, height, size, stream);
    (void)n;
    return hipGetLastError();
}

size_t cloud_pack_bytes(long long npoints, size_t *boxes_offset)
{
    const size_t pts = ((size_t)npoints * sizeof(float4) + 255) & ~(size_t)255;
    if (boxes_offset) *boxes_offset = pts;
    return pts + (size_t)((npoints + 63) / 64) * 2 * sizeof(float4);
}

hipError_t launch_cloud_pack(const float *xyz, const float *red, long long npoints, void *packed, hipStream_t stream)
{
    size_t off = 0;
    (void)cloud_pack_bytes(npoints, &off);
    if (npoints > 0)
        hipLaunchKernelGGL(nmi_cloud_pack_kernel, dim3((unsigned)((npoints + 255) / 256)), dim3(256), 0, stream, xyz, red, npoints,
                           reinterpret_cast<float4 *>(packed), reinterpret_cast<float4 *>(static_cast<char *>(packed) + off));
    return hipGetLastError();
}

// Words of ONE of a level's two anchor buffers (a multiple of 4).
size_t level_zbuf_pair_words(int S, int width, int height, int size) { return (render_zbuf_words(S, width, height, size) + 3) & ~(size_t)3; }

// Front of a captured point-cloud level: warp stack + splat (+ the clear of the other anchor buffer) in one launch, then the
// resolve.  `zbuf` holds two buffers of level_zbuf_pair_words each, both cleared at creation; `epoch` is the device word the
// level's prep node bumps.  `mvps` holds S matrices followed by 6 planes (a, b, c, d) with every view's frustum on their
// non-negative side (level_views_bound; all zero switches the test off).  S <= 64 views.
hipError_t launch_level_front_points(const void *packed, long long npoints, const float *mvps, int S, uint32_t *zbuf, const uint32_t *epoch,
                                     uint8_t *out, int width, int height, int size, const uint8_t *frame, const float *coeffs, uint8_t *warps,
                                     int Wn, hipStream_t stream, const uint32_t *kept, const uint32_t *kept_count, int compute_units)
{
    if (S > kMaxViewsPerLaunch || !warp_lds_eligible(frame, warps, width) || !kept || !kept_count) return hipErrorInvalidValue;
    if (!((width & 3) == 0 && size <= 5 && (((uintptr_t)zbuf & 15) == 0) && (((uintptr_t)out & 3) == 0))) return hipErrorInvalidValue;
    size_t off = 0;
    (void)cloud_pack_bytes(npoints, &off);
    PackedCloud pc{reinterpret_cast<const float4 *>(packed), reinterpret_cast<const float4 *>(static_cast<const char *>(packed) + off)};
    int warp_blocks = warp_blocks_x(width) * warp_blocks_y(height) * Wn;
    // splat workers: three workgroups (12 wavefronts) per CU keep the CU's atomic path busy and leave five workgroup slots to the
    // warp blocks; never more than one wavefront per 64 points
    static const int per_cu = getenv("NMI_FRONT_WORKERS") ? atoi(getenv("NMI_FRONT_WORKERS")) : 3;
    long long splat_blocks = (long long)(compute_units > 0 ? compute_units : 256) * (per_cu > 0 ? per_cu : 3);
    if (splat_blocks > (npoints + 255) / 256) splat_blocks = (npoints + 255) / 256;
    const int stride = zbuf_stride(width, size);
    const size_t pair_words = level_zbuf_pair_words(S, width, height, size);
    static const int dbg = getenv("NMI_FRONT_DBG") ? atoi(getenv("NMI_FRONT_DBG")) : 0;  // profiling ablations (tools only): 1 no warp blocks, 2 no splat blocks, 4 no clear blocks
    if (dbg & 1) warp_blocks = 0;
    if (dbg & 2) splat_blocks = 0, npoints = 0;
    const int clear_blocks = (dbg & 4) ? 0 : 1024;
    hipLaunchKernelGGL(nmi_level_front_kernel, dim3((unsigned)(warp_blocks + splat_blocks + clear_blocks)), dim3(256), 0, stream, pc, npoints, mvps,
                       S, zbuf, pair_words, epoch, width, height, size, stride, frame, coeffs, warps, warp_blocks, (int)splat_blocks, clear_blocks,
                       kept, kept_count);
    launch_resolve(zbuf, out, S, width, height, size, stream, epoch, pair_words);
    return hipGetLastError();
}

// Six planes (a, b, c, d; a x + b y + c z + d >= 0 inside) that hold the frusta of all S views (column-major MVP matrices): per
// side of the clip volume the views' own planes (row 3 +- row j) are summed to one direction and pushed out until all 8 S
// frustum corners (the corners of clip space taken back through each inverse matrix, double precision) lie inside, plus a margin
// far above the rounding of either side's test.  A view draws only inside the hull of its corners, hence inside all six.
// A matrix that cannot be inverted, or a corner at infinity, gives six zero planes (nothing culled).
void level_views_bound(const float *mvps, int S, float out[24])
{
    // (On the host's critical path between two levels: written to cost ~2 us for 27 views -- no division per component, the
    // adjugate instead of the inverse: a corner is adj(M) (+-1, +-1, +-1, 1) up to a factor that cancels in p.xyz / p.w.)
    for (int k = 0; k < 24; ++k) out[k] = 0.0f;
    if (S <= 0 || S > 64) return;
    double corners[64 * 8][3];
    double dir[6][3] = {};
    double reach = 0.0;
    for (int s = 0; s < S; ++s) {
        double m[16];
        for (int k = 0; k < 16; ++k) m[k] = mvps[(size_t)s * 16 + k];
        // adjugate through the twelve 2x2 minors of the upper and lower row pairs (a(r, c) = m[c * 4 + r]): ~110 operations
        const double a00 = m[0], a10 = m[1], a20 = m[2], a30 = m[3], a01 = m[4], a11 = m[5], a21 = m[6], a31 = m[7];
        const double a02 = m[8], a12 = m[9], a22 = m[10], a32 = m[11], a03 = m[12], a13 = m[13], a23 = m[14], a33 = m[15];
        const double s0 = a00 * a11 - a10 * a01, s1 = a00 * a12 - a10 * a02, s2 = a00 * a13 - a10 * a03;
        const double s3 = a01 * a12 - a11 * a02, s4 = a01 * a13 - a11 * a03, s5 = a02 * a13 - a12 * a03;
        const double c5 = a22 * a33 - a32 * a23, c4 = a21 * a33 - a31 * a23, c3 = a21 * a32 - a31 * a22;
        const double c2 = a20 * a33 - a30 * a23, c1 = a20 * a32 - a30 * a22, c0 = a20 * a31 - a30 * a21;
        const double det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
        if (!(fabs(det) > 1e-300)) return;
        const double adj[4][4] = {   // adj[r][c]: row r of det * inverse
            {a11 * c5 - a12 * c4 + a13 * c3, -a01 * c5 + a02 * c4 - a03 * c3, a31 * s5 - a32 * s4 + a33 * s3, -a21 * s5 + a22 * s4 - a23 * s3},
            {-a10 * c5 + a12 * c2 - a13 * c1, a00 * c5 - a02 * c2 + a03 * c1, -a30 * s5 + a32 * s2 - a33 * s1, a20 * s5 - a22 * s2 + a23 * s1},
            {a10 * c4 - a11 * c2 + a13 * c0, -a00 * c4 + a01 * c2 - a03 * c0, a30 * s4 - a31 * s2 + a33 * s0, -a20 * s4 + a21 * s2 - a23 * s0},
            {-a10 * c3 + a11 * c1 - a12 * c0, a00 * c3 - a01 * c1 + a02 * c0, -a30 * s3 + a31 * s1 - a32 * s0, a20 * s3 - a21 * s1 + a22 * s0}};
        for (int c = 0; c < 8; ++c) {
            double p[4];
            for (int r = 0; r < 4; ++r)
                p[r] = (((c & 1) ? adj[r][0] : -adj[r][0]) + ((c & 2) ? adj[r][1] : -adj[r][1])) + (((c & 4) ? adj[r][2] : -adj[r][2]) + adj[r][3]);
            const double big = fabs(p[0]) + fabs(p[1]) + fabs(p[2]);
            if (!(fabs(p[3]) > 1e-30 * big) || !(fabs(p[3]) > 0.0)) return;   // a corner at infinity (also false for NaN)
            const double iw = 1.0 / p[3];
            double *q = corners[s * 8 + c];
            q[0] = p[0] * iw, q[1] = p[1] * iw, q[2] = p[2] * iw;
            const double size = fabs(q[0]) + fabs(q[1]) + fabs(q[2]);
            if (!(size < 1e30)) return;   // (false for NaN too)
            reach = size > reach ? size : reach;
            // a corner must lie in front of its own view (w_clip > 0): a matrix with the volume turned inside out is not a camera
            if (!(m[3] * q[0] + m[7] * q[1] + m[11] * q[2] + m[15] > 0.0)) return;
        }
        // plane k of this view: row 3 + row j (k even) or row 3 - row j (k odd), j = k / 2.  The views' planes are added up as
        // they are (same projection, so alike in length): any common direction is valid, a good one is tight.
        for (int k = 0; k < 6; ++k) {
            const int j = k >> 1;
            const double sg = (k & 1) ? -1.0 : 1.0;
            dir[k][0] += m[3] + sg * m[j], dir[k][1] += m[7] + sg * m[4 + j], dir[k][2] += m[11] + sg * m[8 + j];
        }
    }
    double n[6][3], least[6];
    for (int k = 0; k < 6; ++k) {
        const double len2 = dir[k][0] * dir[k][0] + dir[k][1] * dir[k][1] + dir[k][2] * dir[k][2];
        if (!(len2 > 1e-60) || !(len2 < 1e300)) return;  // the views face every which way (or a matrix is not finite): no common side
        const double il = 1.0 / sqrt(len2);
        n[k][0] = dir[k][0] * il, n[k][1] = dir[k][1] * il, n[k][2] = dir[k][2] * il;
        least[k] = 1e300;
    }
    for (int i = 0; i < S * 8; ++i) {   // one pass over the corners, six running minima
        const double x = corners[i][0], y = corners[i][1], z = corners[i][2];
        for (int k = 0; k < 6; ++k) {
            const double v = n[k][0] * x + n[k][1] * y + n[k][2] * z;
            least[k] = v < least[k] ? v : least[k];
        }
    }
    float planes[24];
    for (int k = 0; k < 6; ++k) {
        planes[4 * k] = (float)n[k][0], planes[4 * k + 1] = (float)n[k][1], planes[4 * k + 2] = (float)n[k][2];
        planes[4 * k + 3] = (float)(-least[k] + 1e-4 * reach + 1e-6);
    }
    for (int k = 0; k < 24; ++k) out[k] = planes[k];
}

bool level_front_eligible(const void *frame, const void *warps, int width, int S) { return S <= kMaxViewsPerLaunch && warp_lds_eligible(frame, warps, width); }
bool level_points_double_buffered(int width, int size) { return (width & 3) == 0 && size <= 5; }

}  // namespace nmi
