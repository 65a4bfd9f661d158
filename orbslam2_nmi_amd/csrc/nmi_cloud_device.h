// nmi_cloud_device.h -- per-point and per-box device arithmetic of the point-cloud renderer (nmi_producers.hip: one global
// atomicMin per point and view, then a resolve pass), kept apart from the kernels so that every kernel that splats or culls
// evaluates a point with the same expressions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nmi_warp_device.h"

namespace nmi {

// The cloud as a level keeps it (nmi_level_create): one 16-byte record per point -- x, y, z, red -- so that a lane fetches its
// point with ONE load instead of four 4-byte loads 12 bytes apart, and the bounding box of every wavefront's 64 points,
// computed once (the cloud of a level does not change), instead of 36 cross-lane reductions per wavefront and launch.
struct PackedCloud {
    const float4 *points;  // [n]
    const float4 *boxes;   // [2 * ceil(n / 64)]: lo.xyz, hi.xyz of each wavefront's points
};

namespace {

// One point under one view (column-major MVP `m`, any address space the caller has it in): clip, perspective divide, window
// transform, 24-bit depth, and the sprite's ANCHOR -- its lowest-left pixel in the buffer padded by size - 1 on each axis.
// Only the anchor is written by either renderer: two points with the same anchor have the same footprint, so the farther
// one would lose on every pixel anyway; a pixel's value is the minimum over the size^2 anchors whose sprites cover it.
// Returns false for a clipped point or an anchor outside the padded buffer.
// The view loops are bound by vector-instruction issue, so this is written for instruction count: components in pairs through
// the packed fp32 instructions ((cx, cy) and (cz, cw) -- the matrix is column-major, so each pair's coefficients are adjacent
// floats; IEEE results per element, never fused: -ffp-contract=off), the six clip comparisons as one maximum of magnitudes, and
// the reciprocal by warp_rcp (nmi_warp_device.h: bit-identical to the division, checked for every float).  Values are those of
// the plain expressions in the comments, bit for bit (the fp32 twin oracle/render_oracle_np.py checks them).
template <typename M>
__device__ __forceinline__ bool splat_anchor(const M &m, float x, float y, float z, uint32_t colour, int width, int height, int size, int &ax,
                                             int &ay, uint32_t &frag)
{
    // glm mat4 * vec4: c_r = (m[r]*x + m[4+r]*y) + (m[8+r]*z + m[12+r])
    const v2f X = {x, x}, Y = {y, y}, Z = {z, z};
    const v2f cxy = (v2f{m[0], m[1]} * X + v2f{m[4], m[5]} * Y) + (v2f{m[8], m[9]} * Z + v2f{m[12], m[13]});
    const v2f czw = (v2f{m[2], m[3]} * X + v2f{m[6], m[7]} * Y) + (v2f{m[10], m[11]} * Z + v2f{m[14], m[15]});
    const float cw = czw.y;
    // point clipping:  !(cw > 0) || cx < -cw || cx > cw || cy < -cw || cy > cw || cz < -cw || cz > cw.  The six comparisons are
    // "the largest magnitude exceeds cw" (fmaxf passes over a NaN operand, and a comparison with NaN is false, exactly as there)
    if (!(cw > 0.0f)) return false;
    if (fmaxf(fmaxf(fabsf(cxy.x), fabsf(cxy.y)), fabsf(czw.x)) > cw) return false;
    // the perspective divide as one (correctly rounded) reciprocal and three products:
    //   xw = (cx * iw * 0.5f + 0.5f) * width, yw likewise with height, zw = cz * iw * 0.5f + 0.5f
    const float iw = warp_rcp(cw);
    const v2f win = ((cxy * iw) * 0.5f + 0.5f) * v2f{(float)width, (float)height};
    const float zw = czw.x * iw * 0.5f + 0.5f;
    const uint32_t depth = (uint32_t)(zw * 16777215.0f + 0.5f);
    frag = (depth << 8) | colour;
    // odd sizes are centred on floor(xw) + 0.5, even sizes on floor(xw + 0.5); the anchor is the sprite's lowest-left pixel in the
    // buffer padded by size - 1:  x0 + size - 1
    const float r = (size & 1) ? 0.0f : 0.5f;
    const int back = (size & 1) ? (size - 1) / 2 : size / 2;
    ax = (int)floorf(win.x + r) - back + size - 1;   // (win.x + 0.0f has win.x's bits unless win.x is -0: floor gives -0 -> 0 either way)
    ay = (int)floorf(win.y + r) - back + size - 1;
    return ((uint32_t)ax < (uint32_t)(width + size - 1)) & ((uint32_t)ay < (uint32_t)(height + size - 1));
}

// Can the box [lo, hi] reach the view whose matrix has the columns c0..c3?  For each of the six clip planes the box corner
// farthest along the plane's normal, with a margin that covers the rounding of this test and of the per-point test above;
// a view with that corner outside one plane cannot receive anything from inside the box.  The clip tests are affine in the
// position, so the test is exact-conservative.
__device__ __forceinline__ bool box_outside_view(float4 c0, float4 c1, float4 c2, float4 c3, float lox, float loy, float loz, float hix,
                                                 float hiy, float hiz)
{
    const float ax = fmaxf(fabsf(lox), fabsf(hix)), ay = fmaxf(fabsf(loy), fabsf(hiy)), az = fmaxf(fabsf(loz), fabsf(hiz));
    bool outside = false;
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
            outside = outside | (best < -2.0f * e);   // (bitwise: no branch per plane)
        }
    }
    return outside;
}

// The box against the six planes that hold EVERY view's frustum (level_views_bound; `pl` = 24 floats at a wavefront-uniform
// address: scalar loads).  True: no view can receive anything from inside the box.
__device__ __forceinline__ bool box_outside_bound(const float *__restrict__ pl, float4 lo, float4 hi)
{
    bool out = false;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float a = pl[4 * k], b = pl[4 * k + 1], c = pl[4 * k + 2], d = pl[4 * k + 3];
        // the largest value the plane function takes in the box (a comparison with a NaN operand is false: kept)
        out = out | ((a * (a >= 0.0f ? hi.x : lo.x) + b * (b >= 0.0f ? hi.y : lo.y)) + (c * (c >= 0.0f ? hi.z : lo.z) + d) < 0.0f);
    }
    return out;
}

}  // namespace
}  // namespace nmi
