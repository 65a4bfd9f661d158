// nmi_cloud_device.h -- device code shared by the two point-cloud renderers: the scatter form (nmi_producers.hip: one global
// atomicMin per point and view, then a resolve pass) and the tiled form of a captured level (nmi_cloud_tiles.hip: wavefronts
// binned to 64 x 64 tiles, depth test in LDS, resolve fused).  Both evaluate a point with splat_anchor below, so they agree
// bit for bit by construction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
template <typename M>
__device__ __forceinline__ bool splat_anchor(const M &m, float x, float y, float z, uint32_t colour, int width, int height, int size, int &ax,
                                             int &ay, uint32_t &frag)
{
    // glm mat4 * vec4: (m0*x + m1*y) + (m2*z + m3*1), component-wise
    const float cx = (m[0] * x + m[4] * y) + (m[8] * z + m[12]);
    const float cy = (m[1] * x + m[5] * y) + (m[9] * z + m[13]);
    const float cz = (m[2] * x + m[6] * y) + (m[10] * z + m[14]);
    const float cw = (m[3] * x + m[7] * y) + (m[11] * z + m[15]);
    if (!(cw > 0.0f) || cx < -cw || cx > cw || cy < -cw || cy > cw || cz < -cw || cz > cw) return false;  // point clipping
    // the perspective divide as one (correctly rounded) reciprocal and three products: a third of the three divisions' cost
    const float iw = 1.0f / cw;
    const float xw = (cx * iw * 0.5f + 0.5f) * (float)width;
    const float yw = (cy * iw * 0.5f + 0.5f) * (float)height;
    const float zw = cz * iw * 0.5f + 0.5f;
    const uint32_t depth = (uint32_t)(zw * 16777215.0f + 0.5f);
    frag = (depth << 8) | colour;
    int x0, y0;
    if (size & 1) {
        x0 = (int)floorf(xw) - (size - 1) / 2;
        y0 = (int)floorf(yw) - (size - 1) / 2;
    } else {
        x0 = (int)floorf(xw + 0.5f) - size / 2;
        y0 = (int)floorf(yw + 0.5f) - size / 2;
    }
    ax = x0 + size - 1, ay = y0 + size - 1;
    return ax >= 0 && ax < width + size - 1 && ay >= 0 && ay < height + size - 1;
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
            outside = outside || best < -2.0f * e;
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
        out = out || (a * (a >= 0.0f ? hi.x : lo.x) + b * (b >= 0.0f ? hi.y : lo.y)) + (c * (c >= 0.0f ? hi.z : lo.z) + d) < 0.0f;
    }
    return out;
}

}  // namespace
}  // namespace nmi
