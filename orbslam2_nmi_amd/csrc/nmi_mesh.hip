// nmi_mesh.hip -- render-stack producer for textured meshes (nmi_prop_RENDER 1, the reference's default,
// Thirdparty/Localization/allProperties.hpp:41): Rendering<1>::renderToTextureOnGPU, rendering.hpp:530-630 with
// shaders/ShadingWithTexture.* -- glDrawArrays(GL_TRIANGLES) of the OBJ's expanded vertex / uv arrays
// (objloader.cpp:140-224), GL_CULL_FACE (back faces, counter-clockwise front; rendering.hpp:300), depth test GL_LESS,
// fragment colour = 0.299 r + 0.587 g + 0.114 b of the texture sample (fragment shader :16) with GL_REPEAT wrap, GL_LINEAR
// magnification and GL_LINEAR_MIPMAP_LINEAR minification (texture.cpp:88-92).  No OpenGL.
//
// What is computed is the OpenGL 3.3 pipeline in fp32 as the specification words it (pixel centres at +0.5, top-left fill
// rule, perspective-correct interpolation, isotropic level of detail from the per-pixel uv differences, near-plane clipping
// in clip space before the divide); a real driver rasterises in fixed point and is free in its LOD approximation, so
// parity with one is unpinned.  oracle/mesh_oracle_np.py restates the same arithmetic in numpy (test infrastructure).
//
// HOW (round 3: a binned, deferred rasteriser).  The first two rounds gave a lane to every triangle and let it walk its
// pixels alone, shading each covered one and resolving visibility with a device atomicMin on a 44 MB depth|luma buffer:
// 0.34-0.74 ms per 27 views, "arithmetic, not memory and not atomics" -- ~40 instructions of edge functions per
// bounding-box pixel and ~350 of attributes, LOD and texture per COVERED fragment, on lanes that mostly sat idle.  Now:
//   binning pass          one lane per triangle; blocks of 256 triangles are frustum-culled per view.  It only decides where a
//                         triangle goes: one whose pixel box is at most kSmallBox pixels is rasterised right there
//                         (coverage + depth only) into a 64-bit key buffer in memory; a larger one is appended to the BIN of
//                         every 64 x 64 screen tile its box touches (wave-aggregated appends: neighbours in the mesh land in
//                         the same tile); one that crosses the near plane goes to a small queue for the clip kernel.
//                         Small meshes: ONE kernel, nmi_mesh_bin_kernel, a workgroup per block of triangles and share of the
//                         views.  Larger ones: TWO -- nmi_mesh_cull_kernel lists the (block, view) pairs in reach of each
//                         other, nmi_mesh_bin_pairs_kernel deals them out to a bounded number of worker workgroups, every
//                         unit of work one view deep.
//   nmi_mesh_clip_kernel  clips a crossing triangle (Sutherland-Hodgman in clip space) and treats the 1 or 2 pieces the same way.
//   nmi_mesh_tile_kernel  one 512-lane workgroup per (view, tile): the tile's 4096 visibility keys live in LDS (32 KiB, no
//                         device atomics, no clear pass), lane j sets up bin entry j into an LDS record, the 512 lanes deal
//                         out the pixel pairs of the records' boxes in uniform chunks and resolve visibility with ds_min_u64,
//                         then every pixel is shaded ONCE, by the triangle that won it (deferred: attributes, LOD and texture
//                         are not spent on hidden or uncovered pixels, and every lane has a pixel), and leaves as a byte of
//                         the render -- four pixels side by side per lane and step, one dword store.  Two builds: 255 bin
//                         entries per tile (two workgroups per CU) or 127 (nmi_mesh_tile_small_kernel: three).
// Visibility key = depth24 << 40 | triangle << 10 | piece << 9 | slot: smaller depth wins, equal depths go to the triangle
// drawn first (GL_LESS keeps the earlier fragment of glDrawArrays' order) -- deterministic whatever the order of the
// atomics.  Pixels won through the memory buffer (small triangles, bin overflow) carry slot 0x1FF and set their triangle up
// on the fly.  The memory buffer and the bin counters are left clean by the tile kernel (it re-clears what it read), so a
// render has no clear pass; tiles without small triangles never touch the buffer.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "nmi_kernels.h"
#include "nmi_warp_device.h"

namespace nmi {

namespace {

constexpr int kMaxViewsPerLaunch = 64;
constexpr int kTile = 64;
constexpr int kBinMax = 256;           // largest bin stride; slots 0 .. 254 are usable
constexpr int kTileThreads = 512;      // 8 wavefronts per tile, two tiles resident per CU (60 KiB of LDS each)
constexpr uint32_t kNoSlot = 0x1FFu;   // "no LDS record": the pixel was won through the memory buffer
constexpr int kSmallBox = 16;
constexpr unsigned long long kEmptyKey = ~0ull;

struct MeshTexture {
    const float *luma;   // all levels, level l at luma + off[l], row-major, row 0 = v 0
    int levels;
    int w[16], h[16];
    float inv_w[16], inv_h[16];  // 1 / w, 1 / h (for the wrap)
    long long off[16];
};

// x mod n for an integer-valued x (|x| < 2^24) and 0 < n < 2^24, all in fp32: the quotient estimate may be one off, the
// two corrections make the result the true remainder in [0, n) -- what an integer modulo (20+ instructions, four per
// texture sample) would give.
__device__ __forceinline__ int wrap_index(float x, float n, float inv_n)
{
    float r = x - floorf(x * inv_n) * n;
    r = r < 0.0f ? r + n : r;
    r = r >= n ? r - n : r;
    return (int)r;
}

// One level of the pyramid as the sampler wants it: 8 words, kept in LDS by the tile kernel (a per-lane level index into the
// kernel arguments would be a chain of dependent global loads in front of every texel fetch).
struct TexLevel {
    float w, h, inv_w, inv_h;
    uint32_t wi, hi;
    uint32_t off_lo, off_hi;  // offset of the level in `luma`, in texels
};

__device__ __forceinline__ void tex_level_fill(const MeshTexture &t, int l, TexLevel *out)
{
    TexLevel L;
    L.w = (float)t.w[l], L.h = (float)t.h[l], L.inv_w = t.inv_w[l], L.inv_h = t.inv_h[l];
    L.wi = (uint32_t)t.w[l], L.hi = (uint32_t)t.h[l];
    L.off_lo = (uint32_t)((unsigned long long)t.off[l] & 0xFFFFFFFFull), L.off_hi = (uint32_t)((unsigned long long)t.off[l] >> 32);
    *out = L;
}

// GL_LINEAR sample of one level at (u, v) with GL_REPEAT.  Written for instruction count (the shading loop is bound by
// instruction issue): the common texel -- both taps of both rows inside the level -- takes its indices by plain conversion and its
// taps as two 8-byte loads at 32-bit offsets from the pyramid's base; only a sample on the level's border wraps (wrap_index) and
// fetches four single taps.  Values are those of the plain expressions:
//   x = u w - 0.5, y = v h - 0.5;  i0 = floor(x) mod w, i1 = (i0 + 1) mod w, rows likewise;  a = t00 + (t10 - t00) fx, b = t01 + (t11 - t01) fx;  a + (b - a) fy
__device__ __forceinline__ float tex_bilinear(const float *__restrict__ luma, const TexLevel &L, v2f uv)
{
    const v2f xy = uv * v2f{L.w, L.h} - 0.5f;
    const v2f fl = {floorf(xy.x), floorf(xy.y)};
    const v2f fr = xy - fl;
    const int xi = (int)fl.x, yi = (int)fl.y;
    const char *base = reinterpret_cast<const char *>(luma);   // (a pyramid is far below 4 GB: 32-bit byte offsets, scalar base)
    float t00, t10, t01, t11;   // (plain scalars: vectors assigned on two paths end up in a promoted stack slot, i.e. in LDS)
    if (__builtin_expect(((uint32_t)xi < L.wi - 1u) & ((uint32_t)yi < L.hi - 1u), 1)) {
        const uint32_t o = (L.off_lo + (uint32_t)yi * L.wi + (uint32_t)xi) * 4u;
        const float2 top = *reinterpret_cast<const float2 *>(base + o), bot = *reinterpret_cast<const float2 *>(base + (o + L.wi * 4u));
        t00 = top.x, t10 = top.y, t01 = bot.x, t11 = bot.y;
    } else {
        const uint32_t i0 = (uint32_t)wrap_index(fl.x, L.w, L.inv_w), j0 = (uint32_t)wrap_index(fl.y, L.h, L.inv_h);  // GL_REPEAT
        const uint32_t i1 = i0 + 1 == L.wi ? 0 : i0 + 1, j1 = j0 + 1 == L.hi ? 0 : j0 + 1;
        const uint32_t r0 = L.off_lo + j0 * L.wi, r1 = L.off_lo + j1 * L.wi;
        t00 = *reinterpret_cast<const float *>(base + (r0 + i0) * 4u), t10 = *reinterpret_cast<const float *>(base + (r0 + i1) * 4u);
        t01 = *reinterpret_cast<const float *>(base + (r1 + i0) * 4u), t11 = *reinterpret_cast<const float *>(base + (r1 + i1) * 4u);
    }
    const v2f ab = v2f{t00, t01} + (v2f{t10, t11} - v2f{t00, t01}) * fr.x;
    return ab.x + (ab.y - ab.x) * fr.y;
}

__device__ __forceinline__ bool edge_owner(float ex, float ey)
{
    // top-left rule for a counter-clockwise triangle in y-up window coordinates: an edge owns the pixels exactly on it
    // when it is a left edge (going down) or a top edge (horizontal, going left)
    return (ey < 0.0f) | ((ey == 0.0f) & (ex < 0.0f));
}

// One triangle seen by one view: what coverage and depth need.
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
        // glm mat4 * vec4: (m0*x + m1*y) + (m2*z + m3*1), component-wise
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
// order (3 or 4 of them), new corners interpolated from the inside corner towards the outside one, so that two triangles
// sharing an edge cut it at the same point.
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
    // (bitwise & and | on purpose, here and below: the short-circuit forms compile to one branch per comparison)
    if (!((cw[0] > 0.0f) & (cw[1] > 0.0f) & (cw[2] > 0.0f))) return false;  // (a corner on or behind the eye plane survives near clipping only with a degenerate matrix)
    if (((cx[0] < -cw[0]) & (cx[1] < -cw[1]) & (cx[2] < -cw[2])) | ((cx[0] > cw[0]) & (cx[1] > cw[1]) & (cx[2] > cw[2])) |
        ((cy[0] < -cw[0]) & (cy[1] < -cw[1]) & (cy[2] < -cw[2])) | ((cy[0] > cw[0]) & (cy[1] > cw[1]) & (cy[2] > cw[2])) |
        ((cz[0] < -cw[0]) & (cz[1] < -cw[1]) & (cz[2] < -cw[2])) | ((cz[0] > cw[0]) & (cz[1] > cw[1]) & (cz[2] > cw[2])))
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
    if ((t.x_lo > t.x_hi) | (t.y_lo > t.y_hi)) return false;
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

// Corner arrays of triangle `tri`.
__device__ __forceinline__ void load_tri(const float *__restrict__ xyz, const float *__restrict__ uv, long long tri, float (&px)[3], float (&py)[3],
                                         float (&pz)[3], float (&tu)[3], float (&tv)[3])
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        px[k] = xyz[(tri * 3 + k) * 3], py[k] = xyz[(tri * 3 + k) * 3 + 1], pz[k] = xyz[(tri * 3 + k) * 3 + 2];
        tu[k] = uv[(tri * 3 + k) * 2], tv[k] = uv[(tri * 3 + k) * 2 + 1];
    }
}

// (triangle, piece) as view `m` sees it: piece 0 of a triangle wholly behind... in front of the near plane is the triangle
// itself; of one that crosses it, pieces 0 and 1 are the triangles of its clipped polygon.  Every kernel sets a triangle up
// through this one function, so which of them handles a pixel does not change its value.
__device__ __forceinline__ bool setup_piece(const float *__restrict__ xyz, const float *__restrict__ uv, long long tri, int sub,
                                            const float *__restrict__ m, int width, int height, TriView &t, float (&su)[3], float (&sv)[3])
{
    float px[3], py[3], pz[3], cx[3], cy[3], cz[3], cw[3], d[3];
    load_tri(xyz, uv, tri, px, py, pz, su, sv);
    const int n_in = tri_clip_coords(m, px, py, pz, cx, cy, cz, cw, d);
    if (n_in == 0) return false;
    if (n_in < 3) {
        ClipPoly P;
        const float tu[3] = {su[0], su[1], su[2]}, tv[3] = {sv[0], sv[1], sv[2]};
        tri_clip_poly(cx, cy, cz, cw, d, tu, tv, P);
        if (sub + 3 > P.n) return false;
        poly_corners(P, sub, cx, cy, cz, cw, su, sv);
    } else if (sub) {
        return false;
    }
    return tri_setup(cx, cy, cz, cw, width, height, t);
}

// Coverage (top-left rule) and depth of the pixel whose centre is (fxp, fyp).  Returns true with the 24-bit depth if the
// triangle produces a fragment there.
__device__ __forceinline__ bool tri_cover(const TriView &t, float fxp, float fyp, uint32_t &depth)
{
    float bary[3];
    bool inside = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int a = (k + 1) % 3;
        bary[k] = (t.ex[k] * (fyp - t.yw[a]) - t.ey[k] * (fxp - t.xw[a])) * t.inv_area;
        // bary > 0, or bary == 0 on an edge that owns its pixels: ">= 0" there (true for -0 too, like "== 0").  own[k] is the
        // same for all lanes of the callers' inner loops, so this is two compares and a scalar select, no per-lane logic.
        inside = inside & (t.own[k] ? bary[k] >= 0.0f : bary[k] > 0.0f);
    }
    const float z = (bary[0] * t.zw[0] + bary[1] * t.zw[1]) + bary[2] * t.zw[2];
    depth = min((uint32_t)(z * 16777215.0f + 0.5f), 0xFFFFFFu);  // (at z = 1 the fp32 sum rounds up to 2^24: the far plane is the largest depth)
    return inside & (z >= 0.0f) & (z <= 1.0f);  // (depth clipping; false for NaN)
}

// Perspective-correct attributes as three planes over the window: S = u/w, R = v/w, Q = 1/w, each G(x, y) = G0 +
// Gx (x - xr) + Gy (y - yr) about the centre (xr, yr) of the first pixel of the triangle's box.  The barycentric weights are
// affine in the pixel position, so the planes carry the same interpolation as weights evaluated per pixel, at 6 operations
// per attribute instead of 18 + 9; u = S / Q, v = R / Q.
struct Planes {   // S = u/w and R = v/w as pairs (the shader runs them through the packed fp32 instructions), Q = 1/w
    float xr, yr;
    v2f sr0, srx, sry;   // (s0, r0), (sx, rx), (sy, ry)
    float q0, qx, qy;
};

__device__ __forceinline__ void tri_planes(const TriView &t, const float (&tu)[3], const float (&tv)[3], Planes &P)
{
    P.xr = (float)t.x_lo + 0.5f;
    P.yr = (float)t.y_lo + 0.5f;
    float b[3], bx[3], by[3], s[3], r[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int a = (k + 1) % 3;
        b[k] = (t.ex[k] * (P.yr - t.yw[a]) - t.ey[k] * (P.xr - t.xw[a])) * t.inv_area;
        bx[k] = (-t.ey[k]) * t.inv_area;
        by[k] = t.ex[k] * t.inv_area;
        s[k] = tu[k] * t.iw[k];
        r[k] = tv[k] * t.iw[k];
    }
    P.sr0 = v2f{(b[0] * s[0] + b[1] * s[1]) + b[2] * s[2], (b[0] * r[0] + b[1] * r[1]) + b[2] * r[2]};
    P.srx = v2f{(bx[0] * s[0] + bx[1] * s[1]) + bx[2] * s[2], (bx[0] * r[0] + bx[1] * r[1]) + bx[2] * r[2]};
    P.sry = v2f{(by[0] * s[0] + by[1] * s[1]) + by[2] * s[2], (by[0] * r[0] + by[1] * r[1]) + by[2] * r[2]};
    P.q0 = (b[0] * t.iw[0] + b[1] * t.iw[1]) + b[2] * t.iw[2];
    P.qx = (bx[0] * t.iw[0] + bx[1] * t.iw[1]) + bx[2] * t.iw[2];
    P.qy = (by[0] * t.iw[0] + by[1] * t.iw[1]) + by[2] * t.iw[2];
}

// The fragment shader: uv at the pixel and at its right and upper neighbours (one reciprocal of Q each), level of detail,
// GL_LINEAR / GL_LINEAR_MIPMAP_LINEAR sample of the luma pyramid -> grey level 0..255.
__device__ __forceinline__ uint32_t shade_pixel(const Planes &P, const float *__restrict__ luma_base, const TexLevel &base, const TexLevel *levels,
                                            int n_levels, float fxp, float fyp)
{   // (`base` = levels[0] held by the caller in registers -- for the tile kernel in scalar ones, straight from its arguments)
    // S = (s0 + sx dx) + sy dy, R and Q likewise; u = S / Q at the pixel, at its right neighbour (S + sx, Q + qx) and at its upper one.
    // S and R go as a pair through the packed fp32 instructions; the three reciprocals are warp_rcp's (the division's bits).
    const float dx = fxp - P.xr, dy = fyp - P.yr;
    const v2f SR = (P.sr0 + P.srx * dx) + P.sry * dy;
    const float Q = (P.q0 + P.qx * dx) + P.qy * dy;
    const float Qx = Q + P.qx, Qy = Q + P.qy;
    float iq, iqx, iqy;
    if (__builtin_expect(warp_rcp_ok(Q) & warp_rcp_ok(Qx) & warp_rcp_ok(Qy), 1))
        iq = warp_rcp_fast(Q), iqx = warp_rcp_fast(Qx), iqy = warp_rcp_fast(Qy);
    else
        iq = 1.0f / Q, iqx = 1.0f / Qx, iqy = 1.0f / Qy;
    const v2f uv = SR * iq;
    const v2f uvx = (SR + P.srx) * iqx, uvy = (SR + P.sry) * iqy;
    const v2f twh = {base.w, base.h};
    const v2f ddx = (uvx - uv) * twh, ddy = (uvy - uv) * twh;   // (du/dx tw, dv/dx th), (du/dy tw, dv/dy th)
    // lambda = log2(rho), rho = the longer of the two footprint axes: log2 of a square root is half the log2 of the square
    const v2f sqx = ddx * ddx, sqy = ddy * ddy;
    const float rho2 = fmaxf(sqx.x + sqx.y, sqy.x + sqy.y);
    float luma;
    if (!(rho2 > 1.0f)) {
        // magnification (lambda = log2(rho2) / 2 <= 0, or NaN): GL_LINEAR on the base level -- no logarithm, no level arithmetic,
        // no look-up in the level table
        luma = tex_bilinear(luma_base, base, uv);
    } else {
        // minification: GL_LINEAR_MIPMAP_LINEAR between levels floor(lambda) and the next (lambda clamped to the pyramid)
        const float lambda = 0.5f * log2f(rho2);
        const float lc = fminf(fmaxf(lambda, 0.0f), (float)(n_levels - 1));
        const int l0 = (int)floorf(lc), l1 = min(l0 + 1, n_levels - 1);
        const float f = lc - (float)l0;
        luma = tex_bilinear(luma_base, levels[l0], uv);
        if (lambda > 0.0f) {
            const float s1 = tex_bilinear(luma_base, levels[l1], uv);
            luma = luma + (s1 - luma) * f;
        }
    }
    return (uint32_t)(fminf(fmaxf(luma, 0.0f), 1.0f) * 255.0f + 0.5f);
}

__device__ __forceinline__ unsigned long long make_key(uint32_t depth, unsigned long long id /* triangle << 1 | piece */, uint32_t slot)
{
    return ((unsigned long long)depth << 40) | (id << 9) | (unsigned long long)slot;
}

// Geometry of the bins.
struct BinGrid {
    uint32_t *bins;        // [views][tiles][stride] entries: triangle << 1 | piece
    uint32_t *state;       // [views][tiles][2]: entries appended (may exceed the capacity: the excess was rasterised directly) /
                           //                    "the memory buffer holds keys inside this tile"; both zero between renders
    unsigned long long *zbuf;  // [views][height][width] keys of the direct path, all ones between renders
    int stride, cap;       // entries per bin allocated / usable (cap <= stride, cap <= 511; 0: everything goes the direct way)
    int tiles_x, tiles_y;
    int dbg;               // profiling ablations (NMI_MESH_DBG, tools only): bit 0 no shading, 1 no visibility sweep, 2 no set-up, 3 no tile work at all,
                           // 4 no key updates, 5 no visibility steps, 6 no record search; bin kernel: 7 stop after the frustum test, 8 no appends
};

// The direct path: coverage + depth of the triangle's pixels inside [x0, x1] x [y0, y1], visibility by a 64-bit atomicMin in
// memory.  Small triangles (a handful of pixels: nothing to share out) and whatever did not fit into a bin.
__device__ __forceinline__ void raster_direct(const TriView &t, int x0, int x1, int y0, int y1, unsigned long long id, const BinGrid &g, int s,
                                              int width, int height)
{
    unsigned long long *img = g.zbuf + (size_t)s * width * height;
    const int tiles = g.tiles_x * g.tiles_y;
    for (int ty = y0 / kTile; ty <= y1 / kTile; ++ty)
        for (int tx = x0 / kTile; tx <= x1 / kTile; ++tx) g.state[2 * ((size_t)s * tiles + ty * g.tiles_x + tx) + 1] = 1u;
    for (int yy = y0; yy <= y1; ++yy)
        for (int xx = x0; xx <= x1; ++xx) {
            uint32_t depth;
            if (tri_cover(t, (float)xx + 0.5f, (float)yy + 0.5f, depth)) atomicMin(&img[(size_t)yy * width + xx], make_key(depth, id, kNoSlot));
        }
}

// View-frustum culling per block of 256 lanes.  Each lane brings the bounding box of its own triangle (+inf / -inf for a
// lane without one); the triangles of a block are neighbours in the map's own order (nmi_sort_triangles), so their common
// box is small.  A view whose clip planes put all eight corners of that box beyond ONE plane -- by a margin that covers the
// rounding of both this test and the per-triangle test that follows -- cannot receive anything from the block, which then
// skips that view's 256 transforms.  The clip tests are affine in the position, so the box test is exact-conservative:
// results do not change.  Views [v_first, v_end) are tested; on return (after a barrier) beyond[s] != 0 means "skip view s".
__device__ __forceinline__ void block_frustum_cull(const float *m_all, int v_first, int v_end, const float (&lo_in)[3], const float (&hi_in)[3],
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
    for (int t = threadIdx.x; t < (v_end - v_first) * 8; t += blockDim.x) {
        const int s = v_first + (t >> 3), c = t & 7;
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

struct ClipItem {
    unsigned long long tri;
    uint32_t view, pad;
};

// Appending `id` to bin `want`, in two halves so that a caller with several appends pays one memory round trip for all of them:
// bin_reserve groups the lanes of the wavefront by bin -- the lanes that want the same bin share one counter update, and the
// updates of the different bins of a call are ONE atomic instruction -- and ISSUES that update; bin_commit, called after all the
// reservations of a batch, reads the answer and stores the entry.  (A device atomic that returns takes ~2 us here; a wavefront of
// neighbouring triangles touches several bins per view and two to four tiles each: one round trip per tile step and view was most
// of the bin kernel's time.)  `want` < 0: this lane has nothing to append (it still takes part in both halves).
struct BinTicket {
    int leader;       // lowest lane of this lane's group
    uint32_t rank;    // this lane's position in the group
    uint32_t base;    // the counter before the group's update (valid in the leader, once the atomic has returned)
};

__device__ __forceinline__ BinTicket bin_reserve(const BinGrid &g, int want)
{
    const int lane = (int)(threadIdx.x & 63);
    const unsigned long long below = (1ull << lane) - 1ull;
    BinTicket t{lane, 0u, 0u};
    uint32_t size = 1;
    unsigned long long todo = __ballot(want >= 0);
    while (todo) {  // wavefront-uniform, one round per distinct bin, no memory access
        const int first = __builtin_ctzll(todo);
        const int b = __shfl(want, first, 64);
        const unsigned long long same = __ballot(want == b);
        if (want == b) t.leader = first, t.rank = (uint32_t)__popcll(same & below), size = (uint32_t)__popcll(same);
        todo &= ~same;
    }
    if (want >= 0 && lane == t.leader) t.base = atomicAdd(&g.state[2 * (size_t)want], size);
    return t;
}

// Returns the slot, or -1 when the bin is full (or want < 0).
__device__ __forceinline__ int bin_commit(const BinGrid &g, int want, uint32_t id, const BinTicket &t)
{
    const uint32_t base = (uint32_t)__shfl((int)t.base, t.leader, 64);
    if (want < 0) return -1;
    const uint32_t at = base + t.rank;
    if (at >= (uint32_t)g.cap) return -1;
    g.bins[(size_t)want * g.stride + at] = id;
    return (int)at;
}

}  // namespace

// One triangle per lane (tri, corners px/py/pz; `valid` false: the lane only takes part in the wavefront's appends) through view s
// (matrix m, 16 floats in any address space): a small pixel box is rasterised right here, a large one goes to the bins of the tiles
// it touches, one that crosses the near plane to the clip queue.  Wavefront-level code: no barrier, no LDS.
template <typename M>
__device__ __forceinline__ void bin_one_view(const M *m, int s, long long tri, bool valid, const float (&px)[3], const float (&py)[3],
                                             const float (&pz)[3], int width, int height, const BinGrid &g, ClipItem *__restrict__ clipq,
                                             unsigned long long *__restrict__ clip_state, unsigned long long clip_cap)
{
    const int tiles = g.tiles_x * g.tiles_y;
        TriView t;
        bool large = false;
        if (valid) {
            float cx[3], cy[3], cz[3], cw[3], d[3];
            const int n_in = tri_clip_coords(m, px, py, pz, cx, cy, cz, cw, d);
            if (n_in == 3) {
                if (tri_setup(cx, cy, cz, cw, width, height, t)) {
                    const int bw = t.x_hi - t.x_lo + 1, bh = t.y_hi - t.y_lo + 1;
                    large = bw * bh > kSmallBox && g.cap > 0 && !(g.dbg & 256);
                    if (!large && !(g.dbg & 256)) raster_direct(t, t.x_lo, t.x_hi, t.y_lo, t.y_hi, (unsigned long long)tri << 1, g, s, width, height);
                }
            } else if (n_in > 0) {  // crosses the near plane: nmi_mesh_clip_kernel's business
                const unsigned long long at = atomicAdd(&clip_state[0], 1ull);
                if (at < clip_cap) clipq[at] = ClipItem{(unsigned long long)tri, (uint32_t)s, 0u};
            }
        }
        // large triangles: one bin entry per tile their box touches.  All lanes of the wavefront walk their tiles together, four
        // tile steps to a batch: the batch's counter updates are in flight together (bin_reserve), then its entries are stored.
        const int tx0 = large ? t.x_lo / kTile : 0, tx1 = large ? t.x_hi / kTile : -1, ty0 = large ? t.y_lo / kTile : 0, ty1 = large ? t.y_hi / kTile : -1;
        int cxt = tx0, cyt = ty0;
        bool more = large;
        while (__any(more)) {
            constexpr int kBatch = 4;
            int want[kBatch], at_x[kBatch], at_y[kBatch];
            BinTicket ticket[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                want[u] = more ? s * tiles + cyt * g.tiles_x + cxt : -1;
                at_x[u] = cxt, at_y[u] = cyt;
                ticket[u] = bin_reserve(g, want[u]);
                if (more && ++cxt > tx1) {
                    cxt = tx0;
                    if (++cyt > ty1) more = false;
                }
            }
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                const int slot = bin_commit(g, want[u], (uint32_t)(tri << 1), ticket[u]);
                if (want[u] >= 0 && slot < 0)  // bin full: this lane rasterises its triangle's part of the tile itself
                    raster_direct(t, max(t.x_lo, at_x[u] * kTile), min(t.x_hi, at_x[u] * kTile + kTile - 1), max(t.y_lo, at_y[u] * kTile),
                                  min(t.y_hi, at_y[u] * kTile + kTile - 1), (unsigned long long)tri << 1, g, s, width, height);
            }
        }
}

// (A captured level passes its warp stack's work along: the first wf.blocks workgroups of the launch are warp blocks,
// nmi_warp_device.h, and the graph needs no branch for them.)
struct WarpFuse {
    const uint8_t *frame;
    const float *coeffs;
    uint8_t *warps;
    int blocks;  // 0: none
};

__global__ __launch_bounds__(256) void nmi_mesh_bin_kernel(const float *__restrict__ xyz, const float *__restrict__ uv, long long ntri,
                                                           const float *__restrict__ mvps, int views, int width, int height, BinGrid g,
                                                           ClipItem *__restrict__ clipq, unsigned long long *__restrict__ clip_state,
                                                           unsigned long long clip_cap, int shares, WarpFuse wf)
{
    // The triangle blocks come first in the launch: what they take is round trips (frustum test, appends) on the few hundred of
    // them that see anything, and those should start at once; the warp blocks are plain arithmetic and fill in behind
    // (warp blocks first: 30.4 / 55.8 us for this kernel in a level of 4,800 / 120 k triangles; this order: see profiles/NOTES.md).
    const unsigned tri_total = gridDim.x - (unsigned)wf.blocks;
    if (blockIdx.x >= tri_total) {
        warp_lds_block_linear(wf.frame, wf.coeffs, wf.warps, width, height, (int)(blockIdx.x - tri_total), (int)threadIdx.x);
        return;
    }
    const unsigned tri_blocks = tri_total / (unsigned)shares;
    const unsigned bid = blockIdx.x, share = bid / tri_blocks, tri_block = bid - share * tri_blocks;
    __shared__ float m_all[kMaxViewsPerLaunch * 16];
    __shared__ float wave_box[4][6];
    __shared__ uint32_t beyond[kMaxViewsPerLaunch];
    // share: which part of the views this block takes its 256 triangles through.  Few triangles -> many shares
    // (4,800 triangles x 27 views: a lane per pair), so that the kernel is not 19 workgroups each walking 27 views through
    // dependent atomics; many triangles -> one share, the mesh is read once.
    const int v_first = (int)(((long long)views * share) / shares), v_end = (int)(((long long)views * (share + 1)) / shares);
    for (int t = threadIdx.x + v_first * 16; t < v_end * 16; t += blockDim.x) m_all[t] = mvps[t];
    for (int t = threadIdx.x; t < views; t += blockDim.x) beyond[t] = (t >= v_first && t < v_end) ? 0x3Fu : 0u;
    const long long tri = tri_block * (long long)blockDim.x + threadIdx.x;
    const bool valid = tri < ntri;
    float px[3] = {0, 0, 0}, py[3] = {0, 0, 0}, pz[3] = {0, 0, 0}, tu[3], tv[3];
    if (valid) load_tri(xyz, uv, tri, px, py, pz, tu, tv);
    // (Fetching the block's corners through LDS with 16-byte loads instead -- a lane's nine floats lie 36 bytes from its
    // neighbour's -- measured slower: 12.8 vs 10.4 us for this kernel up to the frustum test, 120 k triangles.)
    {
        // a triangle all of whose corners are beyond one clip plane is rejected by tri_setup; the block's box decides
        // that for its 256 triangles at once
        const float inf = __builtin_huge_valf();
        const float lo[3] = {valid ? fminf(px[0], fminf(px[1], px[2])) : inf, valid ? fminf(py[0], fminf(py[1], py[2])) : inf,
                             valid ? fminf(pz[0], fminf(pz[1], pz[2])) : inf};
        const float hi[3] = {valid ? fmaxf(px[0], fmaxf(px[1], px[2])) : -inf, valid ? fmaxf(py[0], fmaxf(py[1], py[2])) : -inf,
                             valid ? fmaxf(pz[0], fmaxf(pz[1], pz[2])) : -inf};
        block_frustum_cull(m_all, v_first, v_end, lo, hi, wave_box, beyond);
    }
    if (g.dbg & 128) return;
    for (int s = v_first; s < v_end; ++s) {
        if (beyond[s]) continue;  // block-uniform
        bin_one_view(m_all + s * 16, s, tri, valid, px, py, pz, width, height, g, clipq, clip_state, clip_cap);
    }
}

// The two-kernel form of the binning pass (launch_render_mesh takes it whenever the work area holds a pair list).
// nmi_mesh_cull_kernel: one workgroup per 256 triangles finds the views that can see the block's box (block_frustum_cull over ALL
// the views at once) and appends one entry per surviving (block, view) PAIR to a list.  nmi_mesh_bin_pairs_kernel: a bounded number
// of worker workgroups share the list out; a worker fetches a pair's 256 triangles and takes them through that ONE view.
// Why: in the one-kernel form a workgroup that sees something walks its share of the views one after the other, every view a chain
// of set-up and append round trips, while most workgroups of the launch had nothing to do but cost their fixed latency; here every
// unit of work is one view deep and the launch is as wide as the chip (27 views of 120 k triangles, whole stack: 159 -> 146 us).
__global__ __launch_bounds__(256) void nmi_mesh_cull_kernel(const float *__restrict__ xyz, long long ntri, const float *__restrict__ mvps, int views,
                                                            uint32_t *__restrict__ pairs, uint32_t *__restrict__ pair_state)
{
    __shared__ float m_all[kMaxViewsPerLaunch * 16];
    __shared__ float wave_box[4][6];
    __shared__ uint32_t beyond[kMaxViewsPerLaunch];
    for (int t = threadIdx.x; t < views * 16; t += blockDim.x) m_all[t] = mvps[t];
    for (int t = threadIdx.x; t < kMaxViewsPerLaunch; t += blockDim.x) beyond[t] = t < views ? 0x3Fu : 0u;
    const long long tri = blockIdx.x * 256ll + threadIdx.x;
    const float inf = __builtin_huge_valf();
    float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
    if (tri < ntri) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float v = xyz[(tri * 3 + c) * 3 + k];
                lo[k] = fminf(lo[k], v), hi[k] = fmaxf(hi[k], v);
            }
    }
    block_frustum_cull(m_all, 0, views, lo, hi, wave_box, beyond);
    if (threadIdx.x < 64) {   // wavefront 0: lane s speaks for view s
        const int lane = (int)threadIdx.x;
        const bool alive = lane < views && beyond[lane] == 0u;
        const unsigned long long mask = __ballot(alive);
        uint32_t base = 0;
        if (lane == 0 && mask) base = atomicAdd(&pair_state[0], (uint32_t)__popcll(mask));
        base = (uint32_t)__shfl((int)base, 0, 64);
        if (alive) pairs[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = (uint32_t)blockIdx.x * (uint32_t)kMaxViewsPerLaunch + (uint32_t)lane;
    }
}

__global__ __launch_bounds__(256) void nmi_mesh_bin_pairs_kernel(const float *__restrict__ xyz, const float *__restrict__ uv, long long ntri,
                                                                 const float *__restrict__ mvps, int width, int height, BinGrid g,
                                                                 ClipItem *__restrict__ clipq, unsigned long long *__restrict__ clip_state,
                                                                 unsigned long long clip_cap, WarpFuse wf, const uint32_t *__restrict__ pairs,
                                                                 const uint32_t *__restrict__ pair_state)
{
    const unsigned workers = gridDim.x - (unsigned)wf.blocks;   // (workers first in the launch: their round trips start at once)
    if (blockIdx.x >= workers) {
        warp_lds_block_linear(wf.frame, wf.coeffs, wf.warps, width, height, (int)(blockIdx.x - workers), (int)threadIdx.x);
        return;
    }
    const uint32_t n = pair_state[0];
    if (g.dbg & 128) return;
    for (uint32_t p = blockIdx.x; p < n; p += workers) {
        const uint32_t pair = pairs[p];
        const int s = (int)(pair % (uint32_t)kMaxViewsPerLaunch);
        const long long tri = (long long)(pair / (uint32_t)kMaxViewsPerLaunch) * 256ll + threadIdx.x;
        const bool valid = tri < ntri;
        float px[3] = {0, 0, 0}, py[3] = {0, 0, 0}, pz[3] = {0, 0, 0}, tu[3], tv[3];
        if (valid) load_tri(xyz, uv, tri, px, py, pz, tu, tv);
        bin_one_view(mvps + (size_t)s * 16, s, tri, valid, px, py, pz, width, height, g, clipq, clip_state, clip_cap);   // (uniform address: the matrix in scalar registers)
    }
}

// One (triangle, view) that crosses the near plane: clip, then each of the 1 or 2 pieces goes the way of any triangle.
__device__ __forceinline__ void clip_and_bin(const float *__restrict__ xyz, const float *__restrict__ uv, long long tri, int s,
                                             const float *__restrict__ m, int width, int height, const BinGrid &g)
{
    float px[3], py[3], pz[3], tu[3], tv[3], cx[3], cy[3], cz[3], cw[3], d[3];
    load_tri(xyz, uv, tri, px, py, pz, tu, tv);
    const int n_in = tri_clip_coords(m, px, py, pz, cx, cy, cz, cw, d);
    if (n_in == 0 || n_in == 3) return;  // not this kernel's (the rescan visits every triangle)
    ClipPoly P;
    tri_clip_poly(cx, cy, cz, cw, d, tu, tv, P);
    const int tiles = g.tiles_x * g.tiles_y;
    for (int sub = 0; sub + 3 <= P.n; ++sub) {
        TriView t;
        float su[3], sv[3];
        poly_corners(P, sub, cx, cy, cz, cw, su, sv);
        if (!tri_setup(cx, cy, cz, cw, width, height, t)) continue;
        const unsigned long long id = ((unsigned long long)tri << 1) | (unsigned long long)sub;
        const int bw = t.x_hi - t.x_lo + 1, bh = t.y_hi - t.y_lo + 1;
        if (!(bw * bh > kSmallBox && g.cap > 0)) {
            raster_direct(t, t.x_lo, t.x_hi, t.y_lo, t.y_hi, id, g, s, width, height);
            continue;
        }
        for (int ty = t.y_lo / kTile; ty <= t.y_hi / kTile; ++ty)
            for (int tx = t.x_lo / kTile; tx <= t.x_hi / kTile; ++tx) {
                const size_t b = (size_t)s * tiles + ty * g.tiles_x + tx;
                const uint32_t at = atomicAdd(&g.state[2 * b], 1u);
                if (at < (uint32_t)g.cap)
                    g.bins[b * g.stride + at] = (uint32_t)id;
                else
                    raster_direct(t, max(t.x_lo, tx * kTile), min(t.x_hi, tx * kTile + kTile - 1), max(t.y_lo, ty * kTile),
                                  min(t.y_hi, ty * kTile + kTile - 1), id, g, s, width, height);
            }
    }
}

__global__ __launch_bounds__(256) void nmi_mesh_clip_kernel(const float *__restrict__ xyz, const float *__restrict__ uv, long long ntri,
                                                            const float *__restrict__ mvps, int views, int width, int height, BinGrid g,
                                                            const ClipItem *__restrict__ clipq, unsigned long long *__restrict__ clip_state,
                                                            unsigned long long clip_cap, uint32_t *__restrict__ pair_state)
{
    __shared__ unsigned long long claimed_s;
    if (pair_state && blockIdx.x == 0 && threadIdx.x == 0) pair_state[0] = 0u;   // the binning pass that read the list has ended: clean for the next render
    if (threadIdx.x == 0) claimed_s = __hip_atomic_load(&clip_state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned long long claimed = claimed_s;
    // The last block to have read the count zeroes it for the next render (no memset node in a level's graph): every block
    // draws its ticket after its read.
    if (threadIdx.x == 0 && atomicAdd(&clip_state[1], 1ull) == gridDim.x - 1) {
        __hip_atomic_store(&clip_state[0], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&clip_state[1], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const unsigned long long gid = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x, stride = (unsigned long long)gridDim.x * blockDim.x;
    if (claimed <= clip_cap) {
        for (unsigned long long i = gid; i < claimed; i += stride)
            clip_and_bin(xyz, uv, (long long)clipq[i].tri, (int)clipq[i].view, mvps + clipq[i].view * 16, width, height, g);
    } else {
        // more crossing triangles than the queue holds: look at every (triangle, view) again
        const unsigned long long all = (unsigned long long)ntri * (unsigned long long)views;
        for (unsigned long long i = gid; i < all; i += stride) {
            const long long tri = (long long)(i / (unsigned long long)views);
            const int s = (int)(i % (unsigned long long)views);
            clip_and_bin(xyz, uv, tri, s, mvps + s * 16, width, height, g);
        }
    }
}

namespace {

// LDS record of one bin entry (words).
enum : int {
    R_XW = 0, R_YW = 3, R_ZW = 6, R_INV_AREA = 9,
    R_OWN = 10,     // bits 0..2: edge k owns the pixels on it
    R_BOX = 11,     // x0 | y0 << 16: first pixel of the triangle's pixel box inside this tile
    R_BOX_W = 12,   // the box's width | its pixels << 16
    R_FIRST = 13,   // box pixels of the records before this one
    R_ID = 14,      // triangle << 1 | piece
    R_PLANES = 16,  // xr, yr, s0, r0, sx, rx, sy, ry, q0, qx, qy
    R_WORDS = 28,
};

template <int BINMAX>
struct TileLds {
    unsigned long long keys[kTile * kTile];
    uint32_t rec[BINMAX][R_WORDS];
    uint32_t wave_sum[kTileThreads / 64];
    uint32_t hdr[4];
    TexLevel tex[16];
};

// Where the key of tile pixel (x, y) lives: row y, rotated by 8 keys per row (a multiple of 4: a lane's four neighbouring
// output pixels stay neighbours).
constexpr int kVisChunk = 8;  // steps (pixel pairs) of one record a lane takes at a time in the visibility walk

__device__ __forceinline__ int key_index(int x, int y) { return y * kTile + ((x + 8 * y) & (kTile - 1)); }

__device__ __forceinline__ void planes_from_lds(const uint32_t *r, Planes &P)
{
    const float *f = reinterpret_cast<const float *>(r + R_PLANES);
    P.xr = f[0], P.yr = f[1], P.sr0 = v2f{f[2], f[3]}, P.srx = v2f{f[4], f[5]}, P.sry = v2f{f[6], f[7]}, P.q0 = f[8], P.qx = f[9], P.qy = f[10];
}

}  // namespace

template <int BINMAX>
__device__ __forceinline__ void mesh_tile_body(const float *__restrict__ xyz, const float *__restrict__ uv,
                                                                     const float *__restrict__ mvps, uint8_t *__restrict__ out, int width,
                                                                     int height, MeshTexture tex, BinGrid g)
{
    __shared__ TileLds<BINMAX> lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles = g.tiles_x * g.tiles_y;
    const int bin = blockIdx.x;
    const int s = bin / tiles, tile = bin - s * tiles, tyi = tile / g.tiles_x, txi = tile - tyi * g.tiles_x;
    const int X0 = txi * kTile, Y0 = tyi * kTile;
    // Everything the tile needs from memory is asked for at once: the bin's two state words (every lane, one address: no broadcast
    // through LDS, no barrier before the tile knows what it holds) and -- speculatively, by lane j -- bin entry j.  The two used to be
    // dependent round trips with a barrier between them; the key clear runs under their latency.
    const uint32_t c_raw = g.state[2 * (size_t)bin], f_raw = g.state[2 * (size_t)bin + 1];
    const uint32_t my_id = tid < g.cap ? g.bins[(size_t)bin * g.stride + tid] : 0u;   // (meaningful for tid < n only)
    if (tid >= 64 && tid < 64 + tex.levels) tex_level_fill(tex, tid - 64, &lds.tex[tid - 64]);
    for (int i = tid; i < kTile * kTile; i += kTileThreads) lds.keys[i] = kEmptyKey;
    const int n = (int)(c_raw < (uint32_t)g.cap ? c_raw : (uint32_t)g.cap);
    const bool from_memory = f_raw != 0u;
    // this lane's output pixels: four in a row, in rows oy and oy + 32
    const int ox = X0 + (tid & 15) * 4, oy = Y0 + (tid >> 4);
    const bool col_ok = ox < width;
    const bool dword_ok = ox + 3 < width && ((width & 3) == 0) && (((uintptr_t)out & 3) == 0);
    if ((n == 0 && !from_memory) || (g.dbg & 8)) {  // nothing was drawn into this tile: background (glClearColor(1,1,1), rendering.hpp:533)
        if (g.dbg & 8) {   // (measurement switch: the state words still have to be left clean)
            __syncthreads();
            if (tid == 0) g.state[2 * (size_t)bin] = g.state[2 * (size_t)bin + 1] = 0u;
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int y = oy + 32 * half;
            if (!col_ok || y >= height) continue;
            uint8_t *dst = out + ((size_t)s * height + y) * width + ox;
            if (dword_ok)
                *reinterpret_cast<uint32_t *>(dst) = 0xFFFFFFFFu;
            else
                for (int k = 0; k < 4 && ox + k < width; ++k) dst[k] = 255;
        }
        return;
    }
    if (n > 0) {
        // ---- set-up: lane j turns bin entry j into a record ------------------------------------------------------------
        uint32_t nst = 0;
        if (tid < n && !(g.dbg & 4)) {
            const uint32_t id = my_id;
            uint32_t *r = lds.rec[tid];
            TriView t;
            float su[3], sv[3];
            bool ok = setup_piece(xyz, uv, (long long)(id >> 1), (int)(id & 1u), mvps + s * 16, width, height, t, su, sv);
            int bx0 = 0, bx1 = 0, by0 = 0, by1 = 0;
            if (ok) {
                bx0 = max(t.x_lo, X0), bx1 = min(t.x_hi, X0 + kTile - 1), by0 = max(t.y_lo, Y0), by1 = min(t.y_hi, Y0 + kTile - 1);
                ok = bx0 <= bx1 && by0 <= by1;
            }
            if (ok) {
                Planes P;
                tri_planes(t, su, sv, P);
                float *f = reinterpret_cast<float *>(r);
#pragma unroll
                for (int k = 0; k < 3; ++k) f[R_XW + k] = t.xw[k], f[R_YW + k] = t.yw[k], f[R_ZW + k] = t.zw[k];
                f[R_INV_AREA] = t.inv_area;
                f[R_PLANES + 0] = P.xr, f[R_PLANES + 1] = P.yr, f[R_PLANES + 2] = P.sr0.x, f[R_PLANES + 3] = P.sr0.y, f[R_PLANES + 4] = P.srx.x;
                f[R_PLANES + 5] = P.srx.y, f[R_PLANES + 6] = P.sry.x, f[R_PLANES + 7] = P.sry.y, f[R_PLANES + 8] = P.q0, f[R_PLANES + 9] = P.qx;
                f[R_PLANES + 10] = P.qy;
                r[R_OWN] = (t.own[0] ? 1u : 0u) | (t.own[1] ? 2u : 0u) | (t.own[2] ? 4u : 0u);
                r[R_BOX] = (uint32_t)bx0 | ((uint32_t)by0 << 16);
                const uint32_t nsteps = (uint32_t)(((bx1 - bx0 + 2) >> 1) * (by1 - by0 + 1));  // steps of the visibility walk: pairs of pixels in a row; at most 2048
                nst = (nsteps + kVisChunk - 1) / kVisChunk;                                      // ... taken in chunks of kVisChunk
                r[R_BOX_W] = (uint32_t)(bx1 - bx0 + 1) | (nsteps << 16);
            } else {
                r[R_BOX_W] = 0u;
            }
            r[R_ID] = id;
        }
        // stamps of the records before each one (n <= 255: wavefronts 0..3 hold them)
        uint32_t incl = nst;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
            if (lane >= off) incl += up;
        }
        if (lane == 63) lds.wave_sum[wave] = incl;
        __syncthreads();
        if (tid == 0) {   // every wavefront has read the state words: both are left zero for the next render
            if (c_raw) g.state[2 * (size_t)bin] = 0u;
            if (f_raw) g.state[2 * (size_t)bin + 1] = 0u;
        }
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < (BINMAX + 63) / 64; ++w) {
            const uint32_t ws = lds.wave_sum[w];
            before += w < wave ? ws : 0u;
            total += ws;
        }
        if (tid < n) lds.rec[tid][R_FIRST] = before + incl - nst;
        __syncthreads();
        // ---- visibility -------------------------------------------------------------------------------------------------------
        // A step is two horizontally adjacent pixels of a record's box; a record's steps are cut into chunks of kVisChunk, and the
        // chunks of all records form one list that the 512 lanes deal out among themselves.  Every lane of a wavefront is then at
        // the same point of the same code at all times -- find the chunk's record, load its corners, take kVisChunk steps -- whatever
        // the triangles' sizes: no lane waits while another reloads.  (Lanes walking equal contiguous shares of the PIXEL list,
        // each reloading its record wherever its share ran into the next one, spent the walk in divergent reload passes: 45 us
        // of a 112 us kernel, and halving the arithmetic of a step changed nothing.  64-pixel stamps laid over each box, a
        // wavefront per record, had 20-40 % of their lanes inside the box for the 9 x 9-pixel boxes of a 120 k-triangle mesh.)
        // Coverage + depth of both pixels of a step go through the packed fp32 instructions; ds_min_u64 on each covered pixel's key.
        if (!(g.dbg & 6)) {
            for (uint32_t c = (uint32_t)tid; c < total; c += kTileThreads) {
                int lo = 0, hi = n - 1;  // the last record that starts at or before chunk c (records without pixels share their successor's start)
                while (lo < hi && !(g.dbg & 64)) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (lds.rec[mid][R_FIRST] <= c)
                        lo = mid;
                    else
                        hi = mid - 1;
                }
                const int i = (g.dbg & 64) ? (int)(c % (uint32_t)n) : lo;
                const uint32_t *r = lds.rec[i];
                const float *f = reinterpret_cast<const float *>(r);
                TriView t;
#pragma unroll
                for (int k = 0; k < 3; ++k) t.xw[k] = f[R_XW + k], t.yw[k] = f[R_YW + k], t.zw[k] = f[R_ZW + k];
                t.inv_area = f[R_INV_AREA];
                const uint32_t own = r[R_OWN];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int a = (k + 1) % 3, b = (k + 2) % 3;
                    t.ex[k] = t.xw[b] - t.xw[a];
                    t.ey[k] = t.yw[b] - t.yw[a];
                }
                const uint32_t bw_n = r[R_BOX_W], nsteps = bw_n >> 16, bw = bw_n & 0xFFFFu, per_row = (bw + 1u) >> 1;
                uint32_t step = (c - r[R_FIRST]) * kVisChunk;
                const uint32_t row0 = step / per_row;
                const int x_first = (int)(r[R_BOX] & 0xFFFFu), x_last = x_first + (int)bw - 1;
                int xx = x_first + 2 * (int)(step - row0 * per_row), yy = (int)(r[R_BOX] >> 16) + (int)row0;
                const unsigned long long key_lo = ((unsigned long long)r[R_ID] << 9) | (unsigned long long)(uint32_t)i;
#pragma unroll
                for (int st = 0; st < ((g.dbg & 32) ? 0 : kVisChunk); ++st, ++step) {
                    // coverage (top-left rule) and depth of the pixels (xx, yy) and (xx + 1, yy): tri_cover with per-lane ownership bits
                    const v2f fxp = {(float)xx + 0.5f, (float)xx + 1.5f};   // ((float)(xx + 1) + 0.5f: small integers, exact either way)
                    const float fyp = (float)yy + 0.5f;
                    v2f bary[3];
                    bool in0 = step < nsteps, in1 = in0 & (xx < x_last);
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const int a = (k + 1) % 3;
                        const float ea = t.ex[k] * (fyp - t.yw[a]);
                        bary[k] = (v2f{ea, ea} - (fxp - t.xw[a]) * t.ey[k]) * t.inv_area;
                        const bool owns = ((own >> k) & 1u) != 0u;
                        // (bitwise on purpose: the short-circuit forms compile to a maze of 30 branches per step)
                        in0 = in0 & ((bary[k].x > 0.0f) | ((bary[k].x == 0.0f) & owns));
                        in1 = in1 & ((bary[k].y > 0.0f) | ((bary[k].y == 0.0f) & owns));
                    }
                    const v2f z = (bary[0] * t.zw[0] + bary[1] * t.zw[1]) + bary[2] * t.zw[2];
                    if (!(g.dbg & 16)) {
                        const v2f zq = z * 16777215.0f + 0.5f;
                        if (in0 & (z.x >= 0.0f) & (z.x <= 1.0f)) {
                            const uint32_t depth = min((uint32_t)zq.x, 0xFFFFFFu);
                            (void)__hip_atomic_fetch_min(&lds.keys[key_index(xx - X0, yy - Y0)], ((unsigned long long)depth << 40) | key_lo,
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        if (in1 & (z.y >= 0.0f) & (z.y <= 1.0f)) {
                            const uint32_t depth = min((uint32_t)zq.y, 0xFFFFFFu);
                            (void)__hip_atomic_fetch_min(&lds.keys[key_index(xx + 1 - X0, yy - Y0)], ((unsigned long long)depth << 40) | key_lo,
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                    if (xx + 2 > x_last)
                        xx = x_first, ++yy;
                    else
                        xx += 2;
                }
            }
        }
        __syncthreads();
    } else {
        __syncthreads();   // (only what came through the memory buffer: the texture levels' table is in LDS, everyone has read the state words)
        if (tid == 0) g.state[2 * (size_t)bin + 1] = 0u;
    }
    if (!col_ok) return;
    TexLevel base_level;
    tex_level_fill(tex, 0, &base_level);   // from the kernel's arguments: scalar registers
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int y = oy + 32 * half;
        if (y >= height) break;
        unsigned long long keys[4] = {kEmptyKey, kEmptyKey, kEmptyKey, kEmptyKey};
        if (n > 0) {
            const ulonglong2 *k2 = reinterpret_cast<const ulonglong2 *>(&lds.keys[key_index((tid & 15) * 4, y - Y0)]);
            const ulonglong2 a = k2[0], b = k2[1];
            keys[0] = a.x, keys[1] = a.y, keys[2] = b.x, keys[3] = b.y;
        }
        // ---- what came through the memory buffer (small triangles, bin overflow): take it and leave the buffer clean -----
        if (from_memory) {
            unsigned long long *zp = g.zbuf + ((size_t)s * height + y) * width + ox;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (ox + k >= width) break;
                const unsigned long long z = zp[k];
                if (z != kEmptyKey) {
                    zp[k] = kEmptyKey;
                    keys[k] = keys[k] < z ? keys[k] : z;
                }
            }
        }
        // ---- shade every pixel once, by the triangle that won it ------------------------------------------------------------
        uint32_t packed = 0xFFFFFFFFu;
        // Pixels won through the memory buffer have no record in LDS: their triangle is set up here, on the fly.  One copy of
        // that code, outside the unrolled loop below (for a mesh of pixel-sized triangles this IS the shading loop).
        if (from_memory) {
#pragma unroll 1
            for (int k = 0; k < 4; ++k) {
                const unsigned long long key = keys[0];  // (the keys rotate through slot 0: no dynamic register indexing)
                keys[0] = keys[1], keys[1] = keys[2], keys[2] = keys[3], keys[3] = key;
                if (key == kEmptyKey || (uint32_t)(key & 0x1FFu) != kNoSlot || ox + k >= width) continue;
                TriView t;
                float su[3], sv[3];
                const unsigned long long id = (key >> 9) & 0x7FFFFFFFull;
                uint32_t grey = 255u;
                if (setup_piece(xyz, uv, (long long)(id >> 1), (int)(id & 1ull), mvps + s * 16, width, height, t, su, sv)) {  // (true: it produced this key)
                    Planes P;
                    tri_planes(t, su, sv, P);
                    grey = shade_pixel(P, tex.luma, base_level, lds.tex, tex.levels, (float)(ox + k) + 0.5f, (float)y + 0.5f);
                }
                packed = (packed & ~(0xFFu << (8 * k))) | (grey << (8 * k));
                keys[3] = kEmptyKey;  // done
            }
        }
        // the four pixels side by side: their texel fetches overlap
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned long long key = keys[k];
            if ((key != kEmptyKey) & (ox + k < width) & !(g.dbg & 1)) {
                Planes P;
                planes_from_lds(lds.rec[(uint32_t)(key & 0x1FFu)], P);
                const uint32_t grey = shade_pixel(P, tex.luma, base_level, lds.tex, tex.levels, (float)(ox + k) + 0.5f, (float)y + 0.5f);
                packed = (packed & ~(0xFFu << (8 * k))) | (grey << (8 * k));
            }
        }
        uint8_t *dst = out + ((size_t)s * height + y) * width + ox;
        if (dword_ok)
            *reinterpret_cast<uint32_t *>(dst) = packed;
        else
            for (int k = 0; k < 4 && ox + k < width; ++k) dst[k] = (uint8_t)(packed >> (8 * k));
    }
}

// Two builds of the tile kernel.  The usual one holds 255 bin entries per tile (62 KB of LDS: two workgroups per CU).  For a mesh
// whose tiles cannot fill that -- launch_render_mesh decides from the triangle count -- the small one holds 127 (48 KB) and is
// held to 80 registers, so that THREE workgroups share a CU: the kernel spends half its wave-cycles waiting (set-up, barriers,
// texel fetches), and a third workgroup to switch to is worth more than the spills the register cap costs (93.5 -> 81.5 us at
// 4,800 triangles, same box).
__global__ __launch_bounds__(kTileThreads) void nmi_mesh_tile_kernel(const float *__restrict__ xyz, const float *__restrict__ uv,
                                                                     const float *__restrict__ mvps, uint8_t *__restrict__ out, int width,
                                                                     int height, MeshTexture tex, BinGrid g)
{
    mesh_tile_body<kBinMax>(xyz, uv, mvps, out, width, height, tex, g);
}

constexpr int kBinSmall = 128;

__global__ __launch_bounds__(kTileThreads) __attribute__((amdgpu_waves_per_eu(6, 6))) void nmi_mesh_tile_small_kernel(
    const float *__restrict__ xyz, const float *__restrict__ uv, const float *__restrict__ mvps, uint8_t *__restrict__ out, int width, int height,
    MeshTexture tex, BinGrid g)
{
    mesh_tile_body<kBinSmall>(xyz, uv, mvps, out, width, height, tex, g);
}


__global__ __launch_bounds__(256) void nmi_mesh_clear_kernel(unsigned long long *zbuf, size_t n, uint32_t *state, size_t n_state,
                                                             unsigned long long *clip_state)
{
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    if (t < 2) clip_state[t] = 0ull;
    for (size_t i = t; i < n; i += step) zbuf[i] = kEmptyKey;
    for (size_t i = t; i < n_state; i += step) state[i] = 0u;
}

void mesh_geometry(int S, int width, int height, int *tiles_x, int *tiles_y, int *stride)
{
    *tiles_x = (width + kTile - 1) / kTile;
    *tiles_y = (height + kTile - 1) / kTile;
    // bins share a budget of 32 MiB: 256 entries each for 27 views of 848x480 (3 MiB), fewer for very large frames
    const long long bins = (long long)(S > 0 ? S : 1) * *tiles_x * *tiles_y;
    int st = kBinMax;
    while (st > 32 && bins * st * 4 > (32ll << 20)) st >>= 1;
    *stride = st;
}

size_t mesh_zbuf_bytes(int S, int width, int height) { return (size_t)S * width * height * sizeof(unsigned long long); }

size_t mesh_bins_bytes(int S, int width, int height)
{
    int tx, ty, st;
    mesh_geometry(S, width, height, &tx, &ty, &st);
    return (size_t)S * tx * ty * st * sizeof(uint32_t);
}

size_t mesh_state_bytes(int S, int width, int height)
{
    int tx, ty, st;
    mesh_geometry(S, width, height, &tx, &ty, &st);
    return (size_t)S * tx * ty * 2 * sizeof(uint32_t);
}

size_t mesh_clip_item_bytes() { return sizeof(ClipItem); }

hipError_t launch_mesh_clear(const MeshWork &w, int S, int width, int height, hipStream_t stream)
{
    const size_t n = (size_t)S * width * height, ns = mesh_state_bytes(S, width, height) / sizeof(uint32_t);
    hipLaunchKernelGGL(nmi_mesh_clear_kernel, dim3(4096), dim3(256), 0, stream, w.zbuf, n, w.state, ns, w.clip_state);
    return hipGetLastError();
}

size_t mesh_pairs_entries(long long ntri) { return (size_t)((ntri + 255) / 256) * kMaxViewsPerLaunch; }

hipError_t launch_render_mesh(const float *xyz, const float *uv, long long ntri, const float *luma, int levels, const int *lw,
                              const int *lh, const long long *loff, const float *mvps, int S, const MeshWork &w, int layout_views,
                              int bin_cap_limit, unsigned long long clip_cap_limit, uint8_t *out, int width, int height, hipStream_t stream,
                              const uint8_t *warp_frame, const float *warp_coeffs, uint8_t *warp_out, int Wn)
{
    if (S > layout_views) return hipErrorInvalidValue;
    WarpFuse wf{warp_frame, warp_coeffs, warp_out, 0};
    if (warp_frame) {
        if (!warp_lds_eligible(warp_frame, warp_out, width) || ntri <= 0 || S > kMaxViewsPerLaunch) return hipErrorInvalidValue;
        wf.blocks = warp_blocks_x(width) * warp_blocks_y(height) * Wn;
    }
    if (ntri >= (1ll << 30) || width > 65535 || height > 65535) return hipErrorInvalidValue;  // triangle and piece share 31 bits of a key
    if (!w.zbuf || !w.bins || !w.state || !w.clip_queue || !w.clip_state) return hipErrorInvalidValue;
    MeshTexture tex{};
    tex.luma = luma;
    tex.levels = levels;
    for (int l = 0; l < levels && l < 16; ++l)
        tex.w[l] = lw[l], tex.h[l] = lh[l], tex.off[l] = loff[l], tex.inv_w[l] = 1.0f / (float)lw[l], tex.inv_h[l] = 1.0f / (float)lh[l];
    BinGrid g{};
    static const int dbg = getenv("NMI_MESH_DBG") ? atoi(getenv("NMI_MESH_DBG")) : 0;
    g.dbg = dbg;
    mesh_geometry(layout_views, width, height, &g.tiles_x, &g.tiles_y, &g.stride);  // the layout the work area was allocated for
    g.cap = g.stride - 1 < bin_cap_limit ? g.stride - 1 : bin_cap_limit;
    if (g.cap < 0) g.cap = 0;
    const int tiles = g.tiles_x * g.tiles_y;
    // The small tile kernel (127 entries per tile, three workgroups per CU) for meshes that could not fill more even if every
    // triangle were in view and touched two tiles; a fuller bin than its capacity takes the per-lane path as always.
    static const bool no_small = getenv("NMI_MESH_NO_SMALL_TILES") != nullptr;   // measurement switch
    const bool small_tiles = !no_small && (ntri * 2 <= (long long)(kBinSmall - 1) * tiles || g.cap <= kBinSmall - 1);
    if (small_tiles && g.cap > kBinSmall - 1) g.cap = kBinSmall - 1;
    const unsigned long long clip_cap = w.clip_cap < clip_cap_limit ? w.clip_cap : clip_cap_limit;
    for (int s0 = 0; s0 < S; s0 += kMaxViewsPerLaunch) {
        const int views = S - s0 < kMaxViewsPerLaunch ? S - s0 : kMaxViewsPerLaunch;
        g.bins = w.bins + (size_t)s0 * tiles * g.stride;
        g.state = w.state + (size_t)s0 * tiles * 2;
        g.zbuf = w.zbuf + (size_t)s0 * width * height;
        if (ntri > 0) {
            ClipItem *clipq = static_cast<ClipItem *>(w.clip_queue);
            const long long nblocks = (ntri + 255) / 256;
            static const bool no_pairs = getenv("NMI_MESH_NO_PAIRS") != nullptr;   // measurement switch: the one-kernel form
            const long long cus = w.compute_units > 0 ? w.compute_units : 256;
            // (a small mesh -- fewer (block, view) pairs than four workgroups per CU -- already runs one view deep in the one-kernel
            // form, whose view shares make a workgroup per pair, and would only pay for the second launch: 119.7 vs 121.9 us at 4,800 triangles)
            if (w.pairs && w.pair_state && (unsigned long long)nblocks * kMaxViewsPerLaunch <= w.pairs_cap && nblocks < (1ll << 25) && !no_pairs &&
                nblocks * views > 4 * cus) {
                // two kernels: (block, view) pairs that can see each other, then a bounded number of workers over them (16 per CU:
                // 8 resident, the workers are bound by their round trips; 4 / 8 / 16 per CU: 149.3 / 147.2 / 146.2 us at 120 k
                // triangles, 803 / 778 / 756 us at 7.7 M)
                static const int per_cu = getenv("NMI_MESH_WORKERS") ? atoi(getenv("NMI_MESH_WORKERS")) : 16;
                long long workers = cus * (per_cu > 0 ? per_cu : 16);
                if (workers > nblocks * views) workers = nblocks * views;
                hipLaunchKernelGGL(nmi_mesh_cull_kernel, dim3((unsigned)nblocks), dim3(256), 0, stream, xyz, ntri, mvps + (size_t)s0 * 16, views, w.pairs,
                                   w.pair_state);
                hipLaunchKernelGGL(nmi_mesh_bin_pairs_kernel, dim3((unsigned)(workers + wf.blocks)), dim3(256), 0, stream, xyz, uv, ntri,
                                   mvps + (size_t)s0 * 16, width, height, g, clipq, w.clip_state, clip_cap, wf, w.pairs, w.pair_state);
            } else {
                // shares of the views: aim at ~half a million lanes
                static const long long lanes_wanted = getenv("NMI_MESH_LANES") ? atoll(getenv("NMI_MESH_LANES")) : 500000;
                long long shares = (lanes_wanted + ntri - 1) / ntri;
                shares = shares < 1 ? 1 : (shares > views ? views : shares);
                hipLaunchKernelGGL(nmi_mesh_bin_kernel, dim3((unsigned)(wf.blocks + ((ntri + 255) / 256) * shares)), dim3(256), 0, stream, xyz, uv, ntri,
                                   mvps + (size_t)s0 * 16, views, width, height, g, clipq, w.clip_state, clip_cap, (int)shares, wf);
            }
            // (crossing triangles are few: an empty pass should cost little)
            hipLaunchKernelGGL(nmi_mesh_clip_kernel, dim3(64), dim3(256), 0, stream, xyz, uv, ntri, mvps + (size_t)s0 * 16, views, width,
                               height, g, clipq, w.clip_state, clip_cap, w.pair_state);
        }
        if (small_tiles)
            hipLaunchKernelGGL(nmi_mesh_tile_small_kernel, dim3((unsigned)(views * tiles)), dim3(kTileThreads), 0, stream, xyz, uv,
                               mvps + (size_t)s0 * 16, out + (size_t)s0 * width * height, width, height, tex, g);
        else
            hipLaunchKernelGGL(nmi_mesh_tile_kernel, dim3((unsigned)(views * tiles)), dim3(kTileThreads), 0, stream, xyz, uv, mvps + (size_t)s0 * 16,
                               out + (size_t)s0 * width * height, width, height, tex, g);
    }
    return hipGetLastError();
}

}  // namespace nmi
