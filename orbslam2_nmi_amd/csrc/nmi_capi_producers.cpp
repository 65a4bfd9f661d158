// nmi_capi_producers.cpp -- C ABI of the stack producers: warp stack, point-cloud and textured-mesh
// render stacks.  Declared in include/nmi_hip.h.
#include "nmi_ctx.h"

using namespace nmi_internal;

namespace nmi_internal {

// Buffers of the mesh renderer (nmi_mesh.hip) for S views: 8 bytes per pixel of visibility keys for the direct path, the
// tile bins (at most 32 MiB) and their state words, the queue of (triangle, view) pairs crossing the near plane.
int mesh_work_alloc(nmi_ctx *ctx, int S, nmi::MeshWork *w)
{
    const int width = ctx->params.width, height = ctx->params.height;
    constexpr unsigned long long kClipItems = 1ull << 18;  // beyond it the clip pass rescans the mesh
    NMI_HIP_TRY(ctx, hipMalloc((void **)&w->zbuf, nmi::mesh_zbuf_bytes(S, width, height)));
    NMI_HIP_TRY(ctx, hipMalloc((void **)&w->bins, nmi::mesh_bins_bytes(S, width, height)));
    NMI_HIP_TRY(ctx, hipMalloc((void **)&w->state, nmi::mesh_state_bytes(S, width, height)));
    NMI_HIP_TRY(ctx, hipMalloc(&w->clip_queue, (size_t)kClipItems * nmi::mesh_clip_item_bytes()));
    w->clip_cap = kClipItems;
    NMI_HIP_TRY(ctx, hipMalloc((void **)&w->clip_state, 2 * sizeof(unsigned long long)));
    NMI_HIP_TRY(ctx, hipMalloc((void **)&w->pair_state, sizeof(uint32_t)));
    NMI_HIP_TRY(ctx, hipMemsetAsync(w->pair_state, 0, sizeof(uint32_t), ctx->stream));
    w->compute_units = ctx->compute_units;
    NMI_HIP_TRY(ctx, nmi::launch_mesh_clear(*w, S, width, height, ctx->stream));
    return NMI_OK;
}

// The pair list of the two-kernel binning pass: 64 entries per block of 256 triangles (grown on demand; a level sizes it once).
int ensure_mesh_pairs(nmi_ctx *ctx, nmi::MeshWork *w, long long n_triangles)
{
    const unsigned long long need = nmi::mesh_pairs_entries(n_triangles);
    if (need <= w->pairs_cap) return NMI_OK;
    if (w->pairs) {
        NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(w->pairs);
        w->pairs = nullptr, w->pairs_cap = 0;
    }
    if (hipMalloc((void **)&w->pairs, (size_t)need * sizeof(uint32_t)) != hipSuccess) {
        (void)hipGetLastError();
        w->pairs = nullptr;   // (no list: launch_render_mesh takes the one-kernel form)
        return NMI_OK;
    }
    w->pairs_cap = need;
    return NMI_OK;
}

void mesh_work_free(nmi::MeshWork *w)
{
    void *all[] = {w->zbuf, w->bins, w->state, w->clip_queue, w->clip_state, w->pairs, w->pair_state};
    for (void *q : all)
        if (q) (void)hipFree(q);
    *w = nmi::MeshWork{};
}

int ensure_mesh_work(nmi_ctx *ctx, int S)
{
    if (S <= ctx->mesh_views) return NMI_OK;
    if (ctx->mesh_views) NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    mesh_work_free(&ctx->mesh);
    ctx->mesh_views = 0;
    const int rc = mesh_work_alloc(ctx, S, &ctx->mesh);
    if (rc != NMI_OK) {
        mesh_work_free(&ctx->mesh);
        return rc;
    }
    ctx->mesh_views = S;
    return NMI_OK;
}

}  // namespace nmi_internal

extern "C" {

// Image::Image warp matrices, image.cpp:76-107: theta_a starts at -(n_a - 1)/2 * step_a with the integer division
// of the reference, advances by step_a; R = Rz*Ry*Rx; M = K * R * K^-1 (doubles).
int nmi_warp_homographies(const double K[9], const int32_t num[3], const float step[3], double *out)
{
    if (!K || !num || !step || !out || num[0] <= 0 || num[1] <= 0 || num[2] <= 0) return NMI_ERR_INVALID_ARGUMENT;
    const double det = K[0] * (K[4] * K[8] - K[5] * K[7]) - K[1] * (K[3] * K[8] - K[5] * K[6]) + K[2] * (K[3] * K[7] - K[4] * K[6]);
    if (det == 0.0) return NMI_ERR_INVALID_ARGUMENT;
    double Ki[9] = {(K[4] * K[8] - K[5] * K[7]) / det, (K[2] * K[7] - K[1] * K[8]) / det, (K[1] * K[5] - K[2] * K[4]) / det,
                    (K[5] * K[6] - K[3] * K[8]) / det, (K[0] * K[8] - K[2] * K[6]) / det, (K[2] * K[3] - K[0] * K[5]) / det,
                    (K[3] * K[7] - K[4] * K[6]) / det, (K[1] * K[6] - K[0] * K[7]) / det, (K[0] * K[4] - K[1] * K[3]) / det};
    auto mul3 = [](const double *a, const double *b, double *c) {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) c[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
    };
    const int nx = num[0], ny = num[1], nz = num[2];
    double tz = (double)((float)(-(nz - 1) / 2) * step[2]);
    for (int i = 0; i < nz; ++i, tz += step[2]) {
        const double Rz[9] = {cos(tz), -sin(tz), 0, sin(tz), cos(tz), 0, 0, 0, 1};
        double ty = (double)((float)(-(ny - 1) / 2) * step[1]);
        for (int j = 0; j < ny; ++j, ty += step[1]) {
            const double Ry[9] = {cos(ty), 0, sin(ty), 0, 1, 0, -sin(ty), 0, cos(ty)};
            double tx = (double)((float)(-(nx - 1) / 2) * step[0]);
            for (int k = 0; k < nx; ++k, tx += step[0]) {
                const double Rx[9] = {1, 0, 0, 0, cos(tx), -sin(tx), 0, sin(tx), cos(tx)};
                double t1[9], R[9], t2[9];
                mul3(Rz, Ry, t1);
                mul3(t1, Rx, R);
                mul3(K, R, t2);
                mul3(t2, Ki, out + ((size_t)(i * ny + j) * nx + k) * 9);
            }
        }
    }
    return NMI_OK;
}

int nmi_warp_stack(nmi_ctx *ctx, const uint8_t *d_frame, const double *h_forward, int32_t Wn, uint8_t *d_warp_stack)
{
    if (!ctx || !d_frame || !h_forward || !d_warp_stack || Wn <= 0) return NMI_ERR_INVALID_ARGUMENT;
    ctx->detail.clear();
    DeviceGuard guard(ctx->device);
    if (Wn > ctx->warp_coeffs_cap) {
        NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < nmi_ctx::kWarpRing; ++i) {
            if (ctx->d_warp_coeffs[i]) NMI_HIP_TRY(ctx, hipFree(ctx->d_warp_coeffs[i]));
            if (ctx->h_warp_coeffs[i]) NMI_HIP_TRY(ctx, hipHostFree(ctx->h_warp_coeffs[i]));
            ctx->d_warp_coeffs[i] = ctx->h_warp_coeffs[i] = nullptr;
        }
        ctx->warp_coeffs_cap = 0;
        for (int i = 0; i < nmi_ctx::kWarpRing; ++i) {
            NMI_HIP_TRY(ctx, hipMalloc((void **)&ctx->d_warp_coeffs[i], (size_t)Wn * 9 * sizeof(float)));
            NMI_HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_warp_coeffs[i], (size_t)Wn * 9 * sizeof(float), hipHostMallocDefault));
            if (!ctx->warp_ev[i]) NMI_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->warp_ev[i], hipEventDisableTiming));
        }
        ctx->warp_coeffs_cap = Wn;
    }
    const int ring = (int)(ctx->warp_uses++ % nmi_ctx::kWarpRing);
    // this ring entry was last used kWarpRing submissions ago; normally long finished
    NMI_HIP_TRY(ctx, hipEventSynchronize(ctx->warp_ev[ring]));
    float *h_coeffs = ctx->h_warp_coeffs[ring], *d_coeffs = ctx->d_warp_coeffs[ring];
    // warpPerspective inverts the forward matrix on the host in double and hands 9 floats to the device
    for (int w = 0; w < Wn; ++w) {
        const double *m = h_forward + (size_t)w * 9;
        const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
        if (det == 0.0) return NMI_ERR_INVALID_ARGUMENT;
        const double inv[9] = {(m[4] * m[8] - m[5] * m[7]) / det, (m[2] * m[7] - m[1] * m[8]) / det, (m[1] * m[5] - m[2] * m[4]) / det,
                               (m[5] * m[6] - m[3] * m[8]) / det, (m[0] * m[8] - m[2] * m[6]) / det, (m[2] * m[3] - m[0] * m[5]) / det,
                               (m[3] * m[7] - m[4] * m[6]) / det, (m[1] * m[6] - m[0] * m[7]) / det, (m[0] * m[4] - m[1] * m[3]) / det};
        for (int e = 0; e < 9; ++e) h_coeffs[w * 9 + e] = (float)inv[e];
    }
    NMI_HIP_TRY(ctx, hipMemcpyAsync(d_coeffs, h_coeffs, (size_t)Wn * 9 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    NMI_HIP_TRY(ctx, nmi::launch_warp(d_frame, d_coeffs, d_warp_stack, ctx->params.width, ctx->params.height, Wn, ctx->stream));
    NMI_HIP_TRY(ctx, hipEventRecord(ctx->warp_ev[ring], ctx->stream));
    return NMI_OK;
}

// Projection (rendering.hpp:196-202, glm columns) * glm::lookAt(pos + t, look_at + t, up) (rendering.hpp:547-553), fp32.
int nmi_render_mvp(const nmi_render_params *rp, const float cam_pos[3], const float cam_look_at[3], const float cam_up[3],
                   const float translation[3], float out[16])
{
    if (!rp || !cam_pos || !cam_look_at || !cam_up || !translation || !out) return NMI_ERR_INVALID_ARGUMENT;
    const float eye[3] = {cam_pos[0] + translation[0], cam_pos[1] + translation[1], cam_pos[2] + translation[2]};
    const float ctr[3] = {cam_look_at[0] + translation[0], cam_look_at[1] + translation[1], cam_look_at[2] + translation[2]};
    float f[3] = {ctr[0] - eye[0], ctr[1] - eye[1], ctr[2] - eye[2]};
    float len = sqrtf(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    if (!(len > 0.0f)) return NMI_ERR_INVALID_ARGUMENT;
    for (float &v : f) v /= len;
    float sv[3] = {f[1] * cam_up[2] - f[2] * cam_up[1], f[2] * cam_up[0] - f[0] * cam_up[2], f[0] * cam_up[1] - f[1] * cam_up[0]};
    len = sqrtf(sv[0] * sv[0] + sv[1] * sv[1] + sv[2] * sv[2]);
    if (!(len > 0.0f)) return NMI_ERR_INVALID_ARGUMENT;
    for (float &v : sv) v /= len;
    const float u[3] = {sv[1] * f[2] - sv[2] * f[1], sv[2] * f[0] - sv[0] * f[2], sv[0] * f[1] - sv[1] * f[0]};
    // view matrix, column-major V[c*4 + r]
    float V[16] = {sv[0], u[0], -f[0], 0, sv[1], u[1], -f[1], 0, sv[2], u[2], -f[2], 0, 0, 0, 0, 1};
    V[12] = -(sv[0] * eye[0] + sv[1] * eye[1] + sv[2] * eye[2]);
    V[13] = -(u[0] * eye[0] + u[1] * eye[1] + u[2] * eye[2]);
    V[14] = f[0] * eye[0] + f[1] * eye[1] + f[2] * eye[2];
    const double zn = rp->near_plane, zf = rp->far_plane;
    float P[16] = {0};
    P[0] = (float)(rp->fx / (-rp->cx));          // Projection[0] = (fx / -cx, 0, 0, 0)
    P[5] = (float)(rp->fy / (-rp->cy));          // Projection[1] = (0, fy / -cy, 0, 0)
    P[10] = (float)((zn + zf) / (zn - zf));      // Projection[2] = (0, 0, (zn+zf)/(zn-zf), -1)
    P[11] = -1.0f;
    P[14] = (float)(2 * zn * zf / (zn - zf));    // Projection[3] = (0, 0, 2 zn zf/(zn-zf), 0)
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            float acc = 0.0f;
            for (int k = 0; k < 4; ++k) acc += P[k * 4 + r] * V[c * 4 + k];
            out[c * 4 + r] = acc;
        }
    return NMI_OK;
}

int nmi_render_points(nmi_ctx *ctx, const float *d_xyz, const float *d_red, int64_t n_points, const float *h_mvps, int32_t S,
                      float point_size, uint8_t *d_render_stack)
{
    if (!ctx || !h_mvps || !d_render_stack || S <= 0 || n_points < 0 || (n_points > 0 && (!d_xyz || !d_red)))
        return NMI_ERR_INVALID_ARGUMENT;
    ctx->detail.clear();
    DeviceGuard guard(ctx->device);
    // glPointSize: non-antialiased points use the size rounded to the nearest integer, at least 1 (OpenGL 3.3, 3.4.1)
    int size = (int)floorf(point_size + 0.5f);
    if (size < 1) size = 1;
    if (size > 64) size = 64;
    const int64_t need = (int64_t)nmi::render_zbuf_words(S, ctx->params.width, ctx->params.height, size);
    if (need > ctx->zbuf_cap) {
        NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_zbuf) NMI_HIP_TRY(ctx, hipFree(ctx->d_zbuf));
        ctx->d_zbuf = nullptr;
        ctx->zbuf_cap = 0;
        NMI_HIP_TRY(ctx, hipMalloc((void **)&ctx->d_zbuf, (size_t)need * sizeof(uint32_t)));
        ctx->zbuf_cap = need;
    }
    float *d_mvps = nullptr;
    const int src = stage_floats(ctx, ctx->mvp_ring, h_mvps, (size_t)S * 16, &d_mvps);
    if (src != NMI_OK) return src;
    NMI_HIP_TRY(ctx, nmi::launch_render_points(d_xyz, d_red, n_points, d_mvps, S, ctx->d_zbuf, d_render_stack, ctx->params.width,
                                               ctx->params.height, size, ctx->stream));
    return NMI_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Textured-mesh renderer: texture object (mip chain -> per-level luma on the device) and the draw call.
// ---------------------------------------------------------------------------------------------------------
}  // extern "C"

extern "C" {

int nmi_texture_destroy(nmi_texture *tex)
{
    if (!tex) return NMI_OK;
    DeviceGuard guard(tex->ctx->device);
    (void)hipStreamSynchronize(tex->ctx->stream);
    if (tex->d_luma) (void)hipFree(tex->d_luma);
    delete tex;
    return NMI_OK;
}

int nmi_texture_create(nmi_ctx *ctx, const uint8_t *h_rgb, int32_t tw, int32_t th, nmi_texture **out)
{
    if (!ctx || !h_rgb || !out || tw <= 0 || th <= 0 || tw > 32768 || th > 32768) return NMI_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    ctx->detail.clear();
    nmi_texture *tex = new (std::nothrow) nmi_texture;
    if (!tex) return NMI_ERR_INVALID_ARGUMENT;
    tex->ctx = ctx;
    // level sizes: max(1, floor(size / 2)) until 1x1 (OpenGL 3.3, 3.8.14)
    long long total = 0;
    int lw = tw, lh = th;
    for (;;) {
        tex->w[tex->levels] = lw;
        tex->h[tex->levels] = lh;
        tex->off[tex->levels] = total;
        total += (long long)lw * lh;
        ++tex->levels;
        if ((lw == 1 && lh == 1) || tex->levels == 16) break;
        lw = lw > 1 ? lw / 2 : 1;
        lh = lh > 1 ? lh / 2 : 1;
    }
    if (total >= (1ll << 30)) {  // the sampler addresses the pyramid with 32-bit byte offsets (4 bytes per texel)
        delete tex;
        return NMI_ERR_UNSUPPORTED;
    }
    std::vector<uint8_t> cur(h_rgb, h_rgb + (size_t)tw * th * 3), next;
    std::vector<float> luma((size_t)total);
    for (int l = 0; l < tex->levels; ++l) {
        const int w = tex->w[l], h = tex->h[l];
        float *dst = luma.data() + tex->off[l];
        for (size_t i = 0; i < (size_t)w * h; ++i)  // fragment shader :16, on normalised 8-bit channels
            dst[i] = 0.299f * ((float)cur[i * 3] / 255.0f) + 0.587f * ((float)cur[i * 3 + 1] / 255.0f) + 0.114f * ((float)cur[i * 3 + 2] / 255.0f);
        if (l + 1 == tex->levels) break;
        const int nw = tex->w[l + 1], nh = tex->h[l + 1];
        next.assign((size_t)nw * nh * 3, 0);
        for (int y = 0; y < nh; ++y)
            for (int x = 0; x < nw; ++x)
                for (int c = 0; c < 3; ++c) {  // 2x2 box filter, rounded to 8 bits per level
                    const int x0 = 2 * x < w ? 2 * x : w - 1, x1 = 2 * x + 1 < w ? 2 * x + 1 : w - 1;
                    const int y0 = 2 * y < h ? 2 * y : h - 1, y1 = 2 * y + 1 < h ? 2 * y + 1 : h - 1;
                    const int sum = cur[((size_t)y0 * w + x0) * 3 + c] + cur[((size_t)y0 * w + x1) * 3 + c] +
                                    cur[((size_t)y1 * w + x0) * 3 + c] + cur[((size_t)y1 * w + x1) * 3 + c];
                    next[((size_t)y * nw + x) * 3 + c] = (uint8_t)((sum + 2) / 4);
                }
        cur.swap(next);
    }
    DeviceGuard guard(ctx->device);
    hipError_t e = hipMalloc((void **)&tex->d_luma, (size_t)total * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(tex->d_luma, luma.data(), (size_t)total * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        const int rc = hip_fail(ctx, e, "nmi_texture_create");
        nmi_texture_destroy(tex);
        return rc;
    }
    *out = tex;
    return NMI_OK;
}

int nmi_render_mesh(nmi_ctx *ctx, const float *d_xyz, const float *d_uv, int64_t n_triangles, const nmi_texture *tex,
                    const float *h_mvps, int32_t S, uint8_t *d_render_stack)
{
    if (!ctx || !tex || tex->ctx != ctx || !h_mvps || !d_render_stack || S <= 0 || n_triangles < 0 ||
        (n_triangles > 0 && (!d_xyz || !d_uv)))
        return NMI_ERR_INVALID_ARGUMENT;
    ctx->detail.clear();
    DeviceGuard guard(ctx->device);
    // (the bins' geometry depends on the number of views: a context renders with the work area of its largest stack so far,
    // laid out for that many views; the first S of them are used)
    int rq = ensure_mesh_work(ctx, S);
    if (rq != NMI_OK) return rq;
    if ((rq = ensure_mesh_pairs(ctx, &ctx->mesh, n_triangles)) != NMI_OK) return rq;
    float *d_mvps = nullptr;
    int rc = stage_floats(ctx, ctx->mvp_ring, h_mvps, (size_t)S * 16, &d_mvps);
    if (rc != NMI_OK) return rc;
    const hipError_t e = nmi::launch_render_mesh(d_xyz, d_uv, n_triangles, tex->d_luma, tex->levels, tex->w, tex->h, tex->off, d_mvps, S, ctx->mesh,
                                                 ctx->mesh_views, (int)(ctx->tile_queue_limit < 511 ? ctx->tile_queue_limit : 511), ctx->clip_queue_limit,
                                                 d_render_stack, ctx->params.width, ctx->params.height, ctx->stream);
    if (e != hipSuccess) {
        ctx->mesh_views = 0;  // whatever state the buffers are in: allocate and clear afresh next time
        return hip_fail(ctx, e, "launch_render_mesh");
    }
    return NMI_OK;
}

int nmi_sort_points(nmi_ctx *ctx, const float *d_xyz, const float *d_red, int64_t n_points, float *d_xyz_out, float *d_red_out)
{
    if (!ctx || n_points < 0 || n_points >= (1ll << 32)) return NMI_ERR_INVALID_ARGUMENT;
    if (n_points > 0 && (!d_xyz || !d_red || !d_xyz_out || !d_red_out || d_xyz == d_xyz_out || d_red == d_red_out)) return NMI_ERR_INVALID_ARGUMENT;
    ctx->detail.clear();
    DeviceGuard guard(ctx->device);
    NMI_HIP_TRY(ctx, nmi::sort_records_morton(d_xyz, 3, 1, d_red, 1, n_points, d_xyz_out, d_red_out, ctx->stream));
    return NMI_OK;
}

int nmi_sort_triangles(nmi_ctx *ctx, const float *d_xyz, const float *d_uv, int64_t n_triangles, float *d_xyz_out, float *d_uv_out)
{
    if (!ctx || n_triangles < 0 || n_triangles >= (1ll << 32)) return NMI_ERR_INVALID_ARGUMENT;
    if (n_triangles > 0 && (!d_xyz || !d_uv || !d_xyz_out || !d_uv_out || d_xyz == d_xyz_out || d_uv == d_uv_out)) return NMI_ERR_INVALID_ARGUMENT;
    ctx->detail.clear();
    DeviceGuard guard(ctx->device);
    NMI_HIP_TRY(ctx, nmi::sort_records_morton(d_xyz, 9, 3, d_uv, 6, n_triangles, d_xyz_out, d_uv_out, ctx->stream));
    return NMI_OK;
}

}  // extern "C"
