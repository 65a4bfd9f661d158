// nmi_capi_pipeline.cpp -- C ABI of the composed forms: one search level as a captured HIP graph (nmi_level_*) and the
// double-buffered streaming pipeline (nmi_stream_*).  Declared in include/nmi_hip.h.
#include "nmi_ctx.h"

using namespace nmi_internal;

extern "C" {

// ---------------------------------------------------------------------------------------------------------
// One search level as a captured HIP graph: cloud -> S renders, frame -> Wn warps, grid search, winner to the host.
// One chain of kernel nodes and no copy nodes replays with one hipGraphLaunch (point cloud):
//   prep    reads the pinned parameter buffers, resets the key, bumps the replay parity; its other workgroups test the cloud's
//           64-point boxes -- a lane per box -- against the six planes around all views and list the survivors
//   front   a bounded number of splat workgroups over that list + the warp stack's workgroups + the clear of the NEXT replay's
//           anchor buffer
//   resolve sprites from anchors
//   search  whose last workgroup stores the winner into pinned host memory.
// (Textured mesh: prep -> binning + warp workgroups -> clip -> tiles -> search.  Where the warp cannot ride along -- frame rows
// not 16-byte aligned -- it runs on a forked branch.)  Only the pinned parameter buffers change between replays; the caller
// polls the winner word.
// ---------------------------------------------------------------------------------------------------------
}  // extern "C"

struct nmi_level {
    nmi_ctx *ctx = nullptr;
    int S = 0, Wn = 0, size = 1;                // this rank's block: S views x Wn warps ...
    int s_offset = 0, S_total = 0, w_offset = 0, Wn_total = 0;  // ... of an S_total x Wn_total level (block == level on one rank)
    uint8_t *d_renders = nullptr, *d_warps = nullptr;
    uint32_t *d_zbuf = nullptr;                 // point cloud: anchor buffer
    void *d_packed = nullptr;                   // point cloud: the level's own packed copy of the cloud (16-byte records + wavefront boxes)
    nmi::MeshWork mesh;                         // textured mesh: the renderer's work area (kept clean by the renderer itself)
    bool is_mesh = false;
    bool fused_points = false;                  // point cloud, one chain of kernels, double-buffered anchors
    uint32_t *d_kept = nullptr, *d_kept_count = nullptr;  // ... and the wavefronts in reach of a view, listed by the prep kernel per replay
    uint32_t replay = 0;                        // parity of the counters the prep kernel counts under
    float *d_mvps = nullptr, *h_mvps = nullptr, *d_coeffs = nullptr, *h_coeffs = nullptr;
    int *d_order = nullptr;
    float *d_ratings = nullptr;                 // [Wn][S] rating table of the latest replay
    unsigned long long *d_key = nullptr, *h_key = nullptr;
    unsigned int *d_done = nullptr;
    uint32_t *d_epoch = nullptr;                // replays so far, bumped by the prep kernel: parity of the double-buffered anchors (fused
                                                // point-cloud form), tag of the search kernel's hand-offs (nmi_pix_kernel)
    unsigned long long *d_pix_blocks = nullptr; // nmi_pix_kernel's hand-off blocks when the level's grid is a mid-size one
    int pix = 0;                                // pixel ranges per candidate of the captured search (0: nmi_grid_kernel)
    hipStream_t side = nullptr;                 // forked capture branch (warp)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

extern "C" {

int nmi_level_destroy(nmi_level *lv)
{
    if (!lv) return NMI_OK;
    DeviceGuard guard(lv->ctx->device);
    (void)hipStreamSynchronize(lv->ctx->stream);
    if (lv->exec) (void)hipGraphExecDestroy(lv->exec);
    if (lv->graph) (void)hipGraphDestroy(lv->graph);
    void *dev[] = {lv->d_kept, lv->d_kept_count, lv->d_packed, lv->d_renders, lv->d_warps, lv->d_zbuf, lv->d_mvps, lv->d_coeffs, lv->d_order, lv->d_key, lv->d_done, lv->d_ratings, lv->d_epoch, lv->d_pix_blocks};
    for (void *q : dev)
        if (q) (void)hipFree(q);
    void *host[] = {lv->h_mvps, lv->h_coeffs, lv->h_key};
    for (void *q : host)
        if (q) (void)hipHostFree(q);
    if (lv->ev_fork) (void)hipEventDestroy(lv->ev_fork);
    if (lv->ev_join) (void)hipEventDestroy(lv->ev_join);
    if (lv->side) (void)hipStreamDestroy(lv->side);
    mesh_work_free(&lv->mesh);
    delete lv;
    return NMI_OK;
}

}  // extern "C"

// Common part of nmi_level_create (tex == nullptr: coloured points, d_attr = red) and nmi_level_create_mesh (tex: textured
// triangles, d_attr = uv, n = triangles).
struct LevelBlock {
    int32_t S, s_offset, S_total, Wn, w_offset, Wn_total;
};

static int level_create(nmi_ctx *ctx, const float *d_xyz, const float *d_attr, int64_t n_points, const nmi_texture *tex,
                        const uint8_t *d_frame, const LevelBlock &blk, float point_size, nmi_level **out)
{
    const float *d_red = d_attr;
    const int32_t S = blk.S, Wn = blk.Wn;
    if (!ctx || !out || !d_frame || S < 0 || Wn < 0 || n_points < 0 || (n_points > 0 && (!d_xyz || !d_attr)))
        return NMI_ERR_INVALID_ARGUMENT;
    if (blk.S_total <= 0 || blk.Wn_total <= 0 || blk.s_offset < 0 || blk.w_offset < 0 || blk.s_offset + S > blk.S_total ||
        blk.w_offset + Wn > blk.Wn_total)
        return NMI_ERR_INVALID_ARGUMENT;
    if ((int64_t)blk.S_total * blk.Wn_total >= 0x7FFFFFFFll) return NMI_ERR_UNSUPPORTED;  // index lives in 32 bits of the key
    if (tex && tex->ctx != ctx) return NMI_ERR_INVALID_ARGUMENT;
    if (!ctx->params.use_bg) return NMI_ERR_UNSUPPORTED;
    *out = nullptr;
    ctx->detail.clear();
    DeviceGuard guard(ctx->device);
    nmi_level *lv = new (std::nothrow) nmi_level;
    if (!lv) return NMI_ERR_INVALID_ARGUMENT;
    lv->ctx = ctx;
    lv->S = S;
    lv->Wn = Wn;
    lv->s_offset = blk.s_offset;
    lv->S_total = blk.S_total;
    lv->w_offset = blk.w_offset;
    lv->Wn_total = blk.Wn_total;
    if (S == 0 || Wn == 0) {
        // An empty block (more ranks than cells on the sharded axis): nothing to render, warp or score.  The rank still owns
        // a key word -- "no candidate" -- for the level's collective (nmi_level_run_rccl).
        hipError_t e0 = hipMalloc((void **)&lv->d_key, sizeof(unsigned long long));
        if (e0 == hipSuccess) e0 = hipMemset(lv->d_key, 0, sizeof(unsigned long long));
        if (e0 != hipSuccess) {
            const int rc = hip_fail(ctx, e0, "nmi_level_create (empty block)");
            nmi_level_destroy(lv);
            return rc;
        }
        *out = lv;
        return NMI_OK;
    }
    int size = (int)floorf(point_size + 0.5f);
    lv->size = size < 1 ? 1 : (size > 64 ? 64 : size);
    const nmi_params &p = ctx->params;
    const size_t npix = (size_t)ctx->npix;
    const int64_t total = (int64_t)S * Wn;
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t r) {
        if (e == hipSuccess) e = r;
        return r == hipSuccess;
    };
    ok(hipMalloc((void **)&lv->d_renders, npix * S));
    ok(hipMalloc((void **)&lv->d_warps, npix * Wn));
    lv->is_mesh = tex != nullptr;
    if (tex) {
        if (mesh_work_alloc(ctx, S, &lv->mesh) != NMI_OK) e = hipErrorOutOfMemory;
        if (e == hipSuccess && ensure_mesh_pairs(ctx, &lv->mesh, n_points) != NMI_OK) e = hipErrorOutOfMemory;
        ok(hipMalloc((void **)&lv->d_epoch, sizeof(uint32_t)));
        if (e == hipSuccess) ok(hipMemsetAsync(lv->d_epoch, 0, sizeof(uint32_t), ctx->stream));
    } else {
        // Anchors: two buffers when the level runs as one chain of kernels (the front kernel of replay k clears the buffer of
        // replay k + 1); the classic form keeps one and clears it in its prep node.
        lv->fused_points = nmi::level_front_eligible(d_frame, nullptr, p.width, S) && n_points > 0 && nmi::level_points_double_buffered(p.width, lv->size);
        const size_t words = lv->fused_points ? 2 * nmi::level_zbuf_pair_words(S, p.width, p.height, lv->size)
                                              : nmi::render_zbuf_words(S, p.width, p.height, lv->size);
        ok(hipMalloc((void **)&lv->d_zbuf, words * sizeof(uint32_t)));
        ok(hipMalloc((void **)&lv->d_epoch, sizeof(uint32_t)));
        ok(hipMalloc(&lv->d_packed, nmi::cloud_pack_bytes(n_points, nullptr) + 16));
        if (e == hipSuccess) ok(hipMemsetAsync(lv->d_zbuf, 0xFF, words * sizeof(uint32_t), ctx->stream));
        if (e == hipSuccess) ok(hipMemsetAsync(lv->d_epoch, 0, sizeof(uint32_t), ctx->stream));
        if (e == hipSuccess) ok(nmi::launch_cloud_pack(d_xyz, d_red, n_points, lv->d_packed, ctx->stream));
        if (lv->fused_points) {
            ok(hipMalloc((void **)&lv->d_kept, (size_t)(2 * ((n_points + 63) / 64) + 1) * sizeof(uint32_t)));   // (twice the most one replay lists)
            ok(hipMalloc((void **)&lv->d_kept_count, 2 * sizeof(uint32_t)));
            if (e == hipSuccess) ok(hipMemsetAsync(lv->d_kept_count, 0, 2 * sizeof(uint32_t), ctx->stream));
        }
    }
    ok(hipMalloc((void **)&lv->d_mvps, ((size_t)S * 16 + nmi::kLevelMvpExtra) * sizeof(float)));
    ok(hipMalloc((void **)&lv->d_coeffs, (size_t)Wn * 9 * sizeof(float)));
    ok(hipMalloc((void **)&lv->d_order, (size_t)total * sizeof(int)));
    ok(hipMalloc((void **)&lv->d_ratings, (size_t)total * sizeof(float)));
    ok(hipMalloc((void **)&lv->d_key, sizeof(unsigned long long)));
    ok(hipMalloc((void **)&lv->d_done, sizeof(unsigned int)));
    // pinned, device-mapped, fine-grained: the prep kernel reads the parameters and the search kernel posts the winner
    ok(hipHostMalloc((void **)&lv->h_mvps, ((size_t)S * 16 + nmi::kLevelMvpExtra) * sizeof(float), hipHostMallocMapped | hipHostMallocCoherent));
    ok(hipHostMalloc((void **)&lv->h_coeffs, (size_t)Wn * 9 * sizeof(float), hipHostMallocMapped | hipHostMallocCoherent));
    ok(hipHostMalloc((void **)&lv->h_key, sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent));
    ok(hipStreamCreateWithFlags(&lv->side, hipStreamNonBlocking));
    ok(hipEventCreateWithFlags(&lv->ev_fork, hipEventDisableTiming));
    ok(hipEventCreateWithFlags(&lv->ev_join, hipEventDisableTiming));
    int *order = e == hipSuccess ? new (std::nothrow) int[(size_t)total] : nullptr;
    if (e != hipSuccess || !order) {
        const int rc = e != hipSuccess ? hip_fail(ctx, e, "nmi_level_create") : NMI_ERR_INVALID_ARGUMENT;
        nmi_level_destroy(lv);
        return rc;
    }
    build_order(S, Wn, order);
    ok(hipMemcpy(lv->d_order, order, (size_t)total * sizeof(int), hipMemcpyHostToDevice));
    delete[] order;
    ok(hipMemset(lv->d_done, 0, sizeof(unsigned int)));
    memset(lv->h_mvps, 0, ((size_t)S * 16 + nmi::kLevelMvpExtra) * sizeof(float));
    memset(lv->h_coeffs, 0, (size_t)Wn * 9 * sizeof(float));
    ok(hipDeviceSynchronize());

    nmi::GridArgs a{};
    a.render_stack = lv->d_renders;
    a.warp_stack = lv->d_warps;
    a.S_local = S;
    a.Wn = Wn;
    a.s_offset = blk.s_offset;  // global indices in the key (commit_score): the winner of a block is a cell of the whole level
    a.S_total = blk.S_total;
    a.w_offset = blk.w_offset;
    nmi::set_geometry(a, p.width, p.height, lv->d_renders, lv->d_warps, p.render_bottom_up != 0);
    a.shift = ctx->shift;
    a.mode = p.mode;
    a.table = ctx->table;
    a.order = lv->d_order;
    a.ratings = lv->d_ratings;  // 4 bytes per candidate: kept so that a level can be checked against an oracle (nmi_level_copy_outputs)
    a.key = lv->d_key;        // reset by the prep node before every replay (the ping-pong of plain launches needs
    a.reset_key = nullptr;    // alternating arguments, which a replayed graph does not have)
    a.done = lv->d_done;
    float *hd_mvps = nullptr, *hd_coeffs = nullptr;
    unsigned long long *hd_key = nullptr;
    if (e == hipSuccess) {
        ok(hipHostGetDevicePointer((void **)&hd_mvps, lv->h_mvps, 0));
        ok(hipHostGetDevicePointer((void **)&hd_coeffs, lv->h_coeffs, 0));
        ok(hipHostGetDevicePointer((void **)&hd_key, lv->h_key, 0));
    }
    a.out_key = hd_key;
    a.hist_variant = 3;
    a.phase_mask = 3;
    const int cap = ctx->workgroups > 0 ? ctx->workgroups : ctx->compute_units;
    const int workgroups = (int)(total < cap ? total : cap);
    // Mid-size grids (the live strategy's collapsed levels, a rank's block of a sharded level): P workgroups per candidate
    // (nmi_pix_kernel.hip).  Its hand-off tag = the epoch frozen into the graph + the replay count the prep kernel keeps.
    lv->pix = e == hipSuccess ? choose_pix(ctx, a, total, cap) : 0;
    if (lv->pix) {
        const size_t bytes = nmi::pix_block_bytes((int)total, lv->pix);
        ok(hipMalloc((void **)&lv->d_pix_blocks, bytes));
        if (e == hipSuccess) ok(hipMemset(lv->d_pix_blocks, 0, bytes));
        if (e == hipSuccess && (next_split_epoch(ctx, &a.epoch) != NMI_OK || ensure_pix_timeouts(ctx) != NMI_OK)) e = hipErrorOutOfMemory;
        if (e == hipSuccess) ok(hipStreamSynchronize(ctx->stream));
        a.blocks = lv->d_pix_blocks;
        a.order = nullptr;
    }

    hipStream_t st = ctx->stream;
    if (e == hipSuccess && ok(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal))) {
        // (mesh: nothing to clear -- the renderer leaves its work area clean)
        ok(nmi::launch_level_prep(hd_mvps, lv->d_mvps, S * 16 + nmi::kLevelMvpExtra, hd_coeffs, lv->d_coeffs, Wn * 9, lv->d_key, lv->d_zbuf,
                                  (tex || lv->fused_points) ? 0 : nmi::render_zbuf_words(S, p.width, p.height, lv->size), st,
                                  lv->d_epoch, lv->fused_points ? lv->d_packed : nullptr, n_points,
                                  lv->fused_points ? hd_mvps + (size_t)S * 16 : nullptr, lv->d_kept, lv->d_kept_count));
        // One chain of kernels when the warp blocks can ride along with the render's first kernel (the usual case: frame rows
        // 16-byte aligned); otherwise the warp kernel runs on a forked branch beside the render.
        const bool fused = tex ? (nmi::level_front_eligible(d_frame, lv->d_warps, p.width, S) && n_points > 0) : lv->fused_points;
        if (!fused) {
            ok(hipEventRecord(lv->ev_fork, st));
            ok(hipStreamWaitEvent(lv->side, lv->ev_fork, 0));
            ok(nmi::launch_warp(d_frame, lv->d_coeffs, lv->d_warps, p.width, p.height, Wn, lv->side));
            ok(hipEventRecord(lv->ev_join, lv->side));
        }
        if (tex)
            ok(nmi::launch_render_mesh(d_xyz, d_attr, n_points, tex->d_luma, tex->levels, tex->w, tex->h, tex->off, lv->d_mvps, S, lv->mesh, S,
                                       (int)(ctx->tile_queue_limit < 511 ? ctx->tile_queue_limit : 511), ctx->clip_queue_limit, lv->d_renders,
                                       p.width, p.height, st, fused ? d_frame : nullptr, lv->d_coeffs, lv->d_warps, Wn));
        else if (fused)
            ok(nmi::launch_level_front_points(lv->d_packed, n_points, lv->d_mvps, S, lv->d_zbuf, lv->d_epoch, lv->d_renders, p.width, p.height,
                                              lv->size, d_frame, lv->d_coeffs, lv->d_warps, Wn, st, lv->d_kept, lv->d_kept_count, ctx->compute_units));
        else
            ok(nmi::launch_render_points(d_xyz, d_red, n_points, lv->d_mvps, S, lv->d_zbuf, lv->d_renders, p.width, p.height, lv->size, st,
                                         /*clear_first=*/false));
        if (!fused) ok(hipStreamWaitEvent(st, lv->ev_join, 0));
        if (lv->pix)
            ok(nmi::launch_pix(a, lv->pix, pix_owner_share(ctx, lv->pix), true, lv->d_epoch, ctx->d_pix_timeouts, st));
        else
            ok(nmi::launch_grid(a, workgroups, true, st));
        hipError_t ec = hipStreamEndCapture(st, &lv->graph);
        ok(ec);
    }
    if (e == hipSuccess) ok(hipGraphInstantiate(&lv->exec, lv->graph, nullptr, nullptr, 0));
    if (e != hipSuccess) {
        const int rc = hip_fail(ctx, e, "nmi_level_create (graph capture)");
        nmi_level_destroy(lv);
        return rc;
    }
    *out = lv;
    return NMI_OK;
}

extern "C" {

int nmi_level_create(nmi_ctx *ctx, const float *d_xyz, const float *d_red, int64_t n_points, const uint8_t *d_frame, int32_t S,
                     int32_t Wn, float point_size, nmi_level **out)
{
    if (S <= 0 || Wn <= 0) return NMI_ERR_INVALID_ARGUMENT;
    return level_create(ctx, d_xyz, d_red, n_points, nullptr, d_frame, LevelBlock{S, 0, S, Wn, 0, Wn}, point_size, out);
}

int nmi_level_create_mesh(nmi_ctx *ctx, const float *d_xyz, const float *d_uv, int64_t n_triangles, const nmi_texture *tex,
                          const uint8_t *d_frame, int32_t S, int32_t Wn, nmi_level **out)
{
    if (!tex || S <= 0 || Wn <= 0) return NMI_ERR_INVALID_ARGUMENT;
    return level_create(ctx, d_xyz, d_uv, n_triangles, tex, d_frame, LevelBlock{S, 0, S, Wn, 0, Wn}, 1.0f, out);
}

int nmi_level_create_block(nmi_ctx *ctx, const float *d_xyz, const float *d_red, int64_t n_points, const uint8_t *d_frame, int32_t S_local,
                           int32_t s_offset, int32_t S_total, int32_t Wn_local, int32_t w_offset, int32_t Wn_total, float point_size,
                           nmi_level **out)
{
    return level_create(ctx, d_xyz, d_red, n_points, nullptr, d_frame, LevelBlock{S_local, s_offset, S_total, Wn_local, w_offset, Wn_total},
                        point_size, out);
}

int nmi_level_create_mesh_block(nmi_ctx *ctx, const float *d_xyz, const float *d_uv, int64_t n_triangles, const nmi_texture *tex,
                                const uint8_t *d_frame, int32_t S_local, int32_t s_offset, int32_t S_total, int32_t Wn_local,
                                int32_t w_offset, int32_t Wn_total, nmi_level **out)
{
    if (!tex) return NMI_ERR_INVALID_ARGUMENT;
    return level_create(ctx, d_xyz, d_uv, n_triangles, tex, d_frame, LevelBlock{S_local, s_offset, S_total, Wn_local, w_offset, Wn_total},
                        1.0f, out);
}

}  // extern "C"

// Parameters of this replay into the pinned buffers the graph's first node reads; then the launch.  No waiting.
static int level_launch(nmi_level *lv, const float *h_mvps, const double *h_forward)
{
    nmi_ctx *ctx = lv->ctx;
    memcpy(lv->h_mvps, h_mvps, (size_t)lv->S * 16 * sizeof(float));
    nmi::level_views_bound(h_mvps, lv->S, lv->h_mvps + (size_t)lv->S * 16);  // for the front kernel's first test: all views at once
    static const bool no_bound = getenv("NMI_LEVEL_NO_BOUND") != nullptr;      // measurement switch: six zero planes cull nothing
    if (no_bound) memset(lv->h_mvps + (size_t)lv->S * 16, 0, 24 * sizeof(float));
    const uint32_t parity = ++lv->replay & 1u;  // which of its two counters the prep kernel's cull counts under (it zeroes the other one)
    memcpy(lv->h_mvps + (size_t)lv->S * 16 + 24, &parity, sizeof parity);
    for (int w = 0; w < lv->Wn; ++w) {
        const double *m = h_forward + (size_t)w * 9;
        const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
        if (det == 0.0) return NMI_ERR_INVALID_ARGUMENT;
        const double inv[9] = {(m[4] * m[8] - m[5] * m[7]) / det, (m[2] * m[7] - m[1] * m[8]) / det, (m[1] * m[5] - m[2] * m[4]) / det,
                               (m[5] * m[6] - m[3] * m[8]) / det, (m[0] * m[8] - m[2] * m[6]) / det, (m[2] * m[3] - m[0] * m[5]) / det,
                               (m[3] * m[7] - m[4] * m[6]) / det, (m[1] * m[6] - m[0] * m[7]) / det, (m[0] * m[4] - m[1] * m[3]) / det};
        for (int k = 0; k < 9; ++k) lv->h_coeffs[w * 9 + k] = (float)inv[k];
    }
    constexpr unsigned long long kPending = ~0ull;
    __atomic_store_n(lv->h_key, kPending, __ATOMIC_RELEASE);
    const hipError_t le = hipGraphLaunch(lv->exec, ctx->stream);
    if (le != hipSuccess) {
        --lv->replay;  // (the replay did not run: its counter has not been counted into, the other one has not been zeroed)
        return hip_fail(ctx, le, "hipGraphLaunch");
    }
    return NMI_OK;
}

// nmi_level_run_rccl's device side (nmi_capi_rccl.cpp): replay the block's graph (if the block is not empty) and hand back the
// device word that holds this rank's key once the context's stream has reached this point.
int nmi_internal::level_enqueue(nmi_level *lv, const float *h_mvps, const double *h_forward, const unsigned long long **d_key)
{
    if (!lv || !d_key) return NMI_ERR_INVALID_ARGUMENT;
    *d_key = lv->d_key;
    if (lv->S == 0 || lv->Wn == 0) return NMI_OK;  // d_key holds 0 = "no candidate" since creation
    if (!h_mvps || !h_forward) return NMI_ERR_INVALID_ARGUMENT;
    lv->ctx->detail.clear();
    return level_launch(lv, h_mvps, h_forward);
}

nmi_ctx *nmi_internal::level_ctx(nmi_level *lv) { return lv ? lv->ctx : nullptr; }

extern "C" {

int nmi_level_run(nmi_level *lv, const float *h_mvps, const double *h_forward, int64_t *h_best_index, float *h_best_score)
{
    if (!lv) return NMI_ERR_INVALID_ARGUMENT;
    if (lv->S == 0 || lv->Wn == 0) return nmi_key_unpack(0, h_best_index, h_best_score);  // empty block: no candidate
    if (!h_mvps || !h_forward) return NMI_ERR_INVALID_ARGUMENT;
    nmi_ctx *ctx = lv->ctx;
    ctx->detail.clear();
    DeviceGuard guard(ctx->device);
    // the previous replay has completed (this call is blocking), so the pinned parameter buffers are free to rewrite.
    // The search kernel's last workgroup stores the winner (never all ones: scores are non-negative floats) into
    // *h_key with system scope; polling that word returns ~10 us earlier than waiting for the stream to drain.
    constexpr unsigned long long kPending = ~0ull;
    const int lrc = level_launch(lv, h_mvps, h_forward);
    if (lrc != NMI_OK) return lrc;
    unsigned long long key = kPending;
    for (uint64_t spin = 0; key == kPending; ++spin) {
        key = __atomic_load_n(lv->h_key, __ATOMIC_ACQUIRE);
        if (key == kPending && (spin & 0xFFFF) == 0xFFFF) {
            const hipError_t q = hipStreamQuery(ctx->stream);  // a faulted or finished stream must not leave us spinning
            if (q == hipSuccess) {
                key = __atomic_load_n(lv->h_key, __ATOMIC_ACQUIRE);
                break;
            }
            if (q != hipErrorNotReady) return hip_fail(ctx, q, "hipStreamQuery");
        }
    }
    if (key == kPending) return NMI_ERR_HIP;  // the graph ran without posting: cannot happen with a non-empty grid
    return nmi_key_unpack(key, h_best_index, h_best_score);
}

int nmi_level_copy_outputs(nmi_level *lv, uint8_t *h_renders, uint8_t *h_warps, float *h_ratings)
{
    if (!lv) return NMI_ERR_INVALID_ARGUMENT;
    nmi_ctx *ctx = lv->ctx;
    if (lv->S == 0 || lv->Wn == 0) return NMI_OK;  // empty block: nothing was produced
    DeviceGuard guard(ctx->device);
    NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // nmi_level_run returns when the winner is posted, a little before the graph has drained
    const size_t npix = (size_t)ctx->npix;
    if (h_renders) NMI_HIP_TRY(ctx, hipMemcpy(h_renders, lv->d_renders, npix * lv->S, hipMemcpyDeviceToHost));
    if (h_warps) NMI_HIP_TRY(ctx, hipMemcpy(h_warps, lv->d_warps, npix * lv->Wn, hipMemcpyDeviceToHost));
    if (h_ratings) NMI_HIP_TRY(ctx, hipMemcpy(h_ratings, lv->d_ratings, (size_t)lv->S * lv->Wn * sizeof(float), hipMemcpyDeviceToHost));
    return NMI_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Streaming pipeline (config 5): double-buffered render stacks, copy stream beside the compute stream.
// ---------------------------------------------------------------------------------------------------------
}  // extern "C"

struct nmi_stream {
    nmi_ctx *ctx = nullptr;
    int depth = 0, max_S = 0, max_Wn = 0;
    hipStream_t copy = nullptr;
    struct Slot {
        uint8_t *d_renders = nullptr;
        float *d_ratings = nullptr;  // [max_Wn][max_S], only with nmi_stream_keep_ratings
        int S = 0, Wn = 0;           // grid of the slot's latest submission
        unsigned long long *d_key = nullptr;      // [2]: this rank's key, and the all-reduced one (block submissions with a communicator)
        unsigned long long *h_key = nullptr;
        hipEvent_t copied = nullptr, done = nullptr;
        int64_t ticket = -1;
        bool waited = true;
        bool failed = false;         // its search timed out in the split kernel and could not be redone (nmi_stream_wait)
        int parts = 0;               // split form of the slot's launch (0: nmi_grid_kernel) and its epoch: the launch answers
        uint32_t epoch = 0;          //   for itself when the ticket is waited for, whatever was launched after it
        int s_offset = 0, S_total = 0, w_offset = 0;  // position of the slot's block in its level
        int warp_buf = 0;            // warp buffer the search read, and that buffer's generation at submission
        uint64_t warp_gen = 0;
    };
    Slot *slots = nullptr;
    uint8_t *d_frame[2] = {nullptr, nullptr};  // frames alternate so an upload never overwrites one still being warped
    uint8_t *d_warps[2] = {nullptr, nullptr};
    hipEvent_t frame_copied = nullptr, warps_free[2] = {nullptr, nullptr};
    int warp_buf = 0;      // buffer holding the current warp stack
    uint64_t warp_gen[2] = {0, 0};  // refills of each warp buffer so far
    int cur_Wn = 0;
    bool have_warps = false;
    bool keep_ratings = false;
    int64_t next_ticket = 0;
};

extern "C" {

int nmi_stream_destroy(nmi_stream *st)
{
    if (!st) return NMI_OK;
    DeviceGuard guard(st->ctx->device);
    (void)hipStreamSynchronize(st->ctx->stream);
    if (st->copy) (void)hipStreamSynchronize(st->copy);
    for (int i = 0; st->slots && i < st->depth; ++i) {
        nmi_stream::Slot &s = st->slots[i];
        if (s.d_renders) (void)hipFree(s.d_renders);
        if (s.d_ratings) (void)hipFree(s.d_ratings);
        if (s.d_key) (void)hipFree(s.d_key);
        if (s.h_key) (void)hipHostFree(s.h_key);
        if (s.copied) (void)hipEventDestroy(s.copied);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    delete[] st->slots;
    for (int b = 0; b < 2; ++b) {
        if (st->d_frame[b]) (void)hipFree(st->d_frame[b]);
        if (st->d_warps[b]) (void)hipFree(st->d_warps[b]);
        if (st->warps_free[b]) (void)hipEventDestroy(st->warps_free[b]);
    }
    if (st->frame_copied) (void)hipEventDestroy(st->frame_copied);
    if (st->copy) (void)hipStreamDestroy(st->copy);
    delete st;
    return NMI_OK;
}

int nmi_stream_create(nmi_ctx *ctx, int32_t max_S, int32_t max_Wn, int32_t depth, nmi_stream **out)
{
    if (!ctx || !out || max_S <= 0 || max_Wn <= 0 || depth < 2 || depth > 64) return NMI_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    DeviceGuard guard(ctx->device);
    nmi_stream *st = new (std::nothrow) nmi_stream;
    if (!st) return NMI_ERR_INVALID_ARGUMENT;
    st->ctx = ctx;
    st->depth = depth;
    st->max_S = max_S;
    st->max_Wn = max_Wn;
    st->slots = new (std::nothrow) nmi_stream::Slot[depth];
    const size_t npix = (size_t)ctx->npix;
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t r) {
        if (e == hipSuccess) e = r;
        return r == hipSuccess;
    };
    ok(hipStreamCreateWithFlags(&st->copy, hipStreamNonBlocking));
    for (int i = 0; st->slots && i < depth && e == hipSuccess; ++i) {
        nmi_stream::Slot &s = st->slots[i];
        ok(hipMalloc((void **)&s.d_renders, npix * max_S));
        ok(hipMalloc((void **)&s.d_key, 2 * sizeof(unsigned long long)));
        ok(hipHostMalloc((void **)&s.h_key, sizeof(unsigned long long), hipHostMallocDefault));
        ok(hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
        ok(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    }
    for (int b = 0; b < 2 && e == hipSuccess; ++b) {
        ok(hipMalloc((void **)&st->d_frame[b], npix));
        ok(hipMalloc((void **)&st->d_warps[b], npix * max_Wn));
        ok(hipEventCreateWithFlags(&st->warps_free[b], hipEventDisableTiming));
    }
    ok(hipEventCreateWithFlags(&st->frame_copied, hipEventDisableTiming));
    if (!st->slots || e != hipSuccess) {
        const int rc = st->slots ? hip_fail(ctx, e, "nmi_stream_create") : NMI_ERR_INVALID_ARGUMENT;
        nmi_stream_destroy(st);
        return rc;
    }
    *out = st;
    return NMI_OK;
}

int nmi_stream_submit(nmi_stream *st, const uint8_t *h_render_stack, int32_t S, const uint8_t *h_frame,
                      const double *h_forward, int32_t Wn, int64_t *ticket)
{
    if (S <= 0) return NMI_ERR_INVALID_ARGUMENT;
    return nmi_stream_submit_block(st, h_render_stack, S, 0, S, h_frame, h_forward, Wn, 0, h_frame ? Wn : (st ? st->cur_Wn : 0), nullptr, ticket);
}

int nmi_stream_submit_block(nmi_stream *st, const uint8_t *h_render_stack, int32_t S, int32_t s_offset, int32_t S_total,
                            const uint8_t *h_frame, const double *h_forward, int32_t Wn, int32_t w_offset, int32_t Wn_total,
                            void *nccl_comm, int64_t *ticket)
{
    if (!st || !ticket || S < 0 || S > st->max_S || (S > 0 && !h_render_stack)) return NMI_ERR_INVALID_ARGUMENT;
    if (h_frame && (!h_forward || Wn <= 0 || Wn > st->max_Wn)) return NMI_ERR_INVALID_ARGUMENT;
    if (!h_frame && !st->have_warps && !(S == 0 && nccl_comm)) return NMI_ERR_INVALID_ARGUMENT;
    const int32_t Wn_block = h_frame ? Wn : st->cur_Wn;
    if (s_offset < 0 || w_offset < 0 || s_offset + S > S_total || w_offset + Wn_block > Wn_total) return NMI_ERR_INVALID_ARGUMENT;
    if ((int64_t)S_total * Wn_total >= 0x7FFFFFFFll) return NMI_ERR_UNSUPPORTED;
    nmi_ctx *ctx = st->ctx;
    ctx->detail.clear();
    DeviceGuard guard(ctx->device);
    const int64_t t = st->next_ticket;
    nmi_stream::Slot &s = st->slots[t % st->depth];
    if (!s.waited) return NMI_ERR_NOT_READY;  // the ticket that used this slot has not been collected yet
    const size_t npix = (size_t)ctx->npix;

    // copy stream: render stack of this level into the slot (the slot's previous search finished: it was waited for)
    if (S > 0) NMI_HIP_TRY(ctx, hipMemcpyAsync(s.d_renders, h_render_stack, npix * S, hipMemcpyHostToDevice, st->copy));
    if (h_frame) {
        const int nb = st->have_warps ? st->warp_buf ^ 1 : 0;
        // the buffer being refilled was last read by searches submitted before the previous frame switch
        NMI_HIP_TRY(ctx, hipStreamWaitEvent(st->copy, st->warps_free[nb], 0));
        NMI_HIP_TRY(ctx, hipMemcpyAsync(st->d_frame[nb], h_frame, npix, hipMemcpyHostToDevice, st->copy));
        NMI_HIP_TRY(ctx, hipEventRecord(st->frame_copied, st->copy));
        NMI_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, st->frame_copied, 0));
        if (st->have_warps) NMI_HIP_TRY(ctx, hipEventRecord(st->warps_free[st->warp_buf], ctx->stream));
        int rc = nmi_warp_stack(ctx, st->d_frame[nb], h_forward, Wn, st->d_warps[nb]);
        if (rc != NMI_OK) return rc;
        st->warp_buf = nb;
        ++st->warp_gen[nb];
        st->cur_Wn = Wn;
        st->have_warps = true;
    }
    NMI_HIP_TRY(ctx, hipEventRecord(s.copied, st->copy));

    // compute stream: search on the slot, winner to pinned host memory
    NMI_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, s.copied, 0));
    if (st->keep_ratings && !s.d_ratings)
        NMI_HIP_TRY(ctx, hipMalloc((void **)&s.d_ratings, (size_t)st->max_S * st->max_Wn * sizeof(float)));
    s.S = S;
    s.Wn = st->cur_Wn;
    // nmi_stream_wait checks this very launch for a split-kernel timeout (parts, epoch below) and redoes it -- which it cannot do
    // once the key has gone into a collective, so submissions with a communicator keep to the one-workgroup kernel
    ctx->allow_unchecked_split = nccl_comm == nullptr;
    s.s_offset = s_offset;
    s.S_total = S_total;
    s.w_offset = w_offset;
    int rc = enqueue_grid(ctx, s.d_renders, S, s_offset, S_total, st->d_warps[st->warp_buf], st->cur_Wn, st->keep_ratings ? s.d_ratings : nullptr,
                          s.d_key, false, nullptr, nullptr, nullptr, nullptr, w_offset);
    ctx->allow_unchecked_split = false;
    if (rc != NMI_OK) return rc;
    s.parts = ctx->last_parts;
    s.epoch = ctx->last_epoch;
    s.warp_buf = st->warp_buf;
    s.warp_gen = st->warp_gen[st->warp_buf];
    const unsigned long long *result = s.d_key;
    if (nccl_comm) {
        // the level's only exchange: 8-byte MAX all-reduce of the packed keys, issued in submission order on every rank
        rc = rccl_allreduce_key(ctx, s.d_key, s.d_key + 1, nccl_comm);
        if (rc != NMI_OK) return rc;
        result = s.d_key + 1;
    }
    NMI_HIP_TRY(ctx, hipMemcpyAsync(s.h_key, result, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    NMI_HIP_TRY(ctx, hipEventRecord(s.done, ctx->stream));
    s.ticket = t;
    s.waited = false;
    s.failed = false;
    *ticket = t;
    ++st->next_ticket;
    return NMI_OK;
}

int nmi_stream_keep_ratings(nmi_stream *st, int32_t enabled)
{
    if (!st) return NMI_ERR_INVALID_ARGUMENT;
    st->keep_ratings = enabled != 0;
    return NMI_OK;
}

int nmi_stream_copy_ratings(nmi_stream *st, int64_t ticket, float *h_ratings, int64_t n)
{
    if (!st || !h_ratings || ticket < 0 || ticket >= st->next_ticket) return NMI_ERR_INVALID_ARGUMENT;
    nmi_stream::Slot &s = st->slots[ticket % st->depth];
    // valid from nmi_stream_wait(ticket) until the slot is submitted to again
    if (s.ticket != ticket || !s.waited || s.failed || !s.d_ratings || n != (int64_t)s.S * s.Wn) return NMI_ERR_INVALID_ARGUMENT;
    nmi_ctx *ctx = st->ctx;
    DeviceGuard guard(ctx->device);
    NMI_HIP_TRY(ctx, hipMemcpy(h_ratings, s.d_ratings, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return NMI_OK;
}

int nmi_stream_wait(nmi_stream *st, int64_t ticket, int64_t *h_best_index, float *h_best_score)
{
    if (!st || ticket < 0 || ticket >= st->next_ticket) return NMI_ERR_INVALID_ARGUMENT;
    nmi_stream::Slot &s = st->slots[ticket % st->depth];
    if (s.ticket != ticket || s.waited) return NMI_ERR_INVALID_ARGUMENT;  // overwritten or already collected
    nmi_ctx *ctx = st->ctx;
    DeviceGuard guard(ctx->device);
    NMI_HIP_TRY(ctx, hipEventSynchronize(s.done));
    s.waited = true;
    if (split_launch_failed(ctx, s.parts, s.epoch)) {
        // This ticket's search (a small grid on the split kernel) timed out in a hand-off.  Its render stack is still in the
        // slot; if its warp stack is too (no later frame has refilled that buffer) the search is redone here, behind
        // whatever was submitted since, by nmi_grid_kernel (the split forms are paused now).  Otherwise the ticket fails:
        // NMI_ERR_NOT_READY, its rating table is withheld, and the caller submits the level again.
        if (st->warp_gen[s.warp_buf] != s.warp_gen) {
            s.failed = true;
            ctx->detail = "split kernel hand-off timed out and the ticket's warp stack is gone: submit the level again";
            return NMI_ERR_NOT_READY;
        }
        const int rc = enqueue_grid(ctx, s.d_renders, s.S, s.s_offset, s.S_total, st->d_warps[s.warp_buf], s.Wn,
                                    st->keep_ratings ? s.d_ratings : nullptr, s.d_key, false, nullptr, nullptr, nullptr, nullptr, s.w_offset);
        if (rc != NMI_OK) {
            s.failed = true;
            return rc;
        }
        s.parts = 0;
        NMI_HIP_TRY(ctx, hipMemcpyAsync(s.h_key, s.d_key, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        ctx->detail = "split kernel hand-off timed out; ticket redone by the one-workgroup kernel, split forms paused (nmi_split_status)";
    }
    return nmi_key_unpack(*s.h_key, h_best_index, h_best_score);
}

}  // extern "C"
