// nmi_capi.cpp -- the C ABI declared in include/nmi_hip.h on top of the gfx950 kernels: context, options, search.
//
// Host orchestration that replaces CUDAF::NMIWithCuda_noMask (Thirdparty/CUDA_Functions/kernel.cu:49-114)
// and the candidate loop + arg-max of Tracking::RelocalizeWithNMI (src/Tracking.cc:1879-1905,1952):
// a persistent context owns every buffer, one launch scores a whole candidate grid, and the only
// host<->device traffic per search is one 8-byte key.
#include <sched.h>

#include "nmi_ctx.h"

namespace nmi_internal {

int hip_fail(nmi_ctx *ctx, hipError_t e, const char *what)
{
    if (ctx) {
        char buf[256];
        snprintf(buf, sizeof buf, "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
        ctx->detail = buf;
    }
    return NMI_ERR_HIP - (int)e;
}



// Visiting order of the candidates of an S x Wn grid: tiles of kTileW warps x kTileS renders (32 candidates = the
// 32 workgroups one XCD runs at a time), tile after tile; within a tile render-fastest.  Cached per grid shape.
void build_order(int S, int Wn, int *order)
{
    const int tile_s = S >= 8 ? 8 : (S >= 4 ? 4 : (S >= 2 ? 2 : 1)), tile_w = 32 / tile_s;
    int64_t o = 0;
    for (int w0 = 0; w0 < Wn; w0 += tile_w)
        for (int s0 = 0; s0 < S; s0 += tile_s)
            for (int w = w0; w < w0 + tile_w && w < Wn; ++w)
                for (int s = s0; s < s0 + tile_s && s < S; ++s) order[o++] = w * S + s;
}

// The cached visiting order for this grid shape (device pointer in *d_order), built and uploaded on first use.  Never
// synchronises the stream on a hit or on a miss with a free entry; only evicting the least recently used of kOrderCache
// shapes waits (its table may still be read by a kernel in flight).
int ensure_order(nmi_ctx *ctx, int S, int Wn, const int **d_order)
{
    *d_order = nullptr;
    if (!ctx->xcd_tiling) return NMI_OK;
    const int64_t total = (int64_t)S * Wn;
    nmi_ctx::OrderEntry *victim = nullptr;
    for (auto &e : ctx->orders) {
        if (e.S == S && e.Wn == Wn) {
            e.last_use = ++ctx->order_clock;
            *d_order = e.d;
            return NMI_OK;
        }
        if (!victim || (e.d == nullptr && victim->d != nullptr) || ((e.d == nullptr) == (victim->d == nullptr) && e.last_use < victim->last_use))
            victim = &e;
    }
    nmi_ctx::OrderEntry &e = *victim;
    if (e.d) NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // eviction: the old table / staging copy may be in use
    if (total > e.cap) {
        if (e.d) NMI_HIP_TRY(ctx, hipFree(e.d));
        if (e.h) NMI_HIP_TRY(ctx, hipHostFree(e.h));
        e.d = e.h = nullptr;
        e.cap = 0;
        e.S = e.Wn = -1;
        NMI_HIP_TRY(ctx, hipMalloc((void **)&e.d, (size_t)total * sizeof(int)));
        NMI_HIP_TRY(ctx, hipHostMalloc((void **)&e.h, (size_t)total * sizeof(int), hipHostMallocDefault));
        e.cap = total;
    }
    build_order(S, Wn, e.h);
    NMI_HIP_TRY(ctx, hipMemcpyAsync(e.d, e.h, (size_t)total * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    e.S = S;
    e.Wn = Wn;
    e.last_use = ++ctx->order_clock;
    *d_order = e.d;
    return NMI_OK;
}

// How the split kernel should cut each candidate of a launch of `total` candidates on `cap` workgroups: *parts row
// parts and *pix_parts pixel ranges.  *parts = 0: use the one-workgroup-per-candidate kernel.  Automatic choice (256
// CUs): up to 8 candidates 8 x 4, up to 16: 8 x 2, up to 32: 4 x 2, up to 64: 4 x 1 -- a part's time is its pixel stream
// (>= 12 us for a whole 640x480 pair whatever the number of row parts), so pixel ranges come first and 2 row parts,
// measured no faster than none, are available on request only.
static void choose_split(const nmi_ctx *ctx, int64_t total, int cap, int *parts, int *pix_parts)
{
    *parts = 0;
    *pix_parts = 1;
    if (cap > ctx->compute_units) cap = ctx->compute_units;  // all workgroups of a split launch must be resident at once
    if (ctx->hist_variant != 3 || ctx->split_mode == 0 || ctx->split_mode == 1 || total <= 0 || total > cap) return;
    auto fits = [&](int k, int p) { return nmi::split_workgroups((int)total, k, p) <= cap; };
    auto exists = [](int k, int p) { return p == 1 || (k == 8 && (p == 2 || p == 4)) || (k == 4 && p == 2); };
    const int want_p = ctx->split_pixels;  // -1 automatic, 1 never, 2 / 4 that many when it fits
    if (ctx->split_mode > 0) {
        const int k = ctx->split_mode;
        if (!fits(k, 1)) return;
        *parts = k;
        if (want_p == 1) return;
        for (int p = 4; p >= 2; p >>= 1)
            if ((want_p == -1 || want_p == p) && exists(k, p) && fits(k, p)) {
                *pix_parts = p;
                return;
            }
        return;
    }
    static const int order[][2] = {{8, 4}, {8, 2}, {4, 2}, {8, 1}, {4, 1}};
    for (const auto &kp : order) {
        if (kp[1] > 1 && want_p != -1 && want_p != kp[1]) continue;
        if (fits(kp[0], kp[1])) {
            *parts = kp[0];
            *pix_parts = kp[1];
            return;
        }
    }
}

// Pixel ranges per candidate for nmi_pix_kernel (nmi_pix_kernel.hip), 0 = another kernel.  The owner of a candidate adds
// its P - 1 helpers' histograms to its own and waits for the slowest of them, so P grows only while the histogram phase
// (21 us / P at 640x480) shrinks faster: automatic choice 3 up to 85 candidates, 2 up to 128 (256 CUs; measured, with 4 and
// 5: profiles/r04_a/small_grid_time.txt); smaller grids keep the row-split forms, larger ones have no CU to spare.  NMI_OPT_SPLIT 1 + NMI_OPT_SPLIT_PIXELS P forces P wherever it fits.
int choose_pix(const nmi_ctx *ctx, const nmi::GridArgs &a, int64_t total, int cap)
{
    if (cap > ctx->compute_units) cap = ctx->compute_units;  // (an owner that waits for a CU starts a second round)
    if (ctx->hist_variant != 3 || a.width < 32 || (ctx->shift != 0 && !ctx->params.use_bg) || ctx->pair_renders || total <= 0) return 0;
    if ((ctx->phase_mask & ~512) != 3 || (a.dbg_stamps != nullptr && ctx->split_mode != 1)) return 0;
    if (ctx->split_mode == 1) {
        const int p = ctx->split_pixels;
        return (p >= 2 && p <= nmi::pix_max_ranges() && total * p <= cap) ? p : 0;
    }
    if (ctx->split_mode != -1 || ctx->split_pixels != -1 || total * 2 > cap) return 0;
    const int p = (int)(cap / total);
    // Frames whose rows are not whole aligned 16-byte chunks (width % 16 != 0, unaligned stacks): the row-split kernel would
    // read them byte by byte, this one has the unaligned-row form -- so small grids and single pairs come here too, with more ranges
    if (!a.vec_ok) return p > nmi::pix_max_ranges() ? nmi::pix_max_ranges() : p;
    if (total <= 32) return 0;
    return p > 3 ? 3 : p;
}

// the owner's share of a candidate's pixels: an equal one plus what it can add while its helpers' counters travel
double pix_owner_share(const nmi_ctx *ctx, int pix)
{
    const double owner_px = ((double)ctx->npix + (double)(pix - 1) * ctx->pix_owner_bias) / pix;
    return owner_px < ctx->npix ? owner_px / ctx->npix : 1.0;
}

int ensure_pix_timeouts(nmi_ctx *ctx)
{
    if (ctx->d_pix_timeouts) return NMI_OK;
    NMI_HIP_TRY(ctx, hipMalloc((void **)&ctx->d_pix_timeouts, sizeof(uint32_t)));
    NMI_HIP_TRY(ctx, hipMemsetAsync(ctx->d_pix_timeouts, 0, sizeof(uint32_t), ctx->stream));
    return NMI_OK;
}

// Epoch of the next split launch.  Slab granules carry all 32 bits, block granules the low 16: neither may be 0 (the
// cleared state), and whenever the low 16 bits wrap the blocks are cleared so that no granule older than 65535 launches
// can show the current tag (the slabs likewise when all 32 bits wrap).
int next_split_epoch(nmi_ctx *ctx, uint32_t *epoch)
{
    uint32_t e = ctx->split_epoch + 1;
    if ((e & 0xFFFFu) == 0) {
        if (ctx->d_blocks) NMI_HIP_TRY(ctx, hipMemsetAsync(ctx->d_blocks, 0, ctx->blocks_bytes, ctx->stream));
        if (ctx->d_pix_blocks) NMI_HIP_TRY(ctx, hipMemsetAsync(ctx->d_pix_blocks, 0, ctx->pix_blocks_bytes, ctx->stream));  // (its tags: 31 bits)
        if (e == 0 && ctx->d_slabs) NMI_HIP_TRY(ctx, hipMemsetAsync(ctx->d_slabs, 0, (size_t)ctx->slab_cap * sizeof(nmi::SplitSlab), ctx->stream));
        ++e;
    }
    ctx->split_epoch = *epoch = e;
    return NMI_OK;
}

static int ensure_slabs(nmi_ctx *ctx, int n)
{
    if (n <= ctx->slab_cap) return NMI_OK;
    if (ctx->d_slabs) {
        NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        NMI_HIP_TRY(ctx, hipFree(ctx->d_slabs));
        ctx->d_slabs = nullptr;
        ctx->slab_cap = 0;
    }
    const int cap = n > 128 ? n : 128;
    NMI_HIP_TRY(ctx, hipMalloc((void **)&ctx->d_slabs, (size_t)cap * sizeof(nmi::SplitSlab)));
    NMI_HIP_TRY(ctx, hipMemsetAsync(ctx->d_slabs, 0, (size_t)cap * sizeof(nmi::SplitSlab), ctx->stream));  // epoch 0 = never written
    ctx->slab_cap = cap;
    return NMI_OK;
}

// nmi_pix_kernel's hand-off blocks: a buffer of their own -- the row-split kernel tags its granules with 16 bits in the top of
// an 8-byte word, which a packed counter of the other kernel's layout can equal.
static int ensure_pix_blocks(nmi_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->pix_blocks_bytes) return NMI_OK;
    if (ctx->d_pix_blocks) {
        NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        NMI_HIP_TRY(ctx, hipFree(ctx->d_pix_blocks));
        ctx->d_pix_blocks = nullptr;
        ctx->pix_blocks_bytes = 0;
    }
    NMI_HIP_TRY(ctx, hipMalloc((void **)&ctx->d_pix_blocks, bytes));
    NMI_HIP_TRY(ctx, hipMemsetAsync(ctx->d_pix_blocks, 0, bytes, ctx->stream));  // tag 0 = never written
    ctx->pix_blocks_bytes = bytes;
    return NMI_OK;
}

static int ensure_blocks(nmi_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->blocks_bytes) return NMI_OK;
    if (ctx->d_blocks) {
        NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        NMI_HIP_TRY(ctx, hipFree(ctx->d_blocks));
        ctx->d_blocks = nullptr;
        ctx->blocks_bytes = 0;
    }
    NMI_HIP_TRY(ctx, hipMalloc((void **)&ctx->d_blocks, bytes));
    NMI_HIP_TRY(ctx, hipMemsetAsync(ctx->d_blocks, 0, bytes, ctx->stream));  // tag 0 = never written
    ctx->blocks_bytes = bytes;
    return NMI_OK;
}

// Enqueues the grid kernel (one launch, nothing else).  No synchronisation.
int enqueue_grid(nmi_ctx *ctx, const uint8_t *render_stack, int S_local, int s_offset, int S_total,
                 const uint8_t *warp_stack, int Wn, float *d_ratings, unsigned long long *out_key, bool post,
                 uint32_t *dbg_joint, uint32_t *dbg_h1, uint32_t *dbg_h2, float *dbg_sums, int w_offset, bool post_score)
{
    const nmi_params &p = ctx->params;
    nmi::GridArgs a{};
    a.render_stack = render_stack;
    a.warp_stack = warp_stack;
    a.S_local = S_local;
    a.Wn = Wn;
    a.s_offset = s_offset;
    a.S_total = S_total;
    a.w_offset = w_offset;
    nmi::set_geometry(a, p.width, p.height, render_stack, warp_stack, p.render_bottom_up != 0);
    if (ctx->pair_renders) {  // nmi_eval_pairs: per-pair pointers; the 16-byte path needs every one of them aligned
        a.pair_renders = ctx->pair_renders;
        a.pair_warps = ctx->pair_warps;
        uintptr_t bits = 0;
        for (int i = 0; i < S_local; ++i) bits |= (uintptr_t)ctx->pair_renders_host[i] | (uintptr_t)ctx->pair_warps_host[i];
        if (bits % 16) nmi::set_geometry(a, p.width, p.height, (const void *)1, (const void *)1, p.render_bottom_up != 0);
    }
    a.shift = ctx->shift;
    a.mode = p.mode;
    a.table = ctx->table;
    a.scratch = ctx->d_scratch;
    a.order = nullptr;
    a.ratings = d_ratings;
    a.key = ctx->d_keys + ctx->slot;
    a.reset_key = ctx->d_keys + (ctx->slot ^ 1);
    a.out_key = out_key;
    a.done = ctx->d_done;
    // Only launches whose winner the host will poll for post to the mailbox (one bit of sequence is enough
    // because those calls are blocking, hence strictly alternating).  The sequence numbers, the key-slot flip and the
    // "posted" flag are committed only once the launch has been accepted: a failed launch leaves the protocol in step.
    // does somebody look for a split-kernel timeout after this launch?  (blocking calls, nmi_eval_pairs, stream tickets, RCCL form)
    const bool split_checked = post || post_score || ctx->allow_unchecked_split || ctx->pair_renders != nullptr;
    post = post && ctx->result_path == 1;
    post_score = post_score && ctx->result_path == 1;
    a.mailbox = post ? ctx->mailbox : nullptr;
    a.score_post = post_score ? ctx->score_mailbox : nullptr;
    a.seq = post ? ctx->seq + 1 : (post_score ? ctx->pair_seq + 1 : 0);
    a.dbg_joint = dbg_joint;
    a.dbg_h1 = dbg_h1;
    a.dbg_h2 = dbg_h2;
    a.dbg_sums = dbg_sums;
    a.hist_variant = ctx->hist_variant;
    a.phase_mask = ctx->phase_mask;
    a.dbg_stamps = ctx->dbg_stamps;

    const int64_t total = (int64_t)S_local * Wn;
    if (total == 0) {
        // nothing to score: the winner is "none" (key 0); publish it the way the kernel would.  The key slot keeps its
        // "zero on entry" state for the next launch (nothing ever writes a winner into a ping-pong slot from outside:
        // the RCCL form reduces into ctx->d_reduced_key).
        if (out_key) NMI_HIP_TRY(ctx, hipMemsetAsync(out_key, 0, sizeof(unsigned long long), ctx->stream));
        NMI_HIP_TRY(ctx, hipMemsetAsync(ctx->d_keys + ctx->slot, 0, sizeof(unsigned long long), ctx->stream));
        NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (post) {
            ++ctx->seq;
            ctx->mailbox->word = (unsigned long long)(ctx->seq & 1u) << 63;
        }
        ctx->posted = post;
        ctx->last_slot = ctx->slot;
        ctx->last_parts = 0;
        ctx->last_pix = 0;
        ctx->last_epoch = 0;
        return NMI_OK;
    }
    const int cap = ctx->workgroups > 0 ? ctx->workgroups : ctx->compute_units;
    int parts = 0, pix_parts = 1;
    // the split kernel's consumers wait for their producers inside the launch: all its workgroups must be able to run at
    // once, i.e. no more of them than compute units (each takes a whole CU)
    choose_split(ctx, total, cap, &parts, &pix_parts);
    if (pix_parts > 1 && ctx->npix >= (1 << 24)) pix_parts = 1;  // block granules hold 24-bit counts
    int pix = choose_pix(ctx, a, total, cap);  // mid-size grids: pixel ranges (no residence condition, heals itself)
    if (pix) parts = 0;
    if (parts && !split_checked) parts = 0;  // enqueue-only call: nobody would notice a timed-out hand-off, so no split kernel
    if (parts && ctx->split_cooldown > 0) {  // after a timeout: nmi_grid_kernel for a while, then the split forms again
        --ctx->split_cooldown;
        parts = 0;
    }
    if (!parts) pix_parts = 1;
    // Few-levels path (nmi_fewlevels_kernel.hip).  The decision rests on what the most recent probe of a search's stacks
    // found (frames and renders of consecutive searches look alike); it is only a matter of speed, because the probe that
    // goes with every few-levels launch hands the search back to nmi_grid_kernel (gated launch below) when this
    // search's stacks do not qualify.
    bool few = false;
    const bool few_eligible = !parts && !(pix && ctx->split_mode == 1) && a.vec_ok && (ctx->shift == 0 || p.use_bg) && ctx->hist_variant == 3 && ctx->phase_mask == 3 && !dbg_joint &&
                              !dbg_h1 && !dbg_h2 && !dbg_sums && !ctx->pair_renders && !ctx->dbg_stamps && ctx->content_path != 0;
    if (few_eligible) {
        // What the most recent search found in its stacks (nr, nw), posted by the device: by the probe that goes with every
        // few-levels launch, or by nmi_grid_kernel itself -- every general search counts the bins of its candidates'
        // marginals on the way (publish_seen / post_seen), so a change of content shows after ONE search, with no extra launch.
        const unsigned long long posted = __atomic_load_n(ctx->level_post, __ATOMIC_ACQUIRE);
        if ((uint32_t)(posted >> 32) != ctx->level_seen) {
            ctx->level_seen = (uint32_t)(posted >> 32);
            const uint32_t joint = (uint32_t)((posted >> 16) & 0xFFFFu) * (uint32_t)(posted & 0xFFFFu);
            ctx->few_hint = joint > 0 && joint <= (uint32_t)ctx->fewlevels_bins;
        }
        few = ctx->content_path == 1 || ctx->few_hint;
    }
    if (few) pix = 0;  // few distinct intensities: the few-levels kernels are the faster ones at any grid size
    if (ctx->pair_renders && !parts) return NMI_ERR_UNSUPPORTED;  // per-pair pointers exist in the split kernel only (nmi_eval_pairs decides first)
    int workgroups = (int)(total < cap ? total : cap);
    if (parts) {
        int rs = ensure_slabs(ctx, (int)total);
        if (rs == NMI_OK && pix_parts > 1) rs = ensure_blocks(ctx, (size_t)total * nmi::split_block_bytes_per_candidate(pix_parts));
        if (rs != NMI_OK) return rs;
        if (rs == NMI_OK) rs = next_split_epoch(ctx, &a.epoch);
        if (rs != NMI_OK) return rs;
        a.slabs = ctx->d_slabs;
        a.blocks = ctx->d_blocks;
        a.split_error = ctx->d_split_error;
        workgroups = nmi::split_workgroups((int)total, parts, pix_parts);
    } else if (pix) {
        int rs = ensure_pix_blocks(ctx, nmi::pix_block_bytes((int)total, pix));
        if (rs == NMI_OK) rs = next_split_epoch(ctx, &a.epoch);
        if (rs == NMI_OK) rs = ensure_pix_timeouts(ctx);
        if (rs != NMI_OK) return rs;
        a.blocks = ctx->d_pix_blocks;
        workgroups = (int)total * pix;
    } else if (ctx->xcd_tiling && total <= (1ll << 24)) {  // 4 B per candidate
        const int orc = ensure_order(ctx, S_local, Wn, &a.order);
        if (orc != NMI_OK) return orc;
    }
#ifdef NMI_BUILD_ABLATIONS
    if (ctx->hist_variant == 4 && workgroups > ctx->scratch_workgroups) {
        NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_scratch) NMI_HIP_TRY(ctx, hipFree(ctx->d_scratch));
        ctx->d_scratch = nullptr;
        ctx->scratch_workgroups = 0;
        const int alloc = workgroups > ctx->compute_units ? workgroups : ctx->compute_units;
        NMI_HIP_TRY(ctx, hipMalloc((void **)&ctx->d_scratch, nmi::grid_kernel_scratch_bytes(alloc)));
        ctx->scratch_workgroups = alloc;
        a.scratch = ctx->d_scratch;
    }
#endif
    if (!parts && !few && ctx->hist_variant == 3 && ctx->content_path != 0) a.plan = ctx->d_plan;  // the search doubles as a probe
    if (few) {
        const size_t need = (size_t)(S_local + Wn) * (size_t)ctx->npix;
        if (need > ctx->rank_bytes) {
            NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->d_rank_stacks) NMI_HIP_TRY(ctx, hipFree(ctx->d_rank_stacks));
            ctx->d_rank_stacks = nullptr;
            ctx->rank_bytes = 0;
            // with headroom: a coarse-to-fine search alternates between grid shapes, and growing drains the stream
            const size_t want = need + need / 2;
            NMI_HIP_TRY(ctx, hipMalloc((void **)&ctx->d_rank_stacks, want));
            ctx->rank_bytes = want;
        }
        a.plan = ctx->d_plan;
    }
    if (ctx->profiling) NMI_HIP_TRY(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
    if (few)
        NMI_HIP_TRY(ctx, nmi::launch_levels(render_stack, S_local, warp_stack, Wn, ctx->npix, ctx->shift, ctx->d_plan, ctx->level_post, ++ctx->level_seq,
                                            (uint32_t)ctx->fewlevels_bins, few, ctx->stream));
    if (parts) {
        NMI_HIP_TRY(ctx, nmi::launch_split(a, parts, pix_parts, workgroups, p.use_bg != 0, ctx->stream));
    } else if (pix) {
        NMI_HIP_TRY(ctx, nmi::launch_pix(a, pix, pix_owner_share(ctx, pix), p.use_bg != 0, nullptr, ctx->d_pix_timeouts, ctx->stream));
    } else if (few) {
        NMI_HIP_TRY(ctx, nmi::launch_fewlevels(a, ctx->d_rank_stacks, ctx->d_rank_stacks + (size_t)S_local * ctx->npix, workgroups,
                                               p.use_bg != 0, ctx->stream));
        NMI_HIP_TRY(ctx, nmi::launch_grid_gated(a, workgroups, p.use_bg != 0, ctx->stream));
    } else {
        NMI_HIP_TRY(ctx, nmi::launch_grid(a, workgroups, p.use_bg != 0, ctx->stream));
    }
    ctx->last_few = few ? 1 : 0;
    // accepted: commit the protocol state
    if (post) ++ctx->seq;
    if (post_score) ++ctx->pair_seq;
    ctx->posted = post;
    ctx->last_slot = ctx->slot;
    ctx->slot ^= 1;
    ctx->last_parts = parts;
    ctx->last_pix = pix;
    ctx->last_epoch = parts ? a.epoch : 0;
    if (ctx->profiling) {
        NMI_HIP_TRY(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
        ctx->have_timing = true;
    }
    return NMI_OK;
}

// A hand-off of the split kernel timed out (its workgroups could not all run at once -- e.g. the device exposes fewer
// compute units to this process than it reports): wait for the launch to drain, switch the split forms off for this
// context and tell the caller to redo the call, which then goes through nmi_grid_kernel.
bool split_launch_failed(nmi_ctx *ctx, int parts, uint32_t epoch)
{
    if (!parts) return false;
    uint32_t *word = ctx->h_split_error + (epoch % nmi::kSplitRing);
    if (__atomic_load_n(word, __ATOMIC_ACQUIRE) != epoch) {
        // a split launch that went through re-arms the short cooldown -- if it was issued AFTER the last pause was set: a sibling
        // of the failed launch (same nmi_eval_pairs batch, stream tickets already in flight) says nothing about the retry
        if ((int32_t)(epoch - ctx->split_cooldown_epoch) > 0) ctx->split_backoff = nmi_ctx::kSplitBackoffMin;
        return false;
    }
    (void)hipStreamSynchronize(ctx->stream);
    __atomic_store_n(word, 0u, __ATOMIC_RELEASE);
    ++ctx->split_timeouts;
    ctx->split_cooldown = ctx->split_backoff;
    ctx->split_cooldown_epoch = ctx->split_epoch;
    if (ctx->split_backoff < nmi_ctx::kSplitBackoffMax) ctx->split_backoff *= 2;
    return true;
}

// Call once the most recent launch's result has arrived (the kernel raises the flag before it posts anything).
bool split_timed_out(nmi_ctx *ctx) { return split_launch_failed(ctx, ctx->last_parts, ctx->last_epoch); }

// what nmi_last_error_detail says after a call that was redone because of such a timeout (the call itself succeeded)
static const char *const kSplitTimeoutNote = "split kernel hand-off timed out; call redone by the one-workgroup kernel, split forms paused (nmi_split_status)";

// Polls a pinned host word until (word & mask) == want; *out receives the word.  NMI_OPT_WAIT_MODE 0 spins (lowest
// latency; occupies the calling core for the duration of the search), 1 yields the core between polls (the Tracking
// thread shares the host with LocalMapping / LoopClosing).  A faulted or drained stream ends the wait: returns
// NMI_ERR_NOT_READY when the stream is idle and the word still does not match.
int wait_word(nmi_ctx *ctx, const volatile unsigned long long *word, unsigned long long mask, unsigned long long want,
              unsigned long long *out)
{
    const uint64_t check_every = ctx->wait_mode == 1 ? 0x3F : 0xFFFF;
    for (uint64_t spin = 0;; ++spin) {
        const unsigned long long v = __atomic_load_n(word, __ATOMIC_ACQUIRE);
        if ((v & mask) == want) {
            *out = v;
            return NMI_OK;
        }
        if (ctx->wait_mode == 1) sched_yield();
        if ((spin & check_every) == check_every) {
            const hipError_t q = hipStreamQuery(ctx->stream);
            if (q == hipSuccess) {
                const unsigned long long v2 = __atomic_load_n(word, __ATOMIC_ACQUIRE);
                *out = v2;
                return (v2 & mask) == want ? NMI_OK : NMI_ERR_NOT_READY;
            }
            if (q != hipErrorNotReady) return hip_fail(ctx, q, "hipStreamQuery");
        }
    }
}

// Blocks until the launch numbered ctx->seq has published its winner and returns it.
int fetch_key(nmi_ctx *ctx, unsigned long long *key)
{
    if (ctx->posted) {
        // The last workgroup stores (key | parity << 63) to fine-grained pinned memory; poll that one word.
        unsigned long long word = 0;
        const int rc = wait_word(ctx, &ctx->mailbox->word, 1ull << 63, (unsigned long long)(ctx->seq & 1u) << 63, &word);
        if (rc == NMI_OK) {
            *key = word & 0x7FFFFFFFFFFFFFFFull;
            return NMI_OK;
        }
        if (rc != NMI_ERR_NOT_READY) return rc;  // stream drained without a post: fall through to the copy path
    }
    NMI_HIP_TRY(ctx, hipMemcpyAsync(ctx->h_key, ctx->d_keys + ctx->last_slot, sizeof(unsigned long long),
                                    hipMemcpyDeviceToHost, ctx->stream));
    NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *key = *ctx->h_key;
    return NMI_OK;
}

// Uploads `n` floats through a StagingRing slot on the context's stream; *d_out is the device copy (valid for work
// enqueued on that stream until the slot comes round again).
int stage_floats(nmi_ctx *ctx, StagingRing &ring, const float *h_src, size_t n, float **d_out)
{
    if (n > ring.cap) {
        NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < StagingRing::kSlots; ++i) {
            if (ring.d[i]) NMI_HIP_TRY(ctx, hipFree(ring.d[i]));
            if (ring.h[i]) NMI_HIP_TRY(ctx, hipHostFree(ring.h[i]));
            ring.d[i] = ring.h[i] = nullptr;
        }
        ring.cap = 0;
        for (int i = 0; i < StagingRing::kSlots; ++i) {
            NMI_HIP_TRY(ctx, hipMalloc((void **)&ring.d[i], n * sizeof(float)));
            NMI_HIP_TRY(ctx, hipHostMalloc((void **)&ring.h[i], n * sizeof(float), hipHostMallocDefault));
            if (!ring.ev[i]) NMI_HIP_TRY(ctx, hipEventCreateWithFlags(&ring.ev[i], hipEventDisableTiming));
        }
        ring.cap = n;
    }
    const int slot = (int)(ring.uses++ % StagingRing::kSlots);
    NMI_HIP_TRY(ctx, hipEventSynchronize(ring.ev[slot]));
    memcpy(ring.h[slot], h_src, n * sizeof(float));
    NMI_HIP_TRY(ctx, hipMemcpyAsync(ring.d[slot], ring.h[slot], n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    NMI_HIP_TRY(ctx, hipEventRecord(ring.ev[slot], ctx->stream));
    *d_out = ring.d[slot];
    return NMI_OK;
}

int check_grid_args(nmi_ctx *ctx, const uint8_t *render_stack, int S_local, int s_offset, int S_total,
                    const uint8_t *warp_stack, int Wn)
{
    if (!ctx) return NMI_ERR_INVALID_ARGUMENT;
    ctx->detail.clear();
    if (S_local < 0 || Wn < 0 || s_offset < 0 || S_total < S_local || s_offset + S_local > S_total)
        return NMI_ERR_INVALID_ARGUMENT;
    if ((S_local > 0 && !render_stack) || (Wn > 0 && !warp_stack)) return NMI_ERR_INVALID_ARGUMENT;
    if ((int64_t)S_total * Wn >= 0x7FFFFFFFll) return NMI_ERR_UNSUPPORTED;  // index lives in 32 bits of the key
    return NMI_OK;
}

}  // namespace nmi_internal

using namespace nmi_internal;

extern "C" {

int nmi_abi_version(void) { return NMI_HIP_ABI_VERSION; }

const char *nmi_error_string(int code)
{
    if (code == NMI_OK) return "ok";
    if (code == NMI_ERR_INVALID_ARGUMENT) return "invalid argument";
    if (code == NMI_ERR_UNSUPPORTED) return "unsupported configuration";
    if (code == NMI_ERR_NO_DEVICE) return "no usable HIP device";
    if (code == NMI_ERR_NOT_READY) return "not ready";
    if (code <= NMI_ERR_RCCL) return "RCCL error";
    if (code <= NMI_ERR_HIP) return hipGetErrorString((hipError_t)(NMI_ERR_HIP - code));
    return "unknown error";
}

const char *nmi_last_error_detail(nmi_ctx *ctx) { return ctx ? ctx->detail.c_str() : ""; }

int nmi_params_default(nmi_params *p, int32_t width, int32_t height)
{
    if (!p) return NMI_ERR_INVALID_ARGUMENT;
    memset(p, 0, sizeof *p);
    p->width = width;
    p->height = height;
    p->bins = 256;             // HISTOGRAM256_BIN_COUNT, NMI.cuh:39
    p->mode = NMI_MODE_SUC;    // kernel.cuh:22-23 + NMI.cu:344,352
    p->use_bg = 1;             // nmi_prop_BG true, allProperties.hpp:38
    p->render_bottom_up = 1;   // NMI.cu:82
    p->device = -1;
    p->max_candidates = 0;
    p->stream = nullptr;
    return NMI_OK;
}

int nmi_create(const nmi_params *params, nmi_ctx **out_ctx)
{
    if (!params || !out_ctx) return NMI_ERR_INVALID_ARGUMENT;
    *out_ctx = nullptr;
    const nmi_params &p = *params;
    if (p.width <= 0 || p.height <= 0) return NMI_ERR_INVALID_ARGUMENT;
    if ((int64_t)p.width * p.height > (1ll << 24)) return NMI_ERR_UNSUPPORTED;  // counts must stay exact in fp32
    if (p.mode != NMI_MODE_SUC && p.mode != NMI_MODE_ENMI) return NMI_ERR_INVALID_ARGUMENT;
    int shift = -1;
    for (int k = 0; k <= 4; ++k)
        if (p.bins == (256 >> k)) shift = k;
    if (shift < 0) return NMI_ERR_UNSUPPORTED;
    for (int r : p.reserved)
        if (r != 0) return NMI_ERR_INVALID_ARGUMENT;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return NMI_ERR_NO_DEVICE;
    int dev = p.device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return NMI_ERR_NO_DEVICE;
    if (dev >= ndev) return NMI_ERR_INVALID_ARGUMENT;

    nmi_ctx *ctx = new (std::nothrow) nmi_ctx;
    if (!ctx) return NMI_ERR_INVALID_ARGUMENT;
    ctx->params = p;
    ctx->device = dev;
    ctx->npix = p.width * p.height;
    ctx->shift = shift;
    DeviceGuard guard(dev);

    int rc = NMI_OK;
    auto fail = [&](hipError_t e, const char *what) {
        rc = hip_fail(ctx, e, what);
        fprintf(stderr, "nmi_create: %s\n", ctx->detail.c_str());
        nmi_destroy(ctx);
        return rc;
    };
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return fail(e, "hipGetDeviceProperties");
    ctx->compute_units = prop.multiProcessorCount;
    if ((size_t)nmi::grid_kernel_lds_bytes() > prop.sharedMemPerBlock) {
        // The kernel keeps a whole packed joint histogram in LDS: it needs a CU with >= 136 KiB (gfx950: 160 KiB).
        nmi_destroy(ctx);
        return NMI_ERR_UNSUPPORTED;
    }
    if ((e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess)
        return fail(e, "hipStreamCreateWithFlags");
    ctx->stream = p.stream ? (hipStream_t)p.stream : ctx->own_stream;
    if ((e = hipMalloc((void **)&ctx->table, ((size_t)ctx->npix + 1) * sizeof(float))) != hipSuccess)
        return fail(e, "hipMalloc(table)");
    if ((e = hipMalloc((void **)&ctx->d_keys, 2 * sizeof(unsigned long long))) != hipSuccess) return fail(e, "hipMalloc(key)");
    if ((e = hipMalloc((void **)&ctx->d_done, sizeof(unsigned int))) != hipSuccess) return fail(e, "hipMalloc(done)");
    // on the context's own (non-blocking) stream: a legacy-stream hipMemset is not ordered against it
    if ((e = hipMemsetAsync(ctx->d_keys, 0, 2 * sizeof(unsigned long long), ctx->stream)) != hipSuccess)
        return fail(e, "hipMemsetAsync(key)");
    if ((e = hipMemsetAsync(ctx->d_done, 0, sizeof(unsigned int), ctx->stream)) != hipSuccess)
        return fail(e, "hipMemsetAsync(done)");
    if ((e = hipHostMalloc((void **)&ctx->mailbox, sizeof(nmi::Mailbox), hipHostMallocCoherent | hipHostMallocMapped)) !=
        hipSuccess)
        return fail(e, "hipHostMalloc(mailbox)");
    memset(ctx->mailbox, 0, sizeof(nmi::Mailbox));
    if ((e = hipMalloc((void **)&ctx->d_pair_rating, sizeof(float))) != hipSuccess) return fail(e, "hipMalloc(rating)");
    if ((e = hipMalloc((void **)&ctx->d_reduced_key, sizeof(unsigned long long))) != hipSuccess) return fail(e, "hipMalloc(reduced key)");
    if ((e = hipHostMalloc((void **)&ctx->score_mailbox, 2 * sizeof(unsigned long long), hipHostMallocCoherent | hipHostMallocMapped)) !=
        hipSuccess)
        return fail(e, "hipHostMalloc(score mailbox)");
    memset(ctx->score_mailbox, 0, 2 * sizeof(unsigned long long));
    if ((e = hipHostMalloc((void **)&ctx->h_split_error, nmi::kSplitRing * sizeof(uint32_t), hipHostMallocCoherent | hipHostMallocMapped)) != hipSuccess)
        return fail(e, "hipHostMalloc(split error)");
    memset(ctx->h_split_error, 0, nmi::kSplitRing * sizeof(uint32_t));
    if ((e = hipHostGetDevicePointer((void **)&ctx->d_split_error, ctx->h_split_error, 0)) != hipSuccess)
        return fail(e, "hipHostGetDevicePointer(split error)");
    if ((e = hipHostMalloc((void **)&ctx->h_key, sizeof(unsigned long long), hipHostMallocDefault)) != hipSuccess)
        return fail(e, "hipHostMalloc(key)");
    if ((e = hipHostMalloc((void **)&ctx->level_post, 64, hipHostMallocCoherent | hipHostMallocMapped)) != hipSuccess)
        return fail(e, "hipHostMalloc(level post)");
    *ctx->level_post = 0;
    if ((e = hipMalloc((void **)&ctx->d_plan, sizeof(nmi::LevelPlan))) != hipSuccess) return fail(e, "hipMalloc(level plan)");
    if ((e = hipMemsetAsync(ctx->d_plan, 0, sizeof(nmi::LevelPlan), ctx->stream)) != hipSuccess) return fail(e, "hipMemsetAsync(level plan)");
    {
        // where the general kernel's last workgroup posts what its search found (every search is a content probe)
        unsigned long long *d_post = nullptr;
        if ((e = hipHostGetDevicePointer((void **)&d_post, ctx->level_post, 0)) != hipSuccess) return fail(e, "hipHostGetDevicePointer(level post)");
        if ((e = hipMemcpyAsync(&ctx->d_plan->seen_post, &d_post, sizeof d_post, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
            return fail(e, "hipMemcpyAsync(level post)");
        const uint32_t max_joint = (uint32_t)ctx->fewlevels_bins;
        if ((e = hipMemcpyAsync(&ctx->d_plan->seen_max_joint, &max_joint, sizeof max_joint, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
            return fail(e, "hipMemcpyAsync(level plan)");
        if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return fail(e, "hipStreamSynchronize");  // (d_post is a local)
    }
    if ((e = hipEventCreate(&ctx->ev_start)) != hipSuccess) return fail(e, "hipEventCreate");
    if ((e = hipEventCreate(&ctx->ev_stop)) != hipSuccess) return fail(e, "hipEventCreate");
    if ((e = nmi::launch_table(ctx->table, ctx->npix, ctx->stream)) != hipSuccess) return fail(e, "launch_table");
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return fail(e, "hipStreamSynchronize");
    *out_ctx = ctx;
    return NMI_OK;
}

int nmi_destroy(nmi_ctx *ctx)
{
    if (!ctx) return NMI_OK;
    DeviceGuard guard(ctx->device);
    if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
    if (ctx->table) (void)hipFree(ctx->table);
    if (ctx->d_reduced_key) (void)hipFree(ctx->d_reduced_key);
    if (ctx->score_mailbox) (void)hipHostFree(ctx->score_mailbox);
    if (ctx->d_slabs) (void)hipFree(ctx->d_slabs);
    if (ctx->h_pair_table) (void)hipHostFree(ctx->h_pair_table);
    if (ctx->d_pair_scores) (void)hipFree(ctx->d_pair_scores);
    if (ctx->d_blocks) (void)hipFree(ctx->d_blocks);
    if (ctx->d_pix_timeouts) (void)hipFree(ctx->d_pix_timeouts);
    if (ctx->d_pix_blocks) (void)hipFree(ctx->d_pix_blocks);
    if (ctx->h_split_error) (void)hipHostFree(ctx->h_split_error);
    if (ctx->level_post) (void)hipHostFree(ctx->level_post);
    if (ctx->d_plan) (void)hipFree(ctx->d_plan);
    if (ctx->d_rank_stacks) (void)hipFree(ctx->d_rank_stacks);
    if (ctx->d_keys) (void)hipFree(ctx->d_keys);
    if (ctx->d_done) (void)hipFree(ctx->d_done);
    if (ctx->mailbox) (void)hipHostFree(ctx->mailbox);
    if (ctx->d_pair_rating) (void)hipFree(ctx->d_pair_rating);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->d_zbuf) (void)hipFree(ctx->d_zbuf);
    mesh_work_free(&ctx->mesh);
    for (int i = 0; i < StagingRing::kSlots; ++i) {
        if (ctx->mvp_ring.d[i]) (void)hipFree(ctx->mvp_ring.d[i]);
        if (ctx->mvp_ring.h[i]) (void)hipHostFree(ctx->mvp_ring.h[i]);
        if (ctx->mvp_ring.ev[i]) (void)hipEventDestroy(ctx->mvp_ring.ev[i]);
    }
    for (auto &oe : ctx->orders) {
        if (oe.d) (void)hipFree(oe.d);
        if (oe.h) (void)hipHostFree(oe.h);
    }
    for (int i = 0; i < nmi_ctx::kWarpRing; ++i) {
        if (ctx->d_warp_coeffs[i]) (void)hipFree(ctx->d_warp_coeffs[i]);
        if (ctx->h_warp_coeffs[i]) (void)hipHostFree(ctx->h_warp_coeffs[i]);
        if (ctx->warp_ev[i]) (void)hipEventDestroy(ctx->warp_ev[i]);
    }
    if (ctx->h_key) (void)hipHostFree(ctx->h_key);
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return NMI_OK;
}

int nmi_set_stream(nmi_ctx *ctx, void *stream)
{
    if (!ctx) return NMI_ERR_INVALID_ARGUMENT;
    ctx->stream = stream ? (hipStream_t)stream : ctx->own_stream;
    return NMI_OK;
}

int nmi_synchronize(nmi_ctx *ctx)
{
    if (!ctx) return NMI_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(ctx->device);
    NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return NMI_OK;
}

int nmi_set_option(nmi_ctx *ctx, int32_t option, int64_t value)
{
    if (!ctx) return NMI_ERR_INVALID_ARGUMENT;
    switch (option) {
    case NMI_OPT_HIST_VARIANT:
        if (value < 0 || value > 4) return NMI_ERR_INVALID_ARGUMENT;
        if (value != 1 && value != 3 && !nmi::ablation_variants_built()) return NMI_ERR_UNSUPPORTED;  // -DNMI_BUILD_ABLATIONS
        ctx->hist_variant = (int)value;
        return NMI_OK;
    case NMI_OPT_SPLIT:
        if (value != -1 && value != 0 && value != 1 && value != 2 && value != 4 && value != 8) return NMI_ERR_INVALID_ARGUMENT;
        ctx->split_mode = (int)value;
        return NMI_OK;
    case NMI_OPT_SPLIT_PIXELS:
        if (value != -1 && (value < 1 || value > 8)) return NMI_ERR_INVALID_ARGUMENT;  // 3, 5 ... 8: with NMI_OPT_SPLIT 1 only
        ctx->split_pixels = (int)value;
        return NMI_OK;
    case NMI_OPT_PIX_OWNER_BIAS:
        if (value < 0 || value > (1 << 24)) return NMI_ERR_INVALID_ARGUMENT;
        ctx->pix_owner_bias = (int)value;
        return NMI_OK;
    case NMI_OPT_STAMPS:
        ctx->dbg_stamps = (unsigned long long *)(uintptr_t)value;
        return NMI_OK;
    case NMI_OPT_WAIT_MODE:
        if (value < 0 || value > 1) return NMI_ERR_INVALID_ARGUMENT;
        ctx->wait_mode = (int)value;
        return NMI_OK;
    case NMI_OPT_PHASE_MASK:
        if (value < 0 || value > 1023) return NMI_ERR_INVALID_ARGUMENT;
        ctx->phase_mask = (int)value;
        return NMI_OK;
    case NMI_OPT_XCD_TILING:
        if (value < 0 || value > 1) return NMI_ERR_INVALID_ARGUMENT;
        ctx->xcd_tiling = (int)value;
        return NMI_OK;
    case NMI_OPT_RESULT_PATH:
        if (value < 0 || value > 1) return NMI_ERR_INVALID_ARGUMENT;
        ctx->result_path = (int)value;
        return NMI_OK;
    case NMI_OPT_TILE_QUEUE:
        if (value < 0 || value > (4ll << 20)) return NMI_ERR_INVALID_ARGUMENT;
        ctx->tile_queue_limit = (unsigned long long)value;
        return NMI_OK;
    case NMI_OPT_CLIP_QUEUE:
        if (value < 0) return NMI_ERR_INVALID_ARGUMENT;
        ctx->clip_queue_limit = (unsigned long long)value;
        return NMI_OK;
    case NMI_OPT_CONTENT_PATH:
        if (value < -1 || value > 1) return NMI_ERR_INVALID_ARGUMENT;
        ctx->content_path = (int)value;
        ctx->few_hint = false;
        {   // the device posts changes of its verdict only: start it from "not few" as well (blocking: a local is copied)
            DeviceGuard guard(ctx->device);
            const uint32_t zero = 0;
            NMI_HIP_TRY(ctx, hipMemcpyAsync(&ctx->d_plan->seen_state, &zero, sizeof zero, hipMemcpyHostToDevice, ctx->stream));
            NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        }
        return NMI_OK;
    case NMI_OPT_FEWLEVELS_BINS:
        if (value < 1 || value > nmi::fewlevels_max_joint()) return NMI_ERR_INVALID_ARGUMENT;
        ctx->fewlevels_bins = (int)value;
        {
            DeviceGuard guard(ctx->device);
            const uint32_t max_joint = (uint32_t)value, zero = 0;
            NMI_HIP_TRY(ctx, hipMemcpyAsync(&ctx->d_plan->seen_max_joint, &max_joint, sizeof max_joint, hipMemcpyHostToDevice, ctx->stream));
            NMI_HIP_TRY(ctx, hipMemcpyAsync(&ctx->d_plan->seen_state, &zero, sizeof zero, hipMemcpyHostToDevice, ctx->stream));
            NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            ctx->few_hint = false;
        }
        return NMI_OK;
    case NMI_OPT_WORKGROUPS:
        if (value < 0 || value > (1 << 20)) return NMI_ERR_INVALID_ARGUMENT;
        ctx->workgroups = (int)value;
        return NMI_OK;
    default:
        return NMI_ERR_INVALID_ARGUMENT;
    }
}

int nmi_set_profiling(nmi_ctx *ctx, int32_t enabled)
{
    if (!ctx) return NMI_ERR_INVALID_ARGUMENT;
    ctx->profiling = enabled != 0;
    ctx->have_timing = false;
    return NMI_OK;
}

int nmi_last_kernel_ms(nmi_ctx *ctx, float *h_ms)
{
    if (!ctx || !h_ms) return NMI_ERR_INVALID_ARGUMENT;
    if (!ctx->have_timing) return NMI_ERR_NOT_READY;
    DeviceGuard guard(ctx->device);
    NMI_HIP_TRY(ctx, hipEventSynchronize(ctx->ev_stop));
    NMI_HIP_TRY(ctx, hipEventElapsedTime(h_ms, ctx->ev_start, ctx->ev_stop));
    return NMI_OK;
}

int nmi_get_info(nmi_ctx *ctx, int32_t *compute_units, int32_t *workgroups_per_launch, int32_t *lds_bytes)
{
    if (!ctx) return NMI_ERR_INVALID_ARGUMENT;
    if (compute_units) *compute_units = ctx->compute_units;
    if (workgroups_per_launch) *workgroups_per_launch = ctx->workgroups > 0 ? ctx->workgroups : ctx->compute_units;
    if (lds_bytes) *lds_bytes = nmi::grid_kernel_lds_bytes();
    return NMI_OK;
}

int nmi_last_content(nmi_ctx *ctx, int32_t *few_levels, int32_t *nr, int32_t *nw)
{
    if (!ctx) return NMI_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(ctx->device);
    NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const unsigned long long posted = __atomic_load_n(ctx->level_post, __ATOMIC_ACQUIRE);
    const int32_t r = (int32_t)((posted >> 16) & 0xFFFFu), w = (int32_t)(posted & 0xFFFFu);
    if (nr) *nr = r;
    if (nw) *nw = w;
    // a few-levels launch carries its own probe, so the post is about that very search
    if (few_levels) *few_levels = (ctx->last_few && r > 0 && w > 0 && (int64_t)r * w <= ctx->fewlevels_bins) ? 1 : 0;
    return NMI_OK;
}

int nmi_copy_term_table(nmi_ctx *ctx, float *h_out, int64_t n)
{
    if (!ctx || !h_out || n != (int64_t)ctx->npix + 1) return NMI_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(ctx->device);
    NMI_HIP_TRY(ctx, hipMemcpyAsync(h_out, ctx->table, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return NMI_OK;
}

uint64_t nmi_key_pack(float score, int64_t global_linear_index)
{
    if (!(score >= 0.0f) || global_linear_index < 0 || global_linear_index >= 0xFFFFFFFFll) return 0;
    uint32_t bits = 0;
    if (score != 0.0f) memcpy(&bits, &score, sizeof bits);
    return ((uint64_t)bits << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)global_linear_index);
}

int nmi_key_unpack(uint64_t key, int64_t *global_linear_index, float *score)
{
    if (key == 0) {
        if (global_linear_index) *global_linear_index = -1;
        if (score) *score = 0.0f;
        return NMI_OK;
    }
    const uint32_t bits = (uint32_t)(key >> 32);
    if (global_linear_index) *global_linear_index = (int64_t)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFu));
    if (score) memcpy(score, &bits, sizeof bits);
    return NMI_OK;
}

int nmi_search_grid_shard(nmi_ctx *ctx, const uint8_t *render_stack, int32_t S_local, int32_t s_offset, int32_t S_total,
                          const uint8_t *warp_stack, int32_t Wn, float *d_ratings, uint64_t *d_key, uint64_t *h_key)
{
    return nmi_search_grid_block(ctx, render_stack, S_local, s_offset, S_total, warp_stack, Wn, 0, Wn, d_ratings, d_key, h_key);
}

int nmi_search_grid_block(nmi_ctx *ctx, const uint8_t *render_stack, int32_t S_local, int32_t s_offset, int32_t S_total,
                          const uint8_t *warp_stack, int32_t Wn_local, int32_t w_offset, int32_t Wn_total, float *d_ratings,
                          uint64_t *d_key, uint64_t *h_key)
{
    return search_block(ctx, render_stack, S_local, s_offset, S_total, warp_stack, Wn_local, w_offset, Wn_total, d_ratings, d_key, h_key,
                        /*caller_checks=*/false);
}

}  // extern "C"

int nmi_internal::search_block(nmi_ctx *ctx, const uint8_t *render_stack, int32_t S_local, int32_t s_offset, int32_t S_total,
                               const uint8_t *warp_stack, int32_t Wn_local, int32_t w_offset, int32_t Wn_total, float *d_ratings,
                               uint64_t *d_key, uint64_t *h_key, bool caller_checks)
{
    int rc = check_grid_args(ctx, render_stack, S_local, s_offset, S_total, warp_stack, Wn_local);
    if (rc != NMI_OK) return rc;
    if (w_offset < 0 || Wn_total < Wn_local || w_offset + Wn_local > Wn_total) return NMI_ERR_INVALID_ARGUMENT;
    if ((int64_t)S_total * Wn_total >= 0x7FFFFFFFll) return NMI_ERR_UNSUPPORTED;  // index lives in 32 bits of the key
    const int32_t Wn = Wn_local;
    DeviceGuard guard(ctx->device);
    ctx->allow_unchecked_split = caller_checks;
    rc = enqueue_grid(ctx, render_stack, S_local, s_offset, S_total, warp_stack, Wn, d_ratings, (unsigned long long *)d_key,
                      h_key != nullptr, nullptr, nullptr, nullptr, nullptr, w_offset);
    ctx->allow_unchecked_split = false;
    if (rc != NMI_OK) return rc;
    if (h_key) {
        unsigned long long k = 0;
        rc = fetch_key(ctx, &k);
        if (rc != NMI_OK) return rc;
        if (split_timed_out(ctx)) {  // (the cooldown now routes the redo through nmi_grid_kernel)
            rc = search_block(ctx, render_stack, S_local, s_offset, S_total, warp_stack, Wn_local, w_offset, Wn_total, d_ratings, d_key,
                              h_key, false);
            if (rc == NMI_OK) ctx->detail = kSplitTimeoutNote;
            return rc;
        }
        *h_key = k;
        // The winner is posted by the last workgroup before the kernel has retired: a caller that asked for the rating
        // table may read it right after this call, from any stream, so the table must be complete (and written back).
        if (d_ratings || d_key) NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return NMI_OK;
}

extern "C" {

int nmi_split_status(nmi_ctx *ctx, int32_t *timeouts, int32_t *cooldown_calls_left, int32_t *next_cooldown, int32_t *last_launch_parts)
{
    if (!ctx) return NMI_ERR_INVALID_ARGUMENT;
    if (timeouts) *timeouts = (int32_t)ctx->split_timeouts;
    if (cooldown_calls_left) *cooldown_calls_left = (int32_t)ctx->split_cooldown;
    if (next_cooldown) *next_cooldown = (int32_t)ctx->split_backoff;
    if (last_launch_parts) *last_launch_parts = ctx->last_parts;
    return NMI_OK;
}

int nmi_pix_status(nmi_ctx *ctx, int32_t *last_launch_ranges, int32_t *healed)
{
    if (!ctx) return NMI_ERR_INVALID_ARGUMENT;
    if (last_launch_ranges) *last_launch_ranges = ctx->last_pix;
    if (healed) {
        *healed = 0;
        if (ctx->d_pix_timeouts) {
            DeviceGuard guard(ctx->device);
            uint32_t n = 0;
            NMI_HIP_TRY(ctx, hipMemcpyAsync(&n, ctx->d_pix_timeouts, sizeof n, hipMemcpyDeviceToHost, ctx->stream));
            NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            *healed = (int32_t)n;
        }
    }
    return NMI_OK;
}

int nmi_search_grid(nmi_ctx *ctx, const uint8_t *render_stack, int32_t S, const uint8_t *warp_stack, int32_t Wn,
                    float *d_ratings, int64_t *h_best_index, float *h_best_score)
{
    uint64_t key = 0;
    int rc = nmi_search_grid_shard(ctx, render_stack, S, 0, S, warp_stack, Wn, d_ratings, nullptr, &key);
    if (rc != NMI_OK) return rc;
    return nmi_key_unpack(key, h_best_index, h_best_score);
}

int nmi_eval_pair_debug(nmi_ctx *ctx, const uint8_t *render, const uint8_t *warped, float *h_score, uint32_t *d_joint,
                        uint32_t *d_hist_render, uint32_t *d_hist_warped, float *d_sums)
{
    if (!ctx || !render || !warped || !h_score) return NMI_ERR_INVALID_ARGUMENT;
    ctx->detail.clear();
    DeviceGuard guard(ctx->device);
    const bool dbg = d_joint || d_hist_render || d_hist_warped || d_sums;
    int rc = enqueue_grid(ctx, render, 1, 0, 1, warped, 1, ctx->d_pair_rating, nullptr, false, d_joint, d_hist_render,
                          d_hist_warped, d_sums, 0, /*post_score=*/true);
    if (rc != NMI_OK) return rc;
    if (ctx->result_path == 1) {
        // kernel.cu:100 copies the score back with a blocking cudaMemcpy; here the scoring lane stores
        // (score bits | call number << 32) into pinned host memory and the call polls that word (~10 us less per call).
        unsigned long long word = 0;
        rc = wait_word(ctx, ctx->score_mailbox, 0xFFFFFFFF00000000ull, (unsigned long long)ctx->pair_seq << 32, &word);
        if (rc == NMI_OK && split_timed_out(ctx)) {
            rc = nmi_eval_pair_debug(ctx, render, warped, h_score, d_joint, d_hist_render, d_hist_warped, d_sums);
            if (rc == NMI_OK) ctx->detail = kSplitTimeoutNote;
            return rc;
        }
        if (rc == NMI_OK) {
            const uint32_t bits = (uint32_t)word;
            memcpy(h_score, &bits, sizeof bits);
            if (dbg) NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // the exported histograms must be complete
            return NMI_OK;
        }
        if (rc != NMI_ERR_NOT_READY) return rc;
    }
    NMI_HIP_TRY(ctx, hipMemcpyAsync(h_score, ctx->d_pair_rating, sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (split_timed_out(ctx)) {
        rc = nmi_eval_pair_debug(ctx, render, warped, h_score, d_joint, d_hist_render, d_hist_warped, d_sums);
        if (rc == NMI_OK) ctx->detail = kSplitTimeoutNote;
        return rc;
    }
    return NMI_OK;
}

int nmi_eval_pairs(nmi_ctx *ctx, const uint8_t *const *h_renders, const uint8_t *const *h_warps, int32_t n, float *h_scores)
{
    if (!ctx || n < 0 || (n > 0 && (!h_renders || !h_warps || !h_scores))) return NMI_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < n; ++i)
        if (!h_renders[i] || !h_warps[i]) return NMI_ERR_INVALID_ARGUMENT;
    ctx->detail.clear();
    if (n == 0) return NMI_OK;
    DeviceGuard guard(ctx->device);
    // launches of at most compute_units / 4 pairs (4 row parts each); small batches get more parts per pair
    const int min_parts = ctx->split_mode > 1 ? ctx->split_mode : 4;
    const int cus = ctx->workgroups > 0 && ctx->workgroups < ctx->compute_units ? ctx->workgroups : ctx->compute_units;
    const int per_launch = cus / min_parts > 0 ? ((cus / min_parts) & ~7) > 0 ? (cus / min_parts) & ~7 : 1 : 1;
    // Decided BEFORE anything is launched: does a split form exist for every chunk of this batch (none does with
    // NMI_OPT_WORKGROUPS below 16, with NMI_OPT_SPLIT 0, or while the split forms are paused after a timeout)?
    if ((int64_t)n > (int64_t)per_launch * 128) {  // the timeout ring answers for at most kSplitRing launches in flight
        for (int off = 0; off < n; off += per_launch * 128) {
            const int m = n - off < per_launch * 128 ? n - off : per_launch * 128;
            const int rc = nmi_eval_pairs(ctx, h_renders + off, h_warps + off, m, h_scores + off);
            if (rc != NMI_OK) return rc;
        }
        return NMI_OK;
    }
    bool split_ok = ctx->hist_variant == 3 && ctx->split_mode != 0 && ctx->split_mode != 1 && ctx->split_cooldown == 0;
    for (int off = 0; split_ok && off < n; off += per_launch) {
        int parts = 0, pix = 1;
        choose_split(ctx, n - off < per_launch ? n - off : per_launch, ctx->workgroups > 0 ? ctx->workgroups : ctx->compute_units, &parts, &pix);
        split_ok = parts != 0;
    }
    if (!split_ok) {  // one pair at a time (nmi_eval_pair consumes the cooldown)
        for (int i = 0; i < n; ++i) {
            const int rc = nmi_eval_pair(ctx, h_renders[i], h_warps[i], &h_scores[i]);
            if (rc != NMI_OK) return rc;
        }
        return NMI_OK;
    }
    // pointer tables (pinned, device-mapped) and the scores (device), grown on demand
    if (n > ctx->pairs_cap) {
        NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->h_pair_table) NMI_HIP_TRY(ctx, hipHostFree(ctx->h_pair_table));
        if (ctx->d_pair_scores) NMI_HIP_TRY(ctx, hipFree(ctx->d_pair_scores));
        ctx->h_pair_table = nullptr;
        ctx->d_pair_scores = nullptr;
        ctx->pairs_cap = 0;
        const int cap = n > 256 ? n : 256;
        NMI_HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_pair_table, (size_t)cap * 2 * sizeof(void *), hipHostMallocMapped | hipHostMallocCoherent));
        NMI_HIP_TRY(ctx, hipHostGetDevicePointer((void **)&ctx->d_pair_table, ctx->h_pair_table, 0));
        NMI_HIP_TRY(ctx, hipMalloc((void **)&ctx->d_pair_scores, (size_t)cap * sizeof(float)));
        ctx->pairs_cap = cap;
    } else {
        NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // an earlier batch may still be reading the tables
    }
    for (int i = 0; i < n; ++i) {
        ctx->h_pair_table[i] = h_renders[i];
        ctx->h_pair_table[ctx->pairs_cap + i] = h_warps[i];
    }
    uint32_t first_epoch = 0, n_launches = 0;
    for (int off = 0; off < n; off += per_launch) {
        const int m = n - off < per_launch ? n - off : per_launch;
        ctx->pair_renders = ctx->d_pair_table + off;
        ctx->pair_warps = ctx->d_pair_table + ctx->pairs_cap + off;
        ctx->pair_renders_host = h_renders + off;
        ctx->pair_warps_host = h_warps + off;
        const int rc = enqueue_grid(ctx, h_renders[off], m, 0, m, h_warps[off], 1, ctx->d_pair_scores + off, nullptr, false, nullptr,
                                    nullptr, nullptr, nullptr);
        ctx->pair_renders = ctx->pair_warps = nullptr;
        if (rc != NMI_OK) return rc;  // (NMI_ERR_UNSUPPORTED, before any launch, had no split form fitted after all)
        if (!n_launches++) first_epoch = ctx->last_epoch;
    }
    NMI_HIP_TRY(ctx, hipMemcpyAsync(h_scores, ctx->d_pair_scores, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    bool failed = false;  // every launch of the batch answers for itself (epochs are consecutive, apart from the skipped multiples of 65536)
    for (uint32_t e = first_epoch; n_launches && e - first_epoch <= ctx->last_epoch - first_epoch; ++e)
        failed = split_launch_failed(ctx, ctx->last_parts, e) || failed;
    if (failed) {  // now one pair at a time, through nmi_grid_kernel
        const int rc = nmi_eval_pairs(ctx, h_renders, h_warps, n, h_scores);
        if (rc == NMI_OK) ctx->detail = kSplitTimeoutNote;
        return rc;
    }
    return NMI_OK;
}

int nmi_eval_pair(nmi_ctx *ctx, const uint8_t *render, const uint8_t *warped, float *h_score)
{
    return nmi_eval_pair_debug(ctx, render, warped, h_score, nullptr, nullptr, nullptr, nullptr);
}

}  // extern "C"
