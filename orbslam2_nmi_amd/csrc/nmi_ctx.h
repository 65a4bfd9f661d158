// nmi_ctx.h -- internal: the context object behind include/nmi_hip.h and the helpers its translation units share
// (nmi_capi.cpp: context + search; nmi_capi_producers.cpp: warp / render producers; nmi_capi_pipeline.cpp: captured level
// and streaming pipeline; nmi_capi_rccl.cpp: the RCCL entry points).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <new>
#include <string>
#include <vector>

#include "nmi_hip.h"
#include "nmi_kernels.h"

// Small host->device parameter uploads (warp coefficients, view matrices) go through a ring of pinned staging buffers so
// that back-to-back submissions never have to wait for the stream: entry i is reused only after the copy that read it.
struct StagingRing {
    static constexpr int kSlots = 4;
    float *h[kSlots] = {};
    float *d[kSlots] = {};
    hipEvent_t ev[kSlots] = {};
    size_t cap = 0;  // floats per slot
    unsigned uses = 0;
};

struct nmi_ctx {
    nmi_params params{};
    int device = 0;
    int compute_units = 0;
    int npix = 0;
    int shift = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    float *table = nullptr;             // [npix + 1]
    unsigned long long *d_keys = nullptr;  // two device slots for the packed winner, used alternately (ping-pong)
    unsigned long long *h_key = nullptr;   // pinned host mirror (copy path)
    unsigned int *d_done = nullptr;        // finished-workgroup counter
    nmi::Mailbox *mailbox = nullptr;       // pinned, fine-grained: the kernel posts the winner here
    unsigned int seq = 0;                  // launches that post to the mailbox so far (blocking calls only)
    int slot = 0;                          // key slot of the next launch
    int last_slot = 0;                     // key slot of the most recent launch
    int last_parts = 0;                    // row parts per candidate of the most recent launch (0 = not the row-split kernel)
    int pix_owner_bias = 49152;            // NMI_OPT_PIX_OWNER_BIAS: pixels the owner of a candidate adds beyond an equal share
    int last_pix = 0;                      // pixel ranges per candidate when it was nmi_pix_kernel (0 = it was not)
    uint32_t *d_pix_timeouts = nullptr;    // candidates whose owner gave up on a helper and scored them alone (nmi_pix_kernel)
    uint32_t last_epoch = 0;               // its split epoch (meaningful when last_parts != 0)
    // Split-kernel liveness (nmi_split_kernel.hip: its consumers spin, so a launch whose workgroups are not all resident
    // times out after 2 ms).  A timeout is attributed to ITS launch (epoch), that search is redone by nmi_grid_kernel, and the
    // split forms stay off for split_cooldown further small-grid launches -- 16, doubling per consecutive timeout up to
    // 4096 -- after which they are tried again; a split launch that is checked and found good re-arms the short cooldown.
    bool allow_unchecked_split = false;    // this launch's caller checks for a timeout itself (blocking calls, stream, RCCL)
    uint32_t split_timeouts = 0;           // timeouts seen so far
    uint32_t split_cooldown = 0;           // small-grid launches still to go through nmi_grid_kernel
    uint32_t split_backoff = kSplitBackoffMin;  // cooldown the next timeout starts
    uint32_t split_cooldown_epoch = 0;     // newest split epoch issued when the last cooldown was set: only launches after it re-arm the short one
    static constexpr uint32_t kSplitBackoffMin = 16, kSplitBackoffMax = 4096;
    int result_path = 1;                   // 1 mailbox spin (default), 0 hipMemcpyAsync + stream sync
    bool posted = false;                   // the most recent launch posts to the mailbox
    float *d_pair_rating = nullptr;
    unsigned long long *d_reduced_key = nullptr;  // receive buffer of the RCCL all-reduce (never one of the ping-pong slots)
    unsigned long long *score_mailbox = nullptr;  // pinned, fine-grained: nmi_eval_pair's (score bits | sequence << 32)
    unsigned int pair_seq = 0;                    // nmi_eval_pair calls that posted so far
    int wait_mode = 0;                            // NMI_OPT_WAIT_MODE: 0 spin on the mailbox, 1 yield the core between polls
    // Visiting orders of the candidates (XCD-aware tiling), one per grid shape seen, so that a coarse-to-fine search
    // alternating between shapes (translation level 27 x 1, rotation level 1 x 27, ...) never waits for the stream.
    struct OrderEntry {
        int S = -1, Wn = -1;
        int *d = nullptr, *h = nullptr;
        int64_t cap = 0;
        uint64_t last_use = 0;
    };
    static constexpr int kOrderCache = 16;
    OrderEntry orders[kOrderCache];
    uint64_t order_clock = 0;
    int xcd_tiling = 1;                   // NMI_OPT_XCD_TILING
    nmi::SplitSlab *d_slabs = nullptr;    // hand-off slabs of the split kernel, one per candidate
    int slab_cap = 0;
    unsigned long long *d_blocks = nullptr;  // counter blocks (granules) of the split kernel's pixel parts
    uint32_t split_epoch = 0;             // tag of the latest split launch
    uint32_t *h_split_error = nullptr, *d_split_error = nullptr;  // pinned ring [nmi::kSplitRing]: word (epoch % kSplitRing) = epoch of a launch whose hand-off timed out
    size_t blocks_bytes = 0;
    unsigned long long *d_pix_blocks = nullptr;  // hand-off blocks of nmi_pix_kernel (mid-size grids)
    size_t pix_blocks_bytes = 0;
    // nmi_eval_pairs: pointer tables [2][pairs_cap] in pinned host memory (renders, then warps) + device scores
    const uint8_t **h_pair_table = nullptr, **d_pair_table = nullptr;
    float *d_pair_scores = nullptr;
    int pairs_cap = 0;
    const uint8_t *const *pair_renders = nullptr, *const *pair_warps = nullptr;            // device views for the launch being enqueued
    const uint8_t *const *pair_renders_host = nullptr, *const *pair_warps_host = nullptr;  // the caller's arrays (alignment check)
    int split_pixels = -1;                // NMI_OPT_SPLIT_PIXELS: -1 automatic, 1 / 2 / 4 (with NMI_OPT_SPLIT 1: 2 ... 8)
    // Few-levels path (nmi_fewlevels_kernel.hip): which kernels score a search is decided from what the last probe of
    // the stacks found, posted by the device to *level_post = probe number << 32 | nr << 16 | nw.
    nmi::LevelPlan *d_plan = nullptr;
    unsigned long long *level_post = nullptr;  // pinned, fine-grained
    uint32_t level_seq = 0;               // probes enqueued so far
    uint32_t level_seen = 0;              // number of the probe the hint below comes from
    bool few_hint = false;                // the last probe seen found nr * nw <= fewlevels_bins
    int content_path = -1;                // NMI_OPT_CONTENT_PATH: -1 automatic (hint), 0 nmi_grid_kernel only, 1 few-levels first
    int fewlevels_bins = 4096;            // NMI_OPT_FEWLEVELS_BINS: largest nr * nw sent down the few-levels path
    uint8_t *d_rank_stacks = nullptr;     // rank images of the search in flight: renders, then warps
    size_t rank_bytes = 0;
    int last_few = 0;                     // the most recent launch went down the few-levels path (it may have fallen back)
    unsigned long long *dbg_stamps = nullptr;  // NMI_OPT_STAMPS
    int split_mode = -1;                  // NMI_OPT_SPLIT: -1 automatic, 0 never, 2 / 4 / 8 row parts whenever the grid fits, 1: pixel ranges only
    uint32_t *d_zbuf = nullptr;           // depth|colour anchor buffers of the point-cloud renderer (padded, per view)
    int64_t zbuf_cap = 0;
    nmi::MeshWork mesh;                    // mesh renderer (nmi_render_mesh): bins, their state, key buffer, clip queue -- for mesh_views views
    int mesh_views = 0;
    unsigned long long tile_queue_limit = 4ull << 20;  // NMI_OPT_TILE_QUEUE: usable entries per bin (capped by the bins' size)
    unsigned long long clip_queue_limit = ~0ull;       // NMI_OPT_CLIP_QUEUE
    StagingRing mvp_ring;
    uint32_t *d_scratch = nullptr;        // drained-counter slabs of the pipelined kernel
    int scratch_workgroups = 0;
    // inverse homographies for the warp producer: a small ring of (pinned staging, device copy, "copy consumed" event)
    // so that back-to-back submissions never wait for the stream
    static constexpr int kWarpRing = 4;
    float *d_warp_coeffs[kWarpRing] = {};
    float *h_warp_coeffs[kWarpRing] = {};
    hipEvent_t warp_ev[kWarpRing] = {};
    int warp_coeffs_cap = 0;
    unsigned warp_uses = 0;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    int hist_variant = 3;
    int phase_mask = 3;
    int workgroups = 0;
    bool profiling = false;
    bool have_timing = false;
    std::string detail;
};

// nmi_texture (nmi_texture_create): mip chain of a mesh texture as per-level fp32 luma on the device.
struct nmi_texture {
    nmi_ctx *ctx = nullptr;
    float *d_luma = nullptr;
    int levels = 0;
    int w[16] = {}, h[16] = {};
    long long off[16] = {};
};

namespace nmi_internal {

// Device buffers of the mesh renderer for S views, allocated and brought to their clean state (the renderer keeps them clean).
int mesh_work_alloc(nmi_ctx *ctx, int S, nmi::MeshWork *w);
void mesh_work_free(nmi::MeshWork *w);
int ensure_mesh_work(nmi_ctx *ctx, int S);  // the context's own, grown on demand
int ensure_mesh_pairs(nmi_ctx *ctx, nmi::MeshWork *w, long long n_triangles);  // the pair list of the two-kernel binning pass
int level_enqueue(nmi_level *lv, const float *h_mvps, const double *h_forward, const unsigned long long **d_key);
nmi_ctx *level_ctx(nmi_level *lv);
// ncclAllReduce(ncclMax, ncclUint64) of one 8-byte key on the context's stream, out of place (nmi_capi_rccl.cpp)
int rccl_allreduce_key(nmi_ctx *ctx, const unsigned long long *d_send, unsigned long long *d_recv, void *nccl_comm);

int hip_fail(nmi_ctx *ctx, hipError_t e, const char *what);
void build_order(int S, int Wn, int *order);
int ensure_order(nmi_ctx *ctx, int S, int Wn, const int **d_order);
// nmi_pix_kernel (mid-size grids): pixel ranges per candidate for a launch of `total` candidates on `cap` workgroups (0: another
// kernel), the owner's share of the pixels, the next hand-off epoch, the context's counter of healed timeouts
int choose_pix(const nmi_ctx *ctx, const nmi::GridArgs &a, int64_t total, int cap);
double pix_owner_share(const nmi_ctx *ctx, int pix);
int next_split_epoch(nmi_ctx *ctx, uint32_t *epoch);
int ensure_pix_timeouts(nmi_ctx *ctx);
int enqueue_grid(nmi_ctx *ctx, const uint8_t *render_stack, int S_local, int s_offset, int S_total, const uint8_t *warp_stack, int Wn,
                 float *d_ratings, unsigned long long *out_key, bool post, uint32_t *dbg_joint, uint32_t *dbg_h1, uint32_t *dbg_h2,
                 float *dbg_sums, int w_offset = 0, bool post_score = false);
int wait_word(nmi_ctx *ctx, const volatile unsigned long long *word, unsigned long long mask, unsigned long long want,
              unsigned long long *out);
int stage_floats(nmi_ctx *ctx, StagingRing &ring, const float *h_src, size_t n, float **d_out);
int fetch_key(nmi_ctx *ctx, unsigned long long *key);
bool split_timed_out(nmi_ctx *ctx);                                     // ... the most recent launch
bool split_launch_failed(nmi_ctx *ctx, int parts, uint32_t epoch);      // ... the launch with this epoch (waits for the stream on a hit)
// nmi_search_grid_block without the argument checks; caller_checks: the caller looks for a split timeout itself, so small
// grids may use the split kernel although the call only enqueues (h_key == nullptr)
int search_block(nmi_ctx *ctx, const uint8_t *render_stack, int32_t S_local, int32_t s_offset, int32_t S_total, const uint8_t *warp_stack,
                 int32_t Wn_local, int32_t w_offset, int32_t Wn_total, float *d_ratings, uint64_t *d_key, uint64_t *h_key,
                 bool caller_checks);
int check_grid_args(nmi_ctx *ctx, const uint8_t *render_stack, int S_local, int s_offset, int S_total, const uint8_t *warp_stack,
                    int Wn);

#define NMI_HIP_TRY(ctx, call)                                  \
    do {                                                        \
        hipError_t e_ = (call);                                 \
        if (e_ != hipSuccess) return nmi_internal::hip_fail((ctx), e_, #call); \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool active = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) active = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (active) (void)hipSetDevice(prev);
    }
};

}  // namespace nmi_internal
