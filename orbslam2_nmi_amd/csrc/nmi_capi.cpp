// nmi_capi.cpp -- the C ABI declared in include/nmi_hip.h on top of the gfx950 kernels.
//
// Host orchestration that replaces CUDAF::NMIWithCuda_noMask (Thirdparty/CUDA_Functions/kernel.cu:49-114)
// and the candidate loop + arg-max of Tracking::RelocalizeWithNMI (src/Tracking.cc:1879-1905,1952):
// a persistent context owns every buffer, one launch scores a whole candidate grid, and the only
// host<->device traffic per search is one 8-byte key.
#include <dlfcn.h>
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
    float *ratings = nullptr;           // internal rating table
    int64_t ratings_cap = 0;
    unsigned long long *d_keys = nullptr;  // two device slots for the packed winner, used alternately (ping-pong)
    unsigned long long *h_key = nullptr;   // pinned host mirror (copy path)
    unsigned int *d_done = nullptr;        // finished-workgroup counter
    nmi::Mailbox *mailbox = nullptr;       // pinned, fine-grained: the kernel posts the winner here
    unsigned int seq = 0;                  // launches that post to the mailbox so far (blocking calls only)
    int slot = 0;                          // key slot of the next launch
    int last_slot = 0;                     // key slot of the most recent launch
    int result_path = 1;                   // 1 mailbox spin (default), 0 hipMemcpyAsync + stream sync
    bool posted = false;                   // the most recent launch posts to the mailbox
    float *d_pair_rating = nullptr;
    int *d_order = nullptr;               // visiting order of the candidates (XCD-aware tiling), cached per grid shape
    int *h_order = nullptr;
    int64_t order_cap = 0;
    int order_S = -1, order_Wn = -1;
    int xcd_tiling = 1;                   // NMI_OPT_XCD_TILING
    uint32_t *d_zbuf = nullptr;           // depth|colour anchor buffers of the point-cloud renderer (padded, per view)
    int64_t zbuf_cap = 0;
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

namespace {

int hip_fail(nmi_ctx *ctx, hipError_t e, const char *what)
{
    if (ctx) {
        char buf[256];
        snprintf(buf, sizeof buf, "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
        ctx->detail = buf;
    }
    return NMI_ERR_HIP - (int)e;
}

#define NMI_HIP_TRY(ctx, call)                                  \
    do {                                                        \
        hipError_t e_ = (call);                                 \
        if (e_ != hipSuccess) return hip_fail((ctx), e_, #call); \
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

int ensure_ratings(nmi_ctx *ctx, int64_t n)
{
    if (n <= ctx->ratings_cap) return NMI_OK;
    if (ctx->ratings) NMI_HIP_TRY(ctx, hipFree(ctx->ratings));
    ctx->ratings = nullptr;
    ctx->ratings_cap = 0;
    NMI_HIP_TRY(ctx, hipMalloc((void **)&ctx->ratings, (size_t)n * sizeof(float)));
    ctx->ratings_cap = n;
    return NMI_OK;
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

int ensure_order(nmi_ctx *ctx, int S, int Wn)
{
    if (!ctx->xcd_tiling) return NMI_OK;
    if (ctx->order_S == S && ctx->order_Wn == Wn) return NMI_OK;
    const int64_t total = (int64_t)S * Wn;
    if (total > ctx->order_cap) {
        NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_order) NMI_HIP_TRY(ctx, hipFree(ctx->d_order));
        if (ctx->h_order) NMI_HIP_TRY(ctx, hipHostFree(ctx->h_order));
        ctx->d_order = ctx->h_order = nullptr;
        ctx->order_cap = 0;
        NMI_HIP_TRY(ctx, hipMalloc((void **)&ctx->d_order, (size_t)total * sizeof(int)));
        NMI_HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_order, (size_t)total * sizeof(int), hipHostMallocDefault));
        ctx->order_cap = total;
    } else {
        NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // the staging copy may still feed an earlier upload
    }
    build_order(S, Wn, ctx->h_order);
    NMI_HIP_TRY(ctx, hipMemcpyAsync(ctx->d_order, ctx->h_order, (size_t)total * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    ctx->order_S = S;
    ctx->order_Wn = Wn;
    return NMI_OK;
}

// Enqueues the grid kernel (one launch, nothing else).  No synchronisation.
int enqueue_grid(nmi_ctx *ctx, const uint8_t *render_stack, int S_local, int s_offset, int S_total,
                 const uint8_t *warp_stack, int Wn, float *d_ratings, unsigned long long *out_key, bool post,
                 uint32_t *dbg_joint, uint32_t *dbg_h1, uint32_t *dbg_h2, float *dbg_sums)
{
    const nmi_params &p = ctx->params;
    nmi::GridArgs a{};
    a.render_stack = render_stack;
    a.warp_stack = warp_stack;
    a.S_local = S_local;
    a.Wn = Wn;
    a.s_offset = s_offset;
    a.S_total = S_total;
    a.width = p.width;
    a.height = p.height;
    a.npix = ctx->npix;
    a.vec_ok = (p.width % 16 == 0) && (((uintptr_t)render_stack | (uintptr_t)warp_stack) % 16 == 0);
    a.chunks_per_row = a.vec_ok ? p.width / 16 : 1;
    a.cpr_magic = a.chunks_per_row > 1 ? (uint32_t)((0x100000000ull + a.chunks_per_row - 1) / a.chunks_per_row) : 0u;
    a.shift = ctx->shift;
    a.mode = p.mode;
    a.flip = p.render_bottom_up ? 1 : 0;
    a.table = ctx->table;
    a.scratch = ctx->d_scratch;
    a.order = nullptr;
    if (ctx->xcd_tiling && (int64_t)S_local * Wn > 0 && (int64_t)S_local * Wn <= (1ll << 24)) {  // 4 B per candidate
        const int orc = ensure_order(ctx, S_local, Wn);
        if (orc != NMI_OK) return orc;
        a.order = ctx->d_order;
    }
    a.ratings = d_ratings;
    a.key = ctx->d_keys + ctx->slot;
    a.reset_key = ctx->d_keys + (ctx->slot ^ 1);
    a.out_key = out_key;
    a.done = ctx->d_done;
    // Only launches whose winner the host will poll for post to the mailbox (one bit of sequence is enough
    // because those calls are blocking, hence strictly alternating).
    post = post && ctx->result_path == 1;
    a.mailbox = post ? ctx->mailbox : nullptr;
    a.seq = post ? ++ctx->seq : 0;
    ctx->posted = post;
    ctx->last_slot = ctx->slot;
    a.dbg_joint = dbg_joint;
    a.dbg_h1 = dbg_h1;
    a.dbg_h2 = dbg_h2;
    a.dbg_sums = dbg_sums;
    a.hist_variant = ctx->hist_variant;
    a.phase_mask = ctx->phase_mask;

    const int64_t total = (int64_t)S_local * Wn;
    if (total == 0) {
        // nothing to score: the winner is "none" (key 0); publish it the way the kernel would
        if (out_key) NMI_HIP_TRY(ctx, hipMemsetAsync(out_key, 0, sizeof(unsigned long long), ctx->stream));
        NMI_HIP_TRY(ctx, hipMemsetAsync(ctx->d_keys + ctx->slot, 0, sizeof(unsigned long long), ctx->stream));
        NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (post) ctx->mailbox->word = (unsigned long long)(ctx->seq & 1u) << 63;
        return NMI_OK;
    }
    ctx->slot ^= 1;
    const int cap = ctx->workgroups > 0 ? ctx->workgroups : ctx->compute_units;
    const int workgroups = (int)(total < cap ? total : cap);
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
    if (ctx->profiling) NMI_HIP_TRY(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
    NMI_HIP_TRY(ctx, nmi::launch_grid(a, workgroups, p.use_bg != 0, ctx->stream));
    if (ctx->profiling) {
        NMI_HIP_TRY(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
        ctx->have_timing = true;
    }
    return NMI_OK;
}

// Blocks until the launch numbered ctx->seq has published its winner and returns it.
int fetch_key(nmi_ctx *ctx, unsigned long long *key)
{
    if (ctx->posted) {
        // The last workgroup stores (key | parity << 63) to fine-grained pinned memory; poll that one word.
        const unsigned long long want = (unsigned long long)(ctx->seq & 1u);
        for (uint64_t spin = 0;; ++spin) {
            const unsigned long long word = __atomic_load_n(&ctx->mailbox->word, __ATOMIC_ACQUIRE);
            if ((word >> 63) == want) {
                *key = word & 0x7FFFFFFFFFFFFFFFull;
                return NMI_OK;
            }
            if ((spin & 0xFFFF) == 0xFFFF) {
                // a faulted or finished stream must not leave us spinning
                const hipError_t q = hipStreamQuery(ctx->stream);
                if (q == hipSuccess) break;  // stream drained: fall through to the copy path below
                if (q != hipErrorNotReady) return hip_fail(ctx, q, "hipStreamQuery");
            }
        }
        const unsigned long long word = __atomic_load_n(&ctx->mailbox->word, __ATOMIC_ACQUIRE);
        if ((word >> 63) == want) {
            *key = word & 0x7FFFFFFFFFFFFFFFull;
            return NMI_OK;
        }
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

}  // namespace

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
    if ((e = hipHostMalloc((void **)&ctx->h_key, sizeof(unsigned long long), hipHostMallocDefault)) != hipSuccess)
        return fail(e, "hipHostMalloc(key)");
    if ((e = hipEventCreate(&ctx->ev_start)) != hipSuccess) return fail(e, "hipEventCreate");
    if ((e = hipEventCreate(&ctx->ev_stop)) != hipSuccess) return fail(e, "hipEventCreate");
    if ((e = nmi::launch_table(ctx->table, ctx->npix, ctx->stream)) != hipSuccess) return fail(e, "launch_table");
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return fail(e, "hipStreamSynchronize");
    if (p.max_candidates > 0 && (rc = ensure_ratings(ctx, p.max_candidates)) != NMI_OK) {
        nmi_destroy(ctx);
        return rc;
    }
    *out_ctx = ctx;
    return NMI_OK;
}

int nmi_destroy(nmi_ctx *ctx)
{
    if (!ctx) return NMI_OK;
    DeviceGuard guard(ctx->device);
    if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
    if (ctx->table) (void)hipFree(ctx->table);
    if (ctx->ratings) (void)hipFree(ctx->ratings);
    if (ctx->d_keys) (void)hipFree(ctx->d_keys);
    if (ctx->d_done) (void)hipFree(ctx->d_done);
    if (ctx->mailbox) (void)hipHostFree(ctx->mailbox);
    if (ctx->d_pair_rating) (void)hipFree(ctx->d_pair_rating);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->d_zbuf) (void)hipFree(ctx->d_zbuf);
    for (int i = 0; i < StagingRing::kSlots; ++i) {
        if (ctx->mvp_ring.d[i]) (void)hipFree(ctx->mvp_ring.d[i]);
        if (ctx->mvp_ring.h[i]) (void)hipHostFree(ctx->mvp_ring.h[i]);
        if (ctx->mvp_ring.ev[i]) (void)hipEventDestroy(ctx->mvp_ring.ev[i]);
    }
    if (ctx->d_order) (void)hipFree(ctx->d_order);
    if (ctx->h_order) (void)hipHostFree(ctx->h_order);
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
        ctx->hist_variant = (int)value;
        return NMI_OK;
    case NMI_OPT_PHASE_MASK:
        if (value < 0 || value > 31) return NMI_ERR_INVALID_ARGUMENT;
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
    if (workgroups_per_launch) *workgroups_per_launch = ctx->compute_units;
    if (lds_bytes) *lds_bytes = nmi::grid_kernel_lds_bytes();
    return NMI_OK;
}

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

struct nmi_texture {
    nmi_ctx *ctx = nullptr;
    float *d_luma = nullptr;
    int levels = 0;
    int w[16] = {}, h[16] = {};
    long long off[16] = {};
};

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
    const int64_t need = (int64_t)S * ctx->npix;
    if (need > ctx->zbuf_cap) {
        NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_zbuf) NMI_HIP_TRY(ctx, hipFree(ctx->d_zbuf));
        ctx->d_zbuf = nullptr;
        ctx->zbuf_cap = 0;
        NMI_HIP_TRY(ctx, hipMalloc((void **)&ctx->d_zbuf, (size_t)need * sizeof(uint32_t)));
        ctx->zbuf_cap = need;
    }
    float *d_mvps = nullptr;
    int rc = stage_floats(ctx, ctx->mvp_ring, h_mvps, (size_t)S * 16, &d_mvps);
    if (rc != NMI_OK) return rc;
    NMI_HIP_TRY(ctx, nmi::launch_render_mesh(d_xyz, d_uv, n_triangles, tex->d_luma, tex->levels, tex->w, tex->h, tex->off, d_mvps, S,
                                             ctx->d_zbuf, d_render_stack, ctx->params.width, ctx->params.height, ctx->stream));
    return NMI_OK;
}

// ---------------------------------------------------------------------------------------------------------
// One search level as a captured HIP graph: cloud -> S renders, frame -> Wn warps, grid search, winner to the host.
// Seven dependent operations (two parameter uploads, clear, splat, resolve, warp, key reset + search, winner copy)
// replay with one hipGraphLaunch; only the pinned parameter buffers change between replays.
// ---------------------------------------------------------------------------------------------------------
}  // extern "C"

struct nmi_level {
    nmi_ctx *ctx = nullptr;
    int S = 0, Wn = 0, size = 1;
    uint8_t *d_renders = nullptr, *d_warps = nullptr;
    uint32_t *d_zbuf = nullptr;
    float *d_mvps = nullptr, *h_mvps = nullptr, *d_coeffs = nullptr, *h_coeffs = nullptr;
    int *d_order = nullptr;
    unsigned long long *d_key = nullptr, *h_key = nullptr;
    unsigned int *d_done = nullptr;
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
    void *dev[] = {lv->d_renders, lv->d_warps, lv->d_zbuf, lv->d_mvps, lv->d_coeffs, lv->d_order, lv->d_key, lv->d_done};
    for (void *q : dev)
        if (q) (void)hipFree(q);
    void *host[] = {lv->h_mvps, lv->h_coeffs, lv->h_key};
    for (void *q : host)
        if (q) (void)hipHostFree(q);
    delete lv;
    return NMI_OK;
}

int nmi_level_create(nmi_ctx *ctx, const float *d_xyz, const float *d_red, int64_t n_points, const uint8_t *d_frame, int32_t S,
                     int32_t Wn, float point_size, nmi_level **out)
{
    if (!ctx || !out || !d_frame || S <= 0 || Wn <= 0 || n_points < 0 || (n_points > 0 && (!d_xyz || !d_red)))
        return NMI_ERR_INVALID_ARGUMENT;
    if (!ctx->params.use_bg) return NMI_ERR_UNSUPPORTED;
    *out = nullptr;
    ctx->detail.clear();
    DeviceGuard guard(ctx->device);
    nmi_level *lv = new (std::nothrow) nmi_level;
    if (!lv) return NMI_ERR_INVALID_ARGUMENT;
    lv->ctx = ctx;
    lv->S = S;
    lv->Wn = Wn;
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
    ok(hipMalloc((void **)&lv->d_zbuf, nmi::render_zbuf_words(S, p.width, p.height, lv->size) * sizeof(uint32_t)));
    ok(hipMalloc((void **)&lv->d_mvps, (size_t)S * 16 * sizeof(float)));
    ok(hipMalloc((void **)&lv->d_coeffs, (size_t)Wn * 9 * sizeof(float)));
    ok(hipMalloc((void **)&lv->d_order, (size_t)total * sizeof(int)));
    ok(hipMalloc((void **)&lv->d_key, sizeof(unsigned long long)));
    ok(hipMalloc((void **)&lv->d_done, sizeof(unsigned int)));
    ok(hipHostMalloc((void **)&lv->h_mvps, (size_t)S * 16 * sizeof(float), hipHostMallocDefault));
    ok(hipHostMalloc((void **)&lv->h_coeffs, (size_t)Wn * 9 * sizeof(float), hipHostMallocDefault));
    ok(hipHostMalloc((void **)&lv->h_key, sizeof(unsigned long long), hipHostMallocDefault));
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
    memset(lv->h_mvps, 0, (size_t)S * 16 * sizeof(float));
    memset(lv->h_coeffs, 0, (size_t)Wn * 9 * sizeof(float));
    ok(hipDeviceSynchronize());

    nmi::GridArgs a{};
    a.render_stack = lv->d_renders;
    a.warp_stack = lv->d_warps;
    a.S_local = S;
    a.Wn = Wn;
    a.s_offset = 0;
    a.S_total = S;
    a.width = p.width;
    a.height = p.height;
    a.npix = ctx->npix;
    a.vec_ok = (p.width % 16 == 0) && (((uintptr_t)lv->d_renders | (uintptr_t)lv->d_warps) % 16 == 0);
    a.chunks_per_row = a.vec_ok ? p.width / 16 : 1;
    a.cpr_magic = a.chunks_per_row > 1 ? (uint32_t)((0x100000000ull + a.chunks_per_row - 1) / a.chunks_per_row) : 0u;
    a.shift = ctx->shift;
    a.mode = p.mode;
    a.flip = p.render_bottom_up ? 1 : 0;
    a.table = ctx->table;
    a.order = lv->d_order;
    a.key = lv->d_key;        // reset by a memset node before every replay (the ping-pong of plain launches needs
    a.reset_key = nullptr;    // alternating arguments, which a replayed graph does not have)
    a.done = lv->d_done;
    a.hist_variant = 3;
    a.phase_mask = 3;
    const int cap = ctx->workgroups > 0 ? ctx->workgroups : ctx->compute_units;
    const int workgroups = (int)(total < cap ? total : cap);

    hipStream_t st = ctx->stream;
    if (ok(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal))) {
        ok(hipMemcpyAsync(lv->d_mvps, lv->h_mvps, (size_t)S * 16 * sizeof(float), hipMemcpyHostToDevice, st));
        ok(nmi::launch_render_points(d_xyz, d_red, n_points, lv->d_mvps, S, lv->d_zbuf, lv->d_renders, p.width, p.height, lv->size, st));
        ok(hipMemcpyAsync(lv->d_coeffs, lv->h_coeffs, (size_t)Wn * 9 * sizeof(float), hipMemcpyHostToDevice, st));
        ok(nmi::launch_warp(d_frame, lv->d_coeffs, lv->d_warps, p.width, p.height, Wn, st));
        ok(hipMemsetAsync(lv->d_key, 0, sizeof(unsigned long long), st));
        ok(nmi::launch_grid(a, workgroups, true, st));
        ok(hipMemcpyAsync(lv->h_key, lv->d_key, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
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

int nmi_level_run(nmi_level *lv, const float *h_mvps, const double *h_forward, int64_t *h_best_index, float *h_best_score)
{
    if (!lv || !h_mvps || !h_forward) return NMI_ERR_INVALID_ARGUMENT;
    nmi_ctx *ctx = lv->ctx;
    ctx->detail.clear();
    DeviceGuard guard(ctx->device);
    // the previous replay has completed (this call is blocking), so the pinned parameter buffers are free to rewrite
    memcpy(lv->h_mvps, h_mvps, (size_t)lv->S * 16 * sizeof(float));
    for (int w = 0; w < lv->Wn; ++w) {
        const double *m = h_forward + (size_t)w * 9;
        const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
        if (det == 0.0) return NMI_ERR_INVALID_ARGUMENT;
        const double inv[9] = {(m[4] * m[8] - m[5] * m[7]) / det, (m[2] * m[7] - m[1] * m[8]) / det, (m[1] * m[5] - m[2] * m[4]) / det,
                               (m[5] * m[6] - m[3] * m[8]) / det, (m[0] * m[8] - m[2] * m[6]) / det, (m[2] * m[3] - m[0] * m[5]) / det,
                               (m[3] * m[7] - m[4] * m[6]) / det, (m[1] * m[6] - m[0] * m[7]) / det, (m[0] * m[4] - m[1] * m[3]) / det};
        for (int k = 0; k < 9; ++k) lv->h_coeffs[w * 9 + k] = (float)inv[k];
    }
    NMI_HIP_TRY(ctx, hipGraphLaunch(lv->exec, ctx->stream));
    NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return nmi_key_unpack(*lv->h_key, h_best_index, h_best_score);
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
        unsigned long long *d_key = nullptr;
        unsigned long long *h_key = nullptr;
        hipEvent_t copied = nullptr, done = nullptr;
        int64_t ticket = -1;
        bool waited = true;
    };
    Slot *slots = nullptr;
    uint8_t *d_frame[2] = {nullptr, nullptr};  // frames alternate so an upload never overwrites one still being warped
    uint8_t *d_warps[2] = {nullptr, nullptr};
    hipEvent_t frame_copied = nullptr, warps_free[2] = {nullptr, nullptr};
    int warp_buf = 0;      // buffer holding the current warp stack
    int cur_Wn = 0;
    bool have_warps = false;
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
        ok(hipMalloc((void **)&s.d_key, sizeof(unsigned long long)));
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
    if (!st || !h_render_stack || !ticket || S <= 0 || S > st->max_S) return NMI_ERR_INVALID_ARGUMENT;
    if (h_frame && (!h_forward || Wn <= 0 || Wn > st->max_Wn)) return NMI_ERR_INVALID_ARGUMENT;
    if (!h_frame && !st->have_warps) return NMI_ERR_INVALID_ARGUMENT;
    nmi_ctx *ctx = st->ctx;
    ctx->detail.clear();
    DeviceGuard guard(ctx->device);
    const int64_t t = st->next_ticket;
    nmi_stream::Slot &s = st->slots[t % st->depth];
    if (!s.waited) return NMI_ERR_NOT_READY;  // the ticket that used this slot has not been collected yet
    const size_t npix = (size_t)ctx->npix;

    // copy stream: render stack of this level into the slot (the slot's previous search finished: it was waited for)
    NMI_HIP_TRY(ctx, hipMemcpyAsync(s.d_renders, h_render_stack, npix * S, hipMemcpyHostToDevice, st->copy));
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
        st->cur_Wn = Wn;
        st->have_warps = true;
    }
    NMI_HIP_TRY(ctx, hipEventRecord(s.copied, st->copy));

    // compute stream: search on the slot, winner to pinned host memory
    NMI_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, s.copied, 0));
    int rc = enqueue_grid(ctx, s.d_renders, S, 0, S, st->d_warps[st->warp_buf], st->cur_Wn, nullptr, s.d_key, false, nullptr,
                          nullptr, nullptr, nullptr);
    if (rc != NMI_OK) return rc;
    NMI_HIP_TRY(ctx, hipMemcpyAsync(s.h_key, s.d_key, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    NMI_HIP_TRY(ctx, hipEventRecord(s.done, ctx->stream));
    s.ticket = t;
    s.waited = false;
    *ticket = t;
    ++st->next_ticket;
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
    return nmi_key_unpack(*s.h_key, h_best_index, h_best_score);
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
    int rc = check_grid_args(ctx, render_stack, S_local, s_offset, S_total, warp_stack, Wn);
    if (rc != NMI_OK) return rc;
    DeviceGuard guard(ctx->device);
    rc = enqueue_grid(ctx, render_stack, S_local, s_offset, S_total, warp_stack, Wn, d_ratings, (unsigned long long *)d_key,
                      h_key != nullptr, nullptr, nullptr, nullptr, nullptr);
    if (rc != NMI_OK) return rc;
    if (h_key) {
        unsigned long long k = 0;
        rc = fetch_key(ctx, &k);
        if (rc != NMI_OK) return rc;
        *h_key = k;
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
    int rc = enqueue_grid(ctx, render, 1, 0, 1, warped, 1, ctx->d_pair_rating, nullptr, false, d_joint, d_hist_render,
                          d_hist_warped, d_sums);
    if (rc != NMI_OK) return rc;
    // kernel.cu:100: the blocking 4-byte copy of the score back to the caller.
    NMI_HIP_TRY(ctx, hipMemcpyAsync(h_score, ctx->d_pair_rating, sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return NMI_OK;
}

int nmi_eval_pair(nmi_ctx *ctx, const uint8_t *render, const uint8_t *warped, float *h_score)
{
    return nmi_eval_pair_debug(ctx, render, warped, h_score, nullptr, nullptr, nullptr, nullptr);
}

// ---------------------------------------------------------------------------------------------
// RCCL (resolved at run time so that single-GPU users never load librccl).
// ---------------------------------------------------------------------------------------------
}  // extern "C"

namespace {
struct NcclUniqueId128 {
    char internal[128];
};
typedef int (*fn_get_unique_id)(NcclUniqueId128 *);
typedef int (*fn_comm_init_rank)(void **, int, NcclUniqueId128, int);
typedef int (*fn_comm_destroy)(void *);
typedef int (*fn_all_reduce)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef const char *(*fn_error_string)(int);

struct Rccl {
    void *handle = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_error_string error_string = nullptr;
    bool ok = false;
};

// ncclDataType_t / ncclRedOp_t values of rccl.h (ncclUint64 = 5, ncclMax = 2).
constexpr int kNcclUint64 = 5;
constexpr int kNcclMax = 2;

Rccl &rccl()
{
    static Rccl r = [] {
        Rccl x;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names)
            if ((x.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!x.handle) return x;
        x.get_unique_id = (fn_get_unique_id)dlsym(x.handle, "ncclGetUniqueId");
        x.comm_init_rank = (fn_comm_init_rank)dlsym(x.handle, "ncclCommInitRank");
        x.comm_destroy = (fn_comm_destroy)dlsym(x.handle, "ncclCommDestroy");
        x.all_reduce = (fn_all_reduce)dlsym(x.handle, "ncclAllReduce");
        x.error_string = (fn_error_string)dlsym(x.handle, "ncclGetErrorString");
        x.ok = x.get_unique_id && x.comm_init_rank && x.comm_destroy && x.all_reduce;
        return x;
    }();
    return r;
}

int rccl_fail(nmi_ctx *ctx, int r, const char *what)
{
    if (ctx) {
        char buf[256];
        snprintf(buf, sizeof buf, "%s: %s (%d)", what, rccl().error_string ? rccl().error_string(r) : "rccl", r);
        ctx->detail = buf;
    }
    return NMI_ERR_RCCL - r;
}
}  // namespace

extern "C" {

int nmi_rccl_unique_id(uint8_t out_id[128])
{
    if (!out_id) return NMI_ERR_INVALID_ARGUMENT;
    if (!rccl().ok) return NMI_ERR_UNSUPPORTED;
    NcclUniqueId128 id;
    int r = rccl().get_unique_id(&id);
    if (r != 0) return rccl_fail(nullptr, r, "ncclGetUniqueId");
    memcpy(out_id, id.internal, 128);
    return NMI_OK;
}

int nmi_rccl_comm_init(nmi_ctx *ctx, const uint8_t id[128], int32_t rank, int32_t nranks, void **out_comm)
{
    if (!ctx || !id || !out_comm || nranks <= 0 || rank < 0 || rank >= nranks) return NMI_ERR_INVALID_ARGUMENT;
    if (!rccl().ok) return NMI_ERR_UNSUPPORTED;
    DeviceGuard guard(ctx->device);
    NcclUniqueId128 uid;
    memcpy(uid.internal, id, 128);
    void *comm = nullptr;
    int r = rccl().comm_init_rank(&comm, nranks, uid, rank);
    if (r != 0) return rccl_fail(ctx, r, "ncclCommInitRank");
    *out_comm = comm;
    return NMI_OK;
}

int nmi_rccl_comm_destroy(void *nccl_comm)
{
    if (!nccl_comm) return NMI_OK;
    if (!rccl().ok) return NMI_ERR_UNSUPPORTED;
    int r = rccl().comm_destroy(nccl_comm);
    return r == 0 ? NMI_OK : NMI_ERR_RCCL - r;
}

int nmi_search_grid_rccl(nmi_ctx *ctx, const uint8_t *render_stack, int32_t S_local, int32_t s_offset, int32_t S_total,
                         const uint8_t *warp_stack, int32_t Wn, float *d_ratings, void *nccl_comm, int64_t *h_best_index,
                         float *h_best_score)
{
    if (!nccl_comm) return NMI_ERR_INVALID_ARGUMENT;
    if (!rccl().ok) return NMI_ERR_UNSUPPORTED;
    int rc = nmi_search_grid_shard(ctx, render_stack, S_local, s_offset, S_total, warp_stack, Wn, d_ratings, nullptr, nullptr);
    if (rc != NMI_OK) return rc;
    DeviceGuard guard(ctx->device);
    // The only exchange of the search: 8 bytes per rank, max over ranks.
    unsigned long long *k = ctx->d_keys + ctx->last_slot;
    int r = rccl().all_reduce(k, k, 1, kNcclUint64, kNcclMax, nccl_comm, ctx->stream);
    if (r != 0) return rccl_fail(ctx, r, "ncclAllReduce");
    NMI_HIP_TRY(ctx, hipMemcpyAsync(ctx->h_key, k, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return nmi_key_unpack(*ctx->h_key, h_best_index, h_best_score);
}

}  // extern "C"
