// nmi_capi_rccl.cpp -- the RCCL entry points of include/nmi_hip.h.
#include <dlfcn.h>

#include "nmi_ctx.h"

using namespace nmi_internal;

// RCCL is resolved at run time (dlopen) so that single-GPU users never load librccl.
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

int nmi_internal::rccl_allreduce_key(nmi_ctx *ctx, const unsigned long long *d_send, unsigned long long *d_recv, void *nccl_comm)
{
    if (!nccl_comm) return NMI_ERR_INVALID_ARGUMENT;
    if (!rccl().ok) return NMI_ERR_UNSUPPORTED;
    const int r = rccl().all_reduce(d_send, d_recv, 1, kNcclUint64, kNcclMax, nccl_comm, ctx->stream);
    return r == 0 ? NMI_OK : rccl_fail(ctx, r, "ncclAllReduce");
}

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

int nmi_search_grid_block_rccl(nmi_ctx *ctx, const uint8_t *render_stack, int32_t S_local, int32_t s_offset, int32_t S_total,
                               const uint8_t *warp_stack, int32_t Wn_local, int32_t w_offset, int32_t Wn_total, float *d_ratings,
                               void *nccl_comm, int64_t *h_best_index, float *h_best_score)
{
    if (!nccl_comm) return NMI_ERR_INVALID_ARGUMENT;
    if (!rccl().ok) return NMI_ERR_UNSUPPORTED;
    int rc = search_block(ctx, render_stack, S_local, s_offset, S_total, warp_stack, Wn_local, w_offset, Wn_total, d_ratings, nullptr, nullptr,
                          /*caller_checks=*/true);
    if (rc != NMI_OK) return rc;
    DeviceGuard guard(ctx->device);
    if (ctx->last_parts) {
        // A small block went to the split kernel, whose hand-offs can time out (nmi_split_kernel.hip).  Every rank must
        // issue exactly one collective with a valid key, so this rank settles its own search first: wait, and on a
        // timeout redo it with the one-workgroup kernel (the split forms are paused for a while, nmi_split_status).
        NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (split_timed_out(ctx)) {
            rc = search_block(ctx, render_stack, S_local, s_offset, S_total, warp_stack, Wn_local, w_offset, Wn_total, d_ratings, nullptr,
                              nullptr, true);
            if (rc != NMI_OK) return rc;
        }
    }
    // The only exchange of the search: 8 bytes per rank, max over ranks.  Out of place: the send
    // buffer is this launch's key slot (zero for a rank whose block is empty), the receive buffer a word of its own, so
    // the global winner never lands in a ping-pong slot that a later launch expects to find zero.
    const unsigned long long *send = ctx->d_keys + ctx->last_slot;
    rc = rccl_allreduce_key(ctx, send, ctx->d_reduced_key, nccl_comm);
    if (rc != NMI_OK) return rc;
    NMI_HIP_TRY(ctx, hipMemcpyAsync(ctx->h_key, ctx->d_reduced_key, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return nmi_key_unpack(*ctx->h_key, h_best_index, h_best_score);
}

// One search level sharded over the ranks of `nccl_comm` (the latency form of BASELINE.json configs[4]): this
// rank's block of the level -- its views rendered, its warps made, its cells scored by one graph replay -- then the level's
// only exchange, the 8-byte MAX all-reduce of the packed key on the context's stream, and the winner on every rank.
int nmi_level_run_rccl(nmi_level *lv, const float *h_mvps, const double *h_forward, void *nccl_comm, int64_t *h_best_index,
                       float *h_best_score)
{
    nmi_ctx *ctx = level_ctx(lv);
    if (!ctx || !nccl_comm) return NMI_ERR_INVALID_ARGUMENT;
    if (!rccl().ok) return NMI_ERR_UNSUPPORTED;
    DeviceGuard guard(ctx->device);
    const unsigned long long *send = nullptr;
    int rc = level_enqueue(lv, h_mvps, h_forward, &send);
    if (rc != NMI_OK) return rc;
    rc = rccl_allreduce_key(ctx, send, ctx->d_reduced_key, nccl_comm);
    if (rc != NMI_OK) return rc;
    NMI_HIP_TRY(ctx, hipMemcpyAsync(ctx->h_key, ctx->d_reduced_key, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    NMI_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return nmi_key_unpack(*ctx->h_key, h_best_index, h_best_score);
}

int nmi_search_grid_rccl(nmi_ctx *ctx, const uint8_t *render_stack, int32_t S_local, int32_t s_offset, int32_t S_total,
                         const uint8_t *warp_stack, int32_t Wn, float *d_ratings, void *nccl_comm, int64_t *h_best_index,
                         float *h_best_score)
{
    return nmi_search_grid_block_rccl(ctx, render_stack, S_local, s_offset, S_total, warp_stack, Wn, 0, Wn, d_ratings, nccl_comm,
                                      h_best_index, h_best_score);
}

}  // extern "C"
