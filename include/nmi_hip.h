/*
 * nmi_hip.h -- C ABI of libnmi_hip.so: MI355X (gfx950) implementation of the NMI pose-candidate
 * scoring path of gsanya/orbslam2_NMI.
 *
 * This is the drop-in boundary.  Each entry point names the reference interface it replaces
 * (paths relative to the reference repository root).  Plain pointers and sizes only; no C++,
 * torch or OpenCV types.  All image pointers are DEVICE pointers unless a name starts with h_.
 * Every function returns NMI_OK (0) or a negative error code; nothing calls exit() (the
 * reference aborts the process through checkCudaErrors, Thirdparty/CUDA_Functions/kernel.cu:53-113).
 *
 * Data conventions:
 *   image           uint8 [height][width], contiguous, row stride = width
 *                   (cv::cuda::createContinuous CV_8UC1, Thirdparty/Localization/image.cpp:67).
 *   render          same shape; stored bottom-up when nmi_params.render_bottom_up = 1, which is how
 *                   the reference samples the GL texture (NMI.cu:82).
 *   render_stack    uint8 [S][height][width],  s = (sZ*nSy + sY)*nSx + sX
 *   warp_stack      uint8 [Wn][height][width], w = (wZ*nWy + wY)*nWx + wX
 *   ratings         float [Wn][S]; ratings[w*S + s] == rating[wZ][wY][wX][sZ][sY][sX]
 *                   (Thirdparty/Localization/localization.hpp:36, src/Tracking.cc:1892).
 *   linear index    w*S + s -- the order helperFunctions::find_max_elements scans
 *                   (Thirdparty/Localization/helperFunctions.cpp:53-64).
 */
#ifndef NMI_HIP_H
#define NMI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NMI_HIP_ABI_VERSION 2 /* 2 (round 4): + nmi_pix_status, NMI_OPT_SPLIT 1; round 3 had added nmi_split_status,
                                 nmi_level_create_block, nmi_level_create_mesh_block, nmi_level_run_rccl, nmi_stream_submit_block and
                                 changed NMI_OPT_TILE_QUEUE from queue items to entries per tile bin without a bump */

/* Error codes.  HIP errors are reported as NMI_ERR_HIP - (int)hipError_t, RCCL as NMI_ERR_RCCL - (int)ncclResult_t. */
#define NMI_OK 0
#define NMI_ERR_INVALID_ARGUMENT (-1)
#define NMI_ERR_UNSUPPORTED (-2)
#define NMI_ERR_NO_DEVICE (-3)
#define NMI_ERR_NOT_READY (-4)
#define NMI_ERR_HIP (-1000)
#define NMI_ERR_RCCL (-2000)

/* Score selector; values follow the reference's macros ENMI 0 / SUC 1 (Thirdparty/CUDA_Functions/kernel.cuh:22-23).
 * The reference ignores its run-time NMI_mode argument (kernel.cu:49) and compiles SUC in (NMI.cu:344,352). */
#define NMI_MODE_ENMI 0
#define NMI_MODE_SUC 1

typedef struct nmi_ctx nmi_ctx;

/*
 * Behaviour switches that are compile-time macros in the reference, as run-time fields whose
 * defaults (nmi_params_default) equal the reference's values.
 */
typedef struct nmi_params {
    int32_t width;            /* Camera.Width  / kernel.cu:49 'width'  */
    int32_t height;           /* Camera.Height / kernel.cu:49 'height' */
    int32_t bins;             /* 256 = HISTOGRAM256_BIN_COUNT (NMI.cuh:39); 128/64/32/16 use intensity >> k */
    int32_t mode;             /* NMI_MODE_SUC (kernel.cuh:23) or NMI_MODE_ENMI (kernel.cuh:22) */
    int32_t use_bg;           /* nmi_prop_BG (Thirdparty/Localization/allProperties.hpp:38); default 1 */
    int32_t render_bottom_up; /* 1 = vertical flip of the render as in NMI.cu:82; default 1 */
    int32_t device;           /* HIP device ordinal; -1 = the calling thread's current device */
    int32_t max_candidates;   /* ignored (kept for ABI compatibility): rating tables are caller-owned */
    void *stream;             /* hipStream_t to run on; NULL = the context creates its own */
    int32_t reserved[8];      /* must be 0 */
} nmi_params;

/* Fills *p with the reference defaults: bins 256, SUC, BG on, bottom-up render, current device. */
int nmi_params_default(nmi_params *p, int32_t width, int32_t height);

/*
 * Persistent workspace.  Replaces the per-call allocations of CUDAF::NMIWithCuda_noMask
 * (7 cudaMalloc kernel.cu:67-73, initHistogram256all NMI.cu:171-177, the frees at kernel.cu:103-109
 * and closeHistogram256all NMI.cu:180-185) and the file-static buffers NMI.cu:165-167.
 * One context = one stream; not thread-safe across threads (the reference is called from the
 * Tracking thread only, src/Tracking.cc:1886).  Several contexts may search at the same time from several threads; if they
 * score SMALL grids (up to 64 candidates) concurrently, give each a share of the device with NMI_OPT_WORKGROUPS
 * (e.g. compute units / number of contexts): the split kernel used for small grids wants all workgroups of a launch
 * resident at once, and two such launches that together exceed the device (or a neighbour -- another process, a GL or
 * compute client of the same GPU -- holding compute units) make each other wait until a 2 ms guard ends the wait.  The
 * search of THAT launch is then redone by the one-workgroup-per-candidate kernel (correct results, one slow call) and the
 * split forms pause for the next 16 small-grid launches of the context -- 32, 64, ... 4096 when the retry times out again --
 * after which they are used again; nmi_split_status reports the state.  Calls that only enqueue (h_key == NULL) never use
 * the split kernel: nobody would look for its timeout.
 */
int nmi_create(const nmi_params *params, nmi_ctx **out_ctx);
int nmi_destroy(nmi_ctx *ctx);

/* Run subsequent work on this hipStream_t (not owned).  NULL restores the context's own stream. */
int nmi_set_stream(nmi_ctx *ctx, void *stream);
/* Blocks until everything enqueued on the context's stream has finished. */
int nmi_synchronize(nmi_ctx *ctx);

/*
 * One candidate: replaces CUDAF::NMIWithCuda_noMask (Thirdparty/CUDA_Functions/kernel.cuh:37,
 * kernel.cu:49-114) with the render given as a linear device buffer instead of a GL texture name.
 * Blocking; *h_score receives the value the reference copies back at kernel.cu:100.
 */
int nmi_eval_pair(nmi_ctx *ctx, const uint8_t *render, const uint8_t *warped, float *h_score);

/*
 * A batch of independent candidates, each a (render, warped frame) pair of device images: the scores the reference would
 * get from n consecutive calls of CUDAF::NMIWithCuda_noMask (kernel.cu:49-114), e.g. the Wn warps of the inner loop at
 * src/Tracking.cc:1883-1894 against the render of the current outer iteration.  h_renders / h_warps are HOST arrays of n
 * device pointers (the same pointer may appear many times); h_scores receives n floats.  Blocking.  One launch scores up
 * to compute_units / 4 pairs (several workgroups per pair), larger batches take several launches.
 */
int nmi_eval_pairs(nmi_ctx *ctx, const uint8_t *const *h_renders, const uint8_t *const *h_warps, int32_t n, float *h_scores);

/*
 * Same evaluation, additionally exporting the exact integer histograms and the three entropy sums
 * (the intermediate buffers d_JointHistogram / d_Histogram1 / d_Histogram2 and element 0 of
 * d_Entropy1 / d_Entropy2 / d_JointEntropyShort before the score is formed, kernel.cu:59-95).
 * Any of the device output pointers may be NULL.  joint is [256][256] indexed [render][warped].
 */
int nmi_eval_pair_debug(nmi_ctx *ctx, const uint8_t *render, const uint8_t *warped, float *h_score,
                        uint32_t *d_joint /*[65536]*/, uint32_t *d_hist_render /*[256]*/,
                        uint32_t *d_hist_warped /*[256]*/, float *d_sums /*[3]: A1, A2, A3*/);

/*
 * Whole candidate grid + best-pose pick: replaces the 6-nested loop of Tracking::RelocalizeWithNMI
 * (src/Tracking.cc:1879-1902: S renders x Wn warps calls of NMIWithCuda_noMask) and
 * helperFunctions::find_max_elements + the caller's [0] pick (helperFunctions.cpp:50-103,
 * Tracking.cc:1905,1952-1953).
 *   d_ratings       device float [Wn*S], or NULL when only the winner is wanted (no rating table is stored then).
 *   h_best_index    linear index w*S + s of the winner: max starts at 0, strict '>', lowest index
 *                   among cells equal to the max; -1 if no cell qualifies (every score negative or
 *                   NaN, where the reference indexes an empty vector).
 *   h_best_score    the winner's score.
 * Blocking (8-byte read-back of the packed winner).  With the default result path the call returns when the kernel's
 * last workgroup has posted the winner to pinned host memory; when d_ratings is given it additionally waits for the
 * context's stream, so the table is complete and visible to every stream when the call returns.
 */
int nmi_search_grid(nmi_ctx *ctx, const uint8_t *render_stack, int32_t S, const uint8_t *warp_stack, int32_t Wn,
                    float *d_ratings, int64_t *h_best_index, float *h_best_score);

/*
 * Sharded form for one rank of a multi-GPU search (new; the reference is single-GPU).
 * This rank holds renders [s_offset, s_offset + S_local) of a global grid with S_total renders and
 * the full warp stack.  The kernel writes the rank's best candidate as a packed key to *d_key
 * (device uint64, may be NULL to use an internal slot) and, if h_key != NULL, blocks and copies it
 * to the host.  key = float_bits(score) << 32 | (0xFFFFFFFF - global_linear_index) for score >= 0
 * and 0 for "no candidate"; the maximum over ranks (uint64 or int64 MAX all-reduce) is the global
 * winner with the reference's lowest-index tie-break.  d_ratings (nullable) is [Wn][S_local].
 * With h_key == NULL the call only enqueues work on the context's stream.
 */
int nmi_search_grid_shard(nmi_ctx *ctx, const uint8_t *render_stack, int32_t S_local, int32_t s_offset,
                          int32_t S_total, const uint8_t *warp_stack, int32_t Wn, float *d_ratings,
                          uint64_t *d_key, uint64_t *h_key);

/*
 * The same for any rectangular block of the grid: renders [s_offset, s_offset + S_local) x warps [w_offset, w_offset +
 * Wn_local) of an S_total x Wn_total grid (used when there are fewer renders than ranks and the warp axis is sharded
 * instead).  d_ratings (nullable) is [Wn_local][S_local]; the key carries the global linear index w * S_total + s.
 */
int nmi_search_grid_block(nmi_ctx *ctx, const uint8_t *render_stack, int32_t S_local, int32_t s_offset, int32_t S_total,
                          const uint8_t *warp_stack, int32_t Wn_local, int32_t w_offset, int32_t Wn_total, float *d_ratings,
                          uint64_t *d_key, uint64_t *h_key);

/*
 * Warp-stack producer: replaces Image::calculateWarping (Thirdparty/Localization/image.cpp:115-128),
 * i.e. Wn calls of cv::cuda::warpPerspective(frame, warped[w], M[w], size) with INTER_LINEAR / BORDER_CONSTANT 0.
 *   h_forward   host doubles [Wn][9], row-major 3x3 forward homographies K*R*K^-1 exactly as image.cpp:106 stores them
 *               (nmi_warp_homographies builds them from K and the warp axes of a grid);
 *   d_frame     device uint8 [H][W]; d_warp_stack device uint8 [Wn][H][W], w = (wZ*nWy + wY)*nWx + wX.
 * Enqueued on the context's stream (no synchronisation).  OpenCV is not part of the reference tree: parity of the
 * interpolation arithmetic is unpinned (see the kernel comment).
 */
int nmi_warp_homographies(const double K[9], const int32_t num_warp_xyz[3], const float step_rad_xyz[3],
                          double *h_forward /*[Wn][9]*/);
int nmi_warp_stack(nmi_ctx *ctx, const uint8_t *d_frame, const double *h_forward, int32_t Wn, uint8_t *d_warp_stack);

/*
 * Render-stack producer for coloured point clouds: replaces Rendering<4>::renderToTextureOnGPU
 * (Thirdparty/Localization/rendering.hpp:530-630, nmi_prop_RENDER 4, shaders/ShadingWithColor.*) for S camera
 * translations of one pose -- no OpenGL.  nmi_render_mvp builds Projection * glm::lookAt for one view exactly as
 * rendering.hpp:196-202,547-553 (column-major float[16] like glm); nmi_render_points draws the cloud into
 * d_render_stack [S][H][W] (uint8, bottom-up rows like the GL texture, background 255).  d_xyz: float [N][3] vertices
 * (loadXYZ output, objloader.cpp:257-260), d_red: float [N] red colour component (objloader.cpp:261: file value / 256).
 * Enqueued on the context's stream.  Parity with an OpenGL driver's rasteriser is unpinned (kernel comment).
 */
typedef struct nmi_render_params {
    double fx, fy, cx, cy;  /* Camera.fx .. Camera.cy (localization.cpp:149-152) */
    float near_plane;       /* NMI.Render.NearPlane */
    float far_plane;        /* NMI.Render.FarPlane  */
    float point_size;       /* NMI.Render.PointSize -> glPointSize (rendering.hpp:307) */
} nmi_render_params;
int nmi_render_mvp(const nmi_render_params *rp, const float cam_pos[3], const float cam_look_at[3], const float cam_up[3],
                   const float translation[3], float out_mvp[16]);
int nmi_render_points(nmi_ctx *ctx, const float *d_xyz, const float *d_red, int64_t n_points, const float *h_mvps, int32_t S,
                      float point_size, uint8_t *d_render_stack);

/*
 * Render-stack producer for textured meshes (nmi_prop_RENDER 1, the reference's default): replaces
 * Rendering<1>::renderToTextureOnGPU (rendering.hpp:530-630, shaders/ShadingWithTexture.*) -- no OpenGL.
 *   nmi_texture_create  takes the RGB8 image exactly as loadBMP_custom passes it to glTexImage2D (texture.cpp:31-86: row 0
 *                       = v 0, three bytes per texel in file order; the shader weighs byte 0 with 0.299, byte 1 with 0.587,
 *                       byte 2 with 0.114), builds the mip chain glGenerateMipmap would (2x2 box, RGB8 per level) and keeps
 *                       per-level luma on the device.  Sides up to 32,768 texels and at most 2^30 texels in the whole
 *                       pyramid (a 28,000 x 28,000 image), else NMI_ERR_UNSUPPORTED.
 *   nmi_render_mesh     d_xyz float [3*T][3] and d_uv float [3*T][2]: the expanded per-corner arrays loadOBJ produces
 *                       (objloader.cpp:140-224); h_mvps as for nmi_render_points; output uint8 [S][H][W], bottom-up rows,
 *                       background 255.  Back faces culled (counter-clockwise front), depth test LESS, GL_REPEAT,
 *                       GL_LINEAR / GL_LINEAR_MIPMAP_LINEAR.  Triangles are clipped against the near plane in clip space
 *                       (a ground plane passing under the camera keeps its visible part).  Among fragments of equal 24-bit
 *                       depth the triangle drawn first (lowest index) wins, as GL_LESS leaves it.  At most 2^30 - 1 triangles.
 * Enqueued on the context's stream.  Parity with an OpenGL driver is unpinned (kernel comment).
 */
typedef struct nmi_texture nmi_texture;
int nmi_texture_create(nmi_ctx *ctx, const uint8_t *h_rgb, int32_t tex_width, int32_t tex_height, nmi_texture **out);
int nmi_texture_destroy(nmi_texture *tex);
int nmi_render_mesh(nmi_ctx *ctx, const float *d_xyz, const float *d_uv, int64_t n_triangles, const nmi_texture *tex,
                    const float *h_mvps, int32_t S, uint8_t *d_render_stack);

/*
 * Map order.  What the renderers draw does not depend on the order of the points / triangles (the depth test is a
 * minimum); how fast they draw does: neighbours in memory are culled together and their fragments share cache lines of the
 * depth buffer (3 M points into 27 views: 96 us in scan order, 406 us shuffled, 116 us after nmi_sort_points).  These two
 * calls copy a map into Morton order of its positions (triangles: centroids), on the device; a loader calls them once
 * after loadXYZ / loadOBJ (objloader.cpp:140-264), whose arrays are in file order.  Outputs must not alias the inputs.
 * Blocking (temporary device storage: 16 bytes per record).  n below 2^32.
 */
int nmi_sort_points(nmi_ctx *ctx, const float *d_xyz /*[N][3]*/, const float *d_red /*[N]*/, int64_t n_points, float *d_xyz_out,
                    float *d_red_out);
int nmi_sort_triangles(nmi_ctx *ctx, const float *d_xyz /*[3*T][3]*/, const float *d_uv /*[3*T][2]*/, int64_t n_triangles,
                       float *d_xyz_out, float *d_uv_out);

/*
 * One whole search level on the device as a captured HIP graph: S renders of the cloud (nmi_render_points), Wn warps of the
 * frame (nmi_warp_stack) and the S x Wn search (nmi_search_grid) replay with a single hipGraphLaunch -- one chain of four
 * kernel nodes (parameter fetch + a cull of the cloud's 64-point boxes against the planes around all views; warp stack + splat of
 * the surviving points in one launch; resolve; search), no copy nodes, no branches.
 * Create once per (cloud, frame, S, Wn); nmi_level_run takes this level's S view matrices (nmi_render_mvp, float[S][16]) and
 * Wn forward homographies (nmi_warp_homographies, double[Wn][9]) and blocks until the search has posted its winner to pinned
 * host memory (the context's stream drains a few microseconds later; work enqueued on it afterwards is ordered as usual).
 * Same results as the three calls made one after the other.  The MAP is taken as it is at creation: a point-cloud level
 * keeps its own packed copy (16 bytes per point + one bounding box per 64 points), and d_xyz / d_red need not outlive the
 * call; a mesh level reads d_xyz / d_uv and the texture in place (they must stay valid and unchanged).  d_frame must stay
 * valid in place; its CONTENTS may change between runs (the next camera frame).
 */
typedef struct nmi_level nmi_level;
int nmi_level_create(nmi_ctx *ctx, const float *d_xyz, const float *d_red, int64_t n_points, const uint8_t *d_frame, int32_t S,
                     int32_t Wn, float point_size, nmi_level **out);
/* The same with the textured mesh as the map (nmi_prop_RENDER 1): d_xyz / d_uv / tex as for nmi_render_mesh. */
int nmi_level_create_mesh(nmi_ctx *ctx, const float *d_xyz, const float *d_uv, int64_t n_triangles, const nmi_texture *tex,
                          const uint8_t *d_frame, int32_t S, int32_t Wn, nmi_level **out);
int nmi_level_run(nmi_level *lv, const float *h_mvps, const double *h_forward, int64_t *h_best_index, float *h_best_score);
/*
 * Level sharded over ranks (new; SURVEY.md 8e applied to the device-side level -- the LATENCY form of BASELINE.json configs[4]:
 * keyframes of a live sequence are not independent, each search is seeded from the drift since the previous NMI fix,
 * src/Tracking.cc:2001-2053, inside the sequential Track(), :598-616, so a live level can only be made faster by sharing ITS
 * candidates).  A _block level covers renders [s_offset, s_offset + S_local) x warps [w_offset, w_offset + Wn_local) of an
 * S_total x Wn_total level: the rank renders only its S_local views (h_mvps [S_local][16]), warps the frame Wn_local times
 * (h_forward [Wn_local][9]) and scores its cells with GLOBAL linear indices w * S_total + s.  Either count may be 0 (more
 * ranks than cells on the sharded axis): the rank then only takes part in the exchange.  nmi_level_run on a block returns
 * the block's own winner (for callers that reduce the keys themselves, e.g. torch.distributed over gloo: nmi_key_pack);
 * nmi_level_run_rccl adds the level's only exchange -- ncclAllReduce(ncclMax, ncclUint64) of the 8-byte key on the
 * context's stream, out of place -- and returns the level's winner on every rank.  Every rank of the communicator must
 * call it once per level.
 */
int nmi_level_create_block(nmi_ctx *ctx, const float *d_xyz, const float *d_red, int64_t n_points, const uint8_t *d_frame,
                           int32_t S_local, int32_t s_offset, int32_t S_total, int32_t Wn_local, int32_t w_offset, int32_t Wn_total,
                           float point_size, nmi_level **out);
int nmi_level_create_mesh_block(nmi_ctx *ctx, const float *d_xyz, const float *d_uv, int64_t n_triangles, const nmi_texture *tex,
                                const uint8_t *d_frame, int32_t S_local, int32_t s_offset, int32_t S_total, int32_t Wn_local,
                                int32_t w_offset, int32_t Wn_total, nmi_level **out);
int nmi_level_run_rccl(nmi_level *lv, const float *h_mvps, const double *h_forward, void *nccl_comm, int64_t *h_best_index,
                       float *h_best_score);
/* Host copies of what the latest nmi_level_run produced: the S renders [S][H][W], the Wn warps [Wn][H][W] and the rating
 * table [Wn][S] (any pointer may be NULL).  Blocking; for tests and debugging (the reference's orb_prop_log dumps,
 * src/Tracking.cc:1911-1948, serve the same purpose). */
int nmi_level_copy_outputs(nmi_level *lv, uint8_t *h_renders, uint8_t *h_warps, float *h_ratings);
int nmi_level_destroy(nmi_level *lv);

/*
 * Streaming form (BASELINE.json config 5): keyframes / search levels whose render stacks arrive from host memory.
 * A stream owns `depth` device slots; nmi_stream_submit enqueues, without blocking,
 *   copy stream    : hipMemcpyAsync of the pinned host render stack [S][H][W] (and the frame [H][W], if given) into a slot
 *   compute stream : (frame given) warp-stack production for h_forward [Wn][9]; the grid kernel; 8-byte winner -> pinned host
 * so the H2D copy of level i+1 overlaps the search of level i (the reference uploads and searches serially,
 * src/Tracking.cc:1871-1902).  Tickets complete in submission order.  nmi_stream_wait blocks on one ticket.
 * A slot is reused by submission i + depth only after ticket i was waited for.  h_* buffers must stay valid (and should be
 * page-locked, e.g. hipHostMalloc) until the ticket completes.  Passing h_frame == NULL re-uses the warp stack produced by
 * the most recent submission that had a frame.
 * A ticket whose small grid timed out in the split kernel (see nmi_create) is redone inside nmi_stream_wait when its warp
 * stack is still on the device; when a later frame has replaced it the wait returns NMI_ERR_NOT_READY, the ticket's rating
 * table is withheld (nmi_stream_copy_ratings fails) and the caller submits that level again.
 */
typedef struct nmi_stream nmi_stream;
int nmi_stream_create(nmi_ctx *ctx, int32_t max_S, int32_t max_Wn, int32_t depth, nmi_stream **out);
int nmi_stream_destroy(nmi_stream *st);
int nmi_stream_submit(nmi_stream *st, const uint8_t *h_render_stack, int32_t S, const uint8_t *h_frame,
                      const double *h_forward, int32_t Wn, int64_t *ticket);
/* Block form for a level sharded over ranks (see nmi_level_create_block): this rank uploads and scores only renders
 * [s_offset, s_offset + S_local) of the level's S_total (the H2D traffic that bounds the streamed form is divided by the
 * number of ranks) against warps [w_offset, w_offset + Wn_local) of Wn_total, which it makes locally from the frame;
 * h_forward holds the Wn_local homographies of its block.  With nccl_comm != NULL the key is MAX-all-reduced on the compute
 * stream right behind the search (every rank submits the levels in the same order) and the ticket completes with the
 * level's winner; with NULL it completes with the block's own key.  S_local may be 0 (h_render_stack may then be NULL). */
int nmi_stream_submit_block(nmi_stream *st, const uint8_t *h_render_stack, int32_t S_local, int32_t s_offset, int32_t S_total,
                            const uint8_t *h_frame, const double *h_forward, int32_t Wn_local, int32_t w_offset, int32_t Wn_total,
                            void *nccl_comm, int64_t *ticket);
int nmi_stream_wait(nmi_stream *st, int64_t ticket, int64_t *h_best_index, float *h_best_score);
/* Optional rating tables: after nmi_stream_keep_ratings(st, 1) every submission also stores its [Wn][S] table in its slot;
 * nmi_stream_copy_ratings copies the table of a ticket that has been waited for (n = Wn * S floats), valid until the
 * slot is submitted to again. */
int nmi_stream_keep_ratings(nmi_stream *st, int32_t enabled);
int nmi_stream_copy_ratings(nmi_stream *st, int64_t ticket, float *h_ratings, int64_t n);

/* Packed-key helpers (host side, pure). */
uint64_t nmi_key_pack(float score, int64_t global_linear_index);
int nmi_key_unpack(uint64_t key, int64_t *global_linear_index, float *score);

/*
 * RCCL form: nmi_search_grid_shard / _block followed by ncclAllReduce(ncclMax, ncclUint64) of the key on the context's
 * stream over `nccl_comm` (an ncclComm_t created by the caller), then the 8-byte read-back.  A rank whose block is
 * empty (S_local == 0 or Wn_local == 0: fewer renders than ranks on a render-sharded level) contributes "no candidate"
 * and still takes part in the collective.  The _block form shards either axis (see nmi_search_grid_block).
 */
int nmi_search_grid_rccl(nmi_ctx *ctx, const uint8_t *render_stack, int32_t S_local, int32_t s_offset,
                         int32_t S_total, const uint8_t *warp_stack, int32_t Wn, float *d_ratings,
                         void *nccl_comm, int64_t *h_best_index, float *h_best_score);
int nmi_search_grid_block_rccl(nmi_ctx *ctx, const uint8_t *render_stack, int32_t S_local, int32_t s_offset, int32_t S_total,
                               const uint8_t *warp_stack, int32_t Wn_local, int32_t w_offset, int32_t Wn_total, float *d_ratings,
                               void *nccl_comm, int64_t *h_best_index, float *h_best_score);

/* Communicator bootstrap for hosts that have no ncclComm_t yet: rank 0 calls nmi_rccl_unique_id and
 * ships the 128 bytes to the other ranks by any means; every rank then calls nmi_rccl_comm_init. */
int nmi_rccl_unique_id(uint8_t out_id[128]);
int nmi_rccl_comm_init(nmi_ctx *ctx, const uint8_t id[128], int32_t rank, int32_t nranks, void **out_comm);
int nmi_rccl_comm_destroy(void *nccl_comm);

/*
 * Timing of the dominant kernel with HIP events on the context's stream.  When enabled, every grid /
 * pair launch is bracketed by hipEventRecord; nmi_last_kernel_ms returns the duration of the most
 * recent launch (synchronises on the stop event).
 */
int nmi_set_profiling(nmi_ctx *ctx, int32_t enabled);
int nmi_last_kernel_ms(nmi_ctx *ctx, float *h_ms);

/*
 * Tuning / ablation knobs (defaults are the shipped configuration; results stay exact for every value
 * except NMI_OPT_HIST_VARIANT = 2, which skips the counter-wrap bookkeeping, and a partial phase mask).
 */
#define NMI_OPT_HIST_VARIANT 1 /* 3 optimistic + verify + exact redo (default), 1 exact wrap bookkeeping throughout.  The
                                  experiments 0 (per-pixel wrap test), 2 (unchecked) and 4 (histogram and decode overlapped by
                                  wavefront role; exact but slower, profiles/NOTES.md) exist only in a library built with
                                  -DNMI_BUILD_ABLATIONS (NMI_ERR_UNSUPPORTED otherwise). */
#define NMI_OPT_PHASE_MASK 2   /* bit 0 histogram phase, bit 1 decode + score, bit 2 disable the flat-chunk shortcut; default 3.
                                  Bit 9 (tests): one part of the split kernel withholds its hand-off, so the bounded wait (2 ms) of
                                  the scoring workgroup times out and the call is redone by the one-workgroup kernel. */
#define NMI_OPT_WORKGROUPS 3   /* workgroups per launch; 0 = one per compute unit (default) */
#define NMI_OPT_RESULT_PATH 4  /* how the 8-byte winner reaches the host: 1 the kernel posts it to pinned host memory
                                  and the call polls it (default), 0 hipMemcpyAsync + hipStreamSynchronize */
#define NMI_OPT_XCD_TILING 5   /* 1 (default): candidates are visited in (warp x render) tiles so that the 32 workgroups
                                  of one XCD share ~12 images in its L2; 0: linear order.  Same results either way. */
#define NMI_OPT_TILE_QUEUE 6   /* mesh renderer: usable entries of each screen-tile bin (at most 255, the default; the value is
                                  clamped).  A triangle that finds a bin full is rasterised by its own lane instead; 0 = every
                                  triangle is.  Same image for every value (small values exercise the overflow path in tests). */
#define NMI_OPT_CLIP_QUEUE 11  /* mesh renderer: capacity of the queue that hands (triangle, view) pairs crossing the near plane to
                                  the clipping pass, at most 262144 (default).  With more such pairs than that the clipping
                                  pass finds them again itself; same image for every value (0 exercises that path in tests). */
#define NMI_OPT_SPLIT 7        /* small grids (nmi_eval_pair, collapsed search levels): K workgroups per candidate, each owning
                                  256 / K rows of the joint histogram, optionally x P pixel ranges (NMI_OPT_SPLIT_PIXELS);
                                  bit-identical results.  -1 (default, on 256 compute units): 8 x 4 up to 8 candidates, 8 x 2
                                  up to 16, 4 x 2 up to 32; 33 ... 128 candidates: 1 x P, pixel ranges only (nmi_pix_status; P = 4
                                  up to 64 candidates, 3 up to 85, 2 up to 128); none for larger grids; 0: never; 2 / 4 / 8: that
                                  K whenever the grid fits; 1: pixel ranges only, NMI_OPT_SPLIT_PIXELS = 2 ... 5 of them, whenever
                                  the grid fits. */
#define NMI_OPT_WAIT_MODE 8    /* how a blocking call waits for the posted result: 0 (default) spins on the pinned word
                                  (lowest latency, occupies the calling core for the search), 1 yields the core between
                                  polls (sched_yield; for hosts whose other threads need the core, e.g. ORB-SLAM2's
                                  LocalMapping / LoopClosing).  NMI_OPT_RESULT_PATH 0 sleeps in hipStreamSynchronize instead. */
#define NMI_OPT_SPLIT_PIXELS 10 /* additionally cut the pixels of each pair into 2 or 4 ranges (one workgroup per row part and
                                  range, merged per row part: nmi_eval_pair = 8 x 4 = 32 workgroups; 4 with 8 row parts only).
                                  -1 (default): see NMI_OPT_SPLIT; 1: never; 2 / 4: that many when it exists and fits
                                  (NMI_OPT_SPLIT 1: 2 ... 5). */
#define NMI_OPT_CONTENT_PATH 12 /* frames with few distinct intensities (posterised, thresholded, quantised): -1 (default)
                                  automatic -- every search by the general kernel also counts the distinct intensities (bins) its
                                  candidates' marginal histograms hold, (nr, nw), at no extra launch, and every few-levels search
                                  probes its two stacks; while nr * nw <= NMI_OPT_FEWLEVELS_BINS searches go down the few-levels
                                  path (rank images + 32-bit replicated counters; csrc/nmi_fewlevels_kernel.hip; with fewer than
                                  256 bins the background rule must be on).  A change of content costs one search on the slower
                                  path either way.  0: never; 1: always try it first.  Every few-levels search falls back to the
                                  general kernel on the device when its stacks do not qualify: results never depend on it. */
#define NMI_OPT_FEWLEVELS_BINS 13 /* largest nr * nw sent down the few-levels path (1..4096, default 4096) */
#define NMI_OPT_PIX_OWNER_BIAS 14 /* pixel-range kernel (nmi_pix_status): pixels a candidate's owner adds beyond an equal share of the
                                  pair while its helpers' counters travel to it (default 49152 = 3.4 us of one CU's histogram
                                  phase); a matter of speed only */
#define NMI_OPT_STAMPS 9       /* profiling tools only: value = device pointer to uint64 [workgroups][8]; workgroups of the
                                  split kernel store wall-clock stamps (100 MHz) at their phase boundaries there; 0 = off */
int nmi_set_option(nmi_ctx *ctx, int32_t option, int64_t value);

/* Split-kernel liveness (see nmi_create): *timeouts = hand-off timeouts so far, *cooldown_calls_left = small-grid launches
 * that will still go through the one-workgroup kernel before the split forms are tried again, *next_cooldown = length of
 * the pause the next timeout would start, *last_launch_parts = row parts per candidate of the most recent launch (0: the
 * one-workgroup kernel scored it).  Any pointer may be null.  Does not wait. */
int nmi_split_status(nmi_ctx *ctx, int32_t *timeouts, int32_t *cooldown_calls_left, int32_t *next_cooldown, int32_t *last_launch_parts);
/* Mid-size grids (33 ... 128 candidates on 256 compute units: the live strategy's collapsed-axis levels, Tracking.cc:2014-2043,
 * and a rank's share of a sharded 729-candidate grid) are scored by P workgroups per candidate, each adding a range of the
 * pair's PIXELS into a histogram of its own (csrc/nmi_pix_kernel.hip); bit-identical results; no residence condition, so
 * enqueue-only calls use it too.  *last_launch_ranges = P of the most recent launch (0: another kernel scored it);
 * *healed = candidates so far whose owner gave up waiting for a helper (2 ms) and scored them alone -- nothing for the
 * host to redo.  Waits for the context's stream when healed is asked for.  Any pointer may be null. */
int nmi_pix_status(nmi_ctx *ctx, int32_t *last_launch_ranges, int32_t *healed);

/* Introspection. */
/* Copies the context's per-count term table, term[c] = (c/len) * log2(c/len) in the reference's fp32 form with
 * len = width * height (ComputeEntropyKernel, NMI.cu:242-263; term[0] = 0), c = 0..len, to host memory.
 * n must be width * height + 1.  Blocking.  Lets a test compare every entry with its oracle. */
int nmi_copy_term_table(nmi_ctx *ctx, float *h_out, int64_t n);
int nmi_abi_version(void);
const char *nmi_error_string(int code);
const char *nmi_last_error_detail(nmi_ctx *ctx); /* text of the last failing HIP/RCCL call, or "" */
int nmi_get_info(nmi_ctx *ctx, int32_t *compute_units, int32_t *workgroups_per_launch, int32_t *lds_bytes);
/* How the most recent search was scored (waits for it): *few_levels = 1 if it was sent down the few-levels path
 * (NMI_OPT_CONTENT_PATH) AND stayed there, 0 if the general kernel scored it; *nr, *nw = distinct intensities the most
 * recent probe found in a render / warp stack (0, 0 before the first probe).  Any pointer may be null.  Diagnostics. */
int nmi_last_content(nmi_ctx *ctx, int32_t *few_levels, int32_t *nr, int32_t *nw);

#ifdef __cplusplus
}
#endif
#endif /* NMI_HIP_H */
