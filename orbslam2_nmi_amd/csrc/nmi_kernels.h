// nmi_kernels.h -- internal interface between the C ABI (nmi_capi.cpp) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NMI_BLOCK_THREADS 1024
#define NMI_MODE_ENMI_ 0  // Thirdparty/CUDA_Functions/kernel.cuh:22
#define NMI_MODE_SUC_ 1   // Thirdparty/CUDA_Functions/kernel.cuh:23

namespace nmi {

// Host-visible result slot (fine-grained pinned memory): the kernel's last workgroup writes one 8-byte word,
// bits 0..62 = packed winner key, bit 63 = parity of the posting launch's sequence number.
struct Mailbox {
    unsigned long long word;
    unsigned long long pad;
};

// What nmi_levels_kernel found in the two stacks of a search (nmi_fewlevels_kernel.hip); lives in device memory.
struct LevelPlan {
    uint32_t use;             // 1: nmi_fewlevels_kernel scores this search, 0: nmi_grid_kernel (its gated launch) does
    uint32_t nr, nw;          // distinct intensities (bins, with fewer than 256 of them) in the render stack / in the warp stack
    uint32_t copies;          // interleaved copies of each counter (8, 16 or 32)
    uint32_t ticket;          // probe workgroups that have merged their masks (left 0)
    uint32_t pad[3];
    uint32_t mask[2][8];      // presence bits being merged by a probe (left 0)
    uint8_t rank_r[256], rank_w[256];    // intensity -> rank of its bin among the bins present (0 where absent)
    uint8_t level_r[256], level_w[256];  // rank -> bin
    // Every search by nmi_grid_kernel is a probe as well: its workgroups OR the bins their candidates' marginal histograms hold
    // into seen[] (word r * 16 + i, bit k: bin i + 16 k for k < 8, bin i + 16 (k - 8) + 128 above; r = 0 render, 1 frame) and the
    // last workgroup counts them, posts (nr, nw) to *seen_post and leaves seen[] zero.
    uint32_t seen[32];
    uint32_t seen_state;                 // 1: the last post said "few levels" (nr * nw <= seen_max_joint), 0: it said not (or none yet)
    uint32_t seen_max_joint;             // NMI_OPT_FEWLEVELS_BINS, for that comparison
    unsigned long long *seen_post;       // pinned host word: (0x80000000 | a time stamp) << 32 | nr << 16 | nw -- written only when
                                         // the verdict CHANGES (a store to host memory holds the kernel's end back by a PCIe trip)
};

// Split-kernel timeouts are reported per launch through a ring of pinned words indexed by the launch's epoch: it must be longer
// than the number of checked split launches that can be in flight (a stream of depth <= 64; an nmi_eval_pairs batch is cut to 128).
constexpr uint32_t kSplitRing = 256;

struct GridArgs {
    const uint8_t *render_stack;  // [S_local][H][W]
    const uint8_t *warp_stack;    // [Wn][H][W]
    // nmi_eval_pairs (split kernel only): candidate p scores (pair_renders[p], pair_warps[p]) instead of stack entries;
    // the tables live in pinned host memory that the device reads directly.  nullptr = stacks.
    const uint8_t *const *pair_renders;
    const uint8_t *const *pair_warps;
    int S_local, Wn;
    int s_offset, S_total;        // position of this shard in the global grid
    int w_offset;                 // first warp of this shard in the global grid (0 unless the warp axis is sharded)
    int width, height, npix;
    int chunks_per_row;           // width / 16 (1 for frames narrower than 32 pixels)
    uint32_t cpr_magic;           // ceil(2^32 / chunks_per_row)
    int vec_ok;                   // width % 16 == 0, width >= 32 and both stacks 16-byte aligned: rows are whole aligned chunks
    // 16-byte chunk c of the frame (row y = c / chunks_per_row) meets render chunk c + flip_base + y * flip_row:
    // (0, 0) for a top-down render, ((H - 1) * cpr, -2 * cpr) for a bottom-up one (NMI.cu:82)
    int flip_base, flip_row;
    int shift;                    // intensity >> shift (bins = 256 >> shift)
    int mode;                     // NMI_MODE_*_
    int flip;                     // render stored bottom-up (NMI.cu:82)
    const float *table;           // [npix + 1] per-count entropy terms
    // One slot, two users that never meet (nmi_grid_kernel sits at its register cap and its allocation shifts with the
    // offsets and the size of this struct, so the struct keeps both):
    union {
        uint32_t *scratch;        // pipelined kernel (ablation build): [workgroups][2][32768] drained packed counters
        const LevelPlan *plan;    // few-levels path: the plan of this search; the gated nmi_grid_kernel runs iff plan->use == 0
    };
    const int *order;             // [S_local * Wn] candidate visited at each ordinal (XCD-aware tiling), or nullptr = identity
    float *ratings;               // [Wn][S_local] or nullptr
    unsigned long long *key;      // packed arg-max slot of this launch (zero on entry)
    unsigned long long *reset_key;  // the slot of the next launch: cleared by this one (ping-pong), or nullptr
    unsigned long long *out_key;    // optional copy of the final key for a caller / a collective
    unsigned int *done;             // count of finished workgroups (zero on entry, left zero)
    Mailbox *mailbox;               // pinned host memory the last workgroup posts the winner to, or nullptr
    unsigned int seq;               // sequence number of this posting launch (its parity is posted with the winner)
    unsigned long long *score_post; // nmi_eval_pair: pinned host word that receives (score bits | seq << 32), or nullptr
    struct SplitSlab *slabs;        // split kernel: one hand-off slab per candidate (see nmi_split_kernel.hip)
    unsigned long long *blocks;     // split kernel with pixel parts: [candidate][row part][pixel part] blocks of 24+24+16-bit granules
    uint32_t epoch;                 // split kernel: this launch's tag (never 0; its low 16 bits never 0)
    uint32_t *split_error;          // pinned host words [kSplitRing]: a launch whose hand-off timed out stores its epoch into word epoch % kSplitRing
    unsigned long long *dbg_stamps; // tools only (NMI_OPT_STAMPS): [workgroup][8] wall_clock64 stamps at phase boundaries
    uint32_t *dbg_joint, *dbg_h1, *dbg_h2;
    float *dbg_sums;
    int hist_variant;             // 0 per-pixel wrap test, 1 batched, 2 unchecked (ablation), 3 optimistic + verify (default), 4 pipelined (experimental)
    int phase_mask;               // bit 0 histogram phase, bit 1 decode + score (ablation; product uses 3)
};

// Image geometry of a launch (shared by the search and the level-graph entry points).
inline void set_geometry(GridArgs &a, int width, int height, const void *render_stack, const void *warp_stack, bool render_bottom_up)
{
    a.width = width;
    a.height = height;
    a.npix = width * height;
    a.vec_ok = (width % 16 == 0) && width >= 32 && (((uintptr_t)render_stack | (uintptr_t)warp_stack) % 16 == 0);
    // widths that are not multiples of 16 (KITTI's 1241) and unaligned stacks: nmi_grid_kernel_rows takes floor(width / 16)
    // 16-byte chunks of every row with unaligned loads and the rows' last width % 16 pixels one by one (launch_grid)
    a.chunks_per_row = width >= 32 ? width / 16 : 1;
    a.cpr_magic = a.chunks_per_row > 1 ? (uint32_t)((0x100000000ull + a.chunks_per_row - 1) / a.chunks_per_row) : 0u;
    a.flip = render_bottom_up ? 1 : 0;
    a.flip_base = render_bottom_up ? (height - 1) * a.chunks_per_row : 0;
    a.flip_row = render_bottom_up ? -2 * a.chunks_per_row : 0;
}

hipError_t launch_table(float *table, int npix, hipStream_t stream);
hipError_t launch_grid(const GridArgs &a, int workgroups, bool use_bg, hipStream_t stream);
// The same kernel as the fallback behind launch_fewlevels: it returns at once unless a.plan->use == 0.  Fewer than 256 bins
// only with the background rule on.
hipError_t launch_grid_gated(const GridArgs &a, int workgroups, bool use_bg, hipStream_t stream);
// nmi_grid_kernel for frames whose rows are not whole aligned 16-byte chunks (width % 16 != 0, or stacks that are not 16-byte
// aligned; width >= 32): nmi_kernels_rows.hip.  launch_grid sends such launches there itself.
hipError_t launch_grid_rows(const GridArgs &a, int workgroups, bool use_bg, hipStream_t stream);
// tools only (NMI_OPT_STAMPS): nmi_grid_kernel with wall-clock stamps at its phase boundaries (nmi_kernels_stamped.hip)
hipError_t launch_grid_stamped(const GridArgs &a, int workgroups, hipStream_t stream);

// Few-levels path (nmi_fewlevels_kernel.hip).  launch_levels probes the stacks (16-byte aligned, npix % 16 == 0) and
// writes *plan (use = commit && nr * nw <= max_joint) and, if given, the pinned word *post = seq << 32 | nr << 16 | nw.
// launch_fewlevels = rank images + scoring kernel; both do nothing when plan->use == 0.
hipError_t launch_levels(const uint8_t *render_stack, int S, const uint8_t *warp_stack, int Wn, int npix, int shift, LevelPlan *plan,
                         unsigned long long *post, uint32_t seq, uint32_t max_joint, bool commit, hipStream_t stream);
hipError_t launch_fewlevels(const GridArgs &a, uint8_t *rank_renders, uint8_t *rank_warps, int workgroups, bool use_bg,
                            hipStream_t stream);
int fewlevels_max_joint();  // largest nr * nw the scoring kernel takes
bool ablation_variants_built();  // HIST variants 0 / 2 / 4 compiled in (-DNMI_BUILD_ABLATIONS)?

// Split form for grids with fewer candidates than compute units (nmi_split_kernel.hip): K workgroups per candidate, each
// owning 256 / K complete rows (render intensities) of the joint histogram as 32-bit LDS counters.  parts = 2, 4 or 8.
// Everything that crosses workgroups travels as 8-byte {payload, epoch} granules, each written by ONE sc1 store, so a
// reader that finds the launch's epoch in a granule has the payload too (no flag, no release / acquire pair).
struct SplitSlab {
    unsigned long long row_sums[256];     // {float bits of the joint-row entropy sum (d_JointEntropyShort, kernel.cu:60,90), epoch}
    unsigned long long hist_render[256];  // {render marginal count = row sum of the counts, epoch}, each row by its owner
    unsigned long long hw_part[8][256];   // {frame marginal: column sum over the rows of one part, epoch}
};
// pix_parts (1, 2 or 4; > 1 only with parts = 8): the pixels of the pair are additionally cut into that many ranges, one
// workgroup per (row part, pixel range); the workgroups of a row part merge their counters through `blocks`.
hipError_t launch_split(const GridArgs &a, int parts, int pix_parts, int workgroups, bool use_bg, hipStream_t stream);
__host__ __device__ int split_workgroups(int candidates, int parts, int pix_parts);  // grid size of a split launch (one unit per workgroup)
inline size_t split_block_bytes_per_candidate(int pix_parts) { return (size_t)pix_parts * 256 * 128 * sizeof(unsigned long long); }
// Pixel-range form for mid-size grids (nmi_pix_kernel.hip): pix_parts workgroups per candidate, each adding a range of the
// pair's pixels into a whole packed joint histogram; the helpers' histograms travel to the candidate's owner through
// a.blocks (pix_block_bytes, zero when allocated; tag from a.epoch + *replay).  *timeouts counts candidates whose owner
// gave up waiting and scored them alone (the launch heals itself).  Needs a.vec_ok, a.order == nullptr, a.epoch != 0.
// owner_share: fraction of the pair's pixels the owner adds itself (the helpers share the rest equally).
hipError_t launch_pix(const GridArgs &a, int pix_parts, double owner_share, bool use_bg, const uint32_t *replay, uint32_t *timeouts, hipStream_t stream);
size_t pix_block_bytes(int candidates, int pix_parts);
int pix_max_ranges();
int grid_kernel_lds_bytes();
size_t grid_kernel_scratch_bytes(int workgroups);
hipError_t launch_warp(const uint8_t *frame, const float *coeffs /*[Wn][9] inverse maps*/, uint8_t *out, int width,
                       int height, int Wn, hipStream_t stream);

// Textured-mesh renderer (nmi_mesh.hip): a binned, deferred rasteriser.  MeshWork = its device buffers, owned by a context
// or by a level; zbuf must be all ones and state all zero before a render (launch_mesh_clear once after allocation) and
// the renderer leaves them so.  bin_cap_limit: usable entries per bin (NMI_OPT_TILE_QUEUE; 0 = every triangle is
// rasterised by its own lane); clip_cap_limit likewise for the queue of triangles crossing the near plane.
struct MeshWork {
    unsigned long long *zbuf = nullptr;        // [S][H][W] visibility keys of the direct path (mesh_zbuf_bytes)
    uint32_t *bins = nullptr;                  // [S][tiles][stride] bin entries (mesh_bins_bytes)
    uint32_t *state = nullptr;                 // [S][tiles][2] entries appended / tile has keys in zbuf (mesh_state_bytes)
    void *clip_queue = nullptr;                // (triangle, view) pairs that cross the near plane
    unsigned long long clip_cap = 0;
    unsigned long long *clip_state = nullptr;  // [2]
    uint32_t *pairs = nullptr;                 // optional: (block of 256 triangles, view) pairs in reach of each other, listed per render
    uint32_t *pair_state = nullptr;            //   [1] entries listed (zero between renders)
    unsigned long long pairs_cap = 0;          //   entries `pairs` holds; the two-kernel binning pass needs 64 per block of the mesh
    int compute_units = 0;
};
size_t mesh_pairs_entries(long long ntri);     // what pairs_cap must be for a mesh of ntri triangles
size_t mesh_zbuf_bytes(int S, int width, int height);
size_t mesh_bins_bytes(int S, int width, int height);
size_t mesh_state_bytes(int S, int width, int height);
size_t mesh_clip_item_bytes();
hipError_t launch_mesh_clear(const MeshWork &w, int S, int width, int height, hipStream_t stream);
hipError_t launch_render_mesh(const float *xyz, const float *uv, long long ntri, const float *luma, int levels, const int *lw,
                              const int *lh, const long long *loff, const float *mvps /*[S][16]*/, int S, const MeshWork &w,
                              int layout_views /* views the work area was allocated for (>= S) */, int bin_cap_limit,
                              unsigned long long clip_cap_limit, uint8_t *out, int width, int height, hipStream_t stream,
                              // a captured level: the Wn warps of `warp_frame` are made by extra workgroups of the first kernel
                              // (level_front_eligible must hold; needs at least one triangle and S <= 64)
                              const uint8_t *warp_frame = nullptr, const float *warp_coeffs = nullptr, uint8_t *warp_out = nullptr, int Wn = 0);
// Words between rows of the renderers' anchor / depth buffer: the padded width, rounded so that the resolve pass can
// read 8 consecutive anchors of any output quad with two aligned 16-byte loads (point sizes > 1); width for size 1.
inline int zbuf_stride(int width, int size) { return size > 1 ? ((width + size - 1 + 3) & ~3) + 4 : width; }
size_t render_zbuf_words(int S, int width, int height, int size);
hipError_t launch_render_points(const float *xyz, const float *red, long long npoints, const float *mvps /*[S][16] column-major*/,
                                int S, uint32_t *zbuf /*[render_zbuf_words]*/, uint8_t *out, int width, int height, int size, hipStream_t stream,
                                bool clear_first = true);
// nmi_sort.hip: records (na floats each in `a`, the first 3 * verts being vertices; nb floats each in `b`, may be null) into
// Morton order of their positions.  Drains the stream.
hipError_t sort_records_morton(const float *a, int na, int verts, const float *b, int nb, long long n, float *a_out, float *b_out,
                               hipStream_t stream);
// Point-cloud level (nmi_level_create): the cloud packed once -- 16-byte point records + one bounding box per wavefront
// (cloud_pack_bytes of device memory) -- and a front kernel that makes the warp stack and splats in one launch.
size_t cloud_pack_bytes(long long npoints, size_t *boxes_offset);
hipError_t launch_cloud_pack(const float *xyz, const float *red, long long npoints, void *packed, hipStream_t stream);
bool level_front_eligible(const void *frame, const void *warps, int width, int S);
void level_views_bound(const float *mvps /*[S][16] column-major*/, int S, float out[24]);  // host: 6 planes around the S views' frusta
constexpr int kLevelMvpExtra = 28;  // floats behind a level's S matrices: those planes (24), the replay's parity as a word, padding
bool level_points_double_buffered(int width, int size);  // the fused front kernel's form of the anchors: two buffers, cleared in turn
size_t level_zbuf_pair_words(int S, int width, int height, int size);
hipError_t launch_level_front_points(const void *packed, long long npoints, const float *mvps, int S, uint32_t *zbuf /* two buffers */,
                                     const uint32_t *epoch, uint8_t *out, int width, int height, int size, const uint8_t *frame,
                                     const float *coeffs, uint8_t *warps, int Wn, hipStream_t stream,
                                     const uint32_t *kept /* [ceil(n / 64)]: wavefronts in reach of a view, made by the prep kernel */,
                                     const uint32_t *kept_count /* [2], by replay parity */, int compute_units);
hipError_t launch_level_prep(const float *h_mvps, float *d_mvps, int n_mvps, const float *h_coeffs, float *d_coeffs, int n_coeffs,
                             unsigned long long *key, uint32_t *zbuf, size_t nz, hipStream_t stream, uint32_t *epoch = nullptr,
                             const void *packed = nullptr, long long npoints = 0, const float *h_bound /* pinned: 24 floats + parity word */ = nullptr,
                             uint32_t *kept = nullptr, uint32_t *kept_count = nullptr);

}  // namespace nmi
