/*
 * nmi_oracle.c -- CPU restatement of the reference's NMI pose-candidate scoring path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (orbslam2_nmi_amd/, include/) links,
 * loads or calls this file.  It may be used only by tests/, __graft_entry__.smoke() and
 * the cpu_baseline leg of bench.py, and only as the checker / reported CPU baseline.
 *
 * PARITY STATUS: "parity unpinned".  The reference (gsanya/orbslam2_NMI) ships no test,
 * golden vector, known-answer value or fixture for this path, and its
 * only implementation is CUDA 9.2 + CUDA/GL interop + OpenCV-CUDA, which cannot be
 * compiled in this image (no nvcc, no OpenCV, no GL headers), so there is no oracle/_ref.
 * What pins this file instead: analytic known answers (identical images -> SUC 1, constant
 * images -> 0, independent uniform pair -> 0.0200299 at 640x480), exact integer
 * invariants (sum(joint) = W*H with BG on, marginals = row/column sums) and agreement with
 * an independently written numpy twin (oracle/nmi_oracle_np.py).  See tests/test_oracle.py.
 *
 * Every function cites the reference lines it restates (paths relative to the reference
 * repository root).  The arithmetic is deliberately written in the reference's order:
 * exact u32 counts, fp32 per-bin terms, stride-halving fp32 trees.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* fp32 evaluation exactly as written: no FMA contraction, no fast-math (also enforced by the Makefile). */
#pragma STDC FP_CONTRACT OFF

#define NMI_BINS 256
#define NMI_JOINT (NMI_BINS * NMI_BINS)

/* Modes follow Thirdparty/CUDA_Functions/kernel.cuh:22-23 (#define ENMI 0 / #define SUC 1). */
#define NMI_ORACLE_MODE_ENMI 0
#define NMI_ORACLE_MODE_SUC 1

/*
 * Joint + marginal histograms of one (render, warped-frame) pair.
 *
 * Restates histogram256Kernel + addByte_noBG, Thirdparty/CUDA_Functions/NMI.cu:42-49,79-87:
 *   pos runs over all W*H pixels; data1 = render texel (pos % W, H-1 - pos / W)  [:82, the
 *   vertical flip of the bottom-up GL texture]; data2 = d_Warped[pos] [:83]; the pixel is
 *   counted iff nmi_prop_BG || (data1 != 0 && data2 != 0) [:85]; counted pixels increment
 *   hist1[data1], hist2[data2] and joint[data1*256 + data2] [:46-48].
 * The partial-histogram merges (NMI.cu:91-103,110-161) are exact integer sums, so the merged
 * result equals the direct count done here.
 *
 * shift: 0 for the reference's 256 bins; bins = 256 >> shift uses (intensity >> shift)
 *        (BASELINE.json config 1 asks for 64 bins; the reference itself has only 256,
 *        NMI.cuh:39).  The BG test is applied to the raw intensities, as in the reference.
 * render_bottom_up: 1 = reference behaviour (render stored bottom-up); 0 = render stored
 *        top-down like the warped frame.
 */
void nmi_oracle_joint_hist(const uint8_t *render, const uint8_t *warped, int width, int height,
                           int shift, int use_bg, int render_bottom_up, uint32_t *joint /*[65536]*/,
                           uint32_t *hist1 /*[256]*/, uint32_t *hist2 /*[256]*/)
{
    memset(joint, 0, NMI_JOINT * sizeof(uint32_t));
    memset(hist1, 0, NMI_BINS * sizeof(uint32_t));
    memset(hist2, 0, NMI_BINS * sizeof(uint32_t));
    for (int y = 0; y < height; ++y) {
        const uint8_t *wrow = warped + (size_t)y * width;
        const uint8_t *rrow = render + (size_t)(render_bottom_up ? (height - 1 - y) : y) * width;
        for (int x = 0; x < width; ++x) {
            uint32_t d1 = rrow[x];
            uint32_t d2 = wrow[x];
            if (use_bg || (d1 != 0 && d2 != 0)) {
                d1 >>= shift;
                d2 >>= shift;
                hist1[d1]++;
                hist2[d2]++;
                joint[d1 * NMI_BINS + d2]++;
            }
        }
    }
}

/*
 * Per-bin term, ComputeEntropyKernel, NMI.cu:242-263:
 *   0 when the count is 0, else ((float)c / (float)length) * log2f((float)c / (float)length).
 * length is always width*height (kernel.cu:85), also when BG pixels were skipped.
 *
 * log2f is a library function whose last bit differs between implementations: CUDA documents its device log2f as
 * within 1 ulp, glibc's is within 1 ulp, and neither is the reference's build.  Two evaluations are offered:
 *   NMI_ORACLE_TERM_LIBM (0, default)      l = log2f(p) of this host's libm -- the expression as written;
 *   NMI_ORACLE_TERM_ROUNDED (1)            l = (float)log2((double)p): the fp64 logarithm rounded once, i.e. the
 *                                          correctly rounded fp32 log2 (what any 1-ulp log2f approximates, and what
 *                                          the product's per-count table holds, csrc/nmi_kernels.hip nmi_table_kernel).
 * With mode 1 the oracle and the GPU evaluate identical fp32 operations in identical order, so rating tables are
 * compared with ==; mode 0 stays as the <= 1e-5 cross-check of the north_star's tolerance.
 */
#define NMI_ORACLE_TERM_LIBM 0
#define NMI_ORACLE_TERM_ROUNDED 1
static int g_term_mode = NMI_ORACLE_TERM_LIBM;

void nmi_oracle_set_term_mode(int mode) { g_term_mode = mode == NMI_ORACLE_TERM_ROUNDED ? NMI_ORACLE_TERM_ROUNDED : NMI_ORACLE_TERM_LIBM; }
int nmi_oracle_get_term_mode(void) { return g_term_mode; }

float nmi_oracle_bin_term(uint32_t count, int length)
{
    if (count == 0)
        return 0.0f;
    float p = (float)count / (float)length;
    float l = g_term_mode == NMI_ORACLE_TERM_ROUNDED ? (float)log2((double)p) : log2f(p);
    return p * l;
}

/* The whole per-count table term[c], c = 0..length, in the current mode (to compare with the product's table). */
void nmi_oracle_term_table(int length, float *out /*[length + 1]*/)
{
    for (int c = 0; c <= length; ++c) out[c] = nmi_oracle_bin_term((uint32_t)c, length);
}

/*
 * In-place stride-halving tree over 256 floats; the sum ends in a[0].
 * AddvectorParwiseMidKernel (NMI.cu:274-284, one 256-float row per block, n = 128..1) and
 * AddVectorPairwiseKernel (NMI.cu:297-307 etc., n = 128..1) both perform
 *   for n in 128,64,...,1: for t < n: a[t] += a[t + n].
 */
float nmi_oracle_tree256(float *a)
{
    for (int n = NMI_BINS / 2; n >= 1; n /= 2)
        for (int t = 0; t < n; ++t) {
            float s = a[t] + a[t + n];
            a[t] = s;
        }
    return a[0];
}

/*
 * Score from the three (negative) entropy sums, AddVectorPairwiseKernel NMI.cu:342-362.
 * The reference reads the other blocks' results without a grid sync (NMI.cu:340-342); this
 * is the intended value, i.e. computed from the three completed sums.
 */
float nmi_oracle_score(float a1, float a2, float a3, int mode)
{
    if (a1 == 0 && a2 == 0 && a3 == 0)
        return 0.0f;
    if (mode == NMI_ORACLE_MODE_ENMI) {
        float num = (-a1) + (-a2);
        return num / (-a3); /* NMI.cu:349 */
    }
    if (mode == NMI_ORACLE_MODE_SUC) {
        float den = (-a1) + (-a2);
        float q = (-a3) / den;
        float r = 1 - q;
        return 2 * r; /* NMI.cu:357 */
    }
    return -1.0f; /* NMI.cu:361 */
}

/*
 * Histograms -> score.  Launch chain kernel.cu:83-95: per-bin terms for hist1, hist2 and the
 * joint; the joint is reduced row by row (row = fixed d1, AddvectorParwiseMidKernel) into 256
 * row sums, then three trees.  sums[0..2] receive the raw (<= 0) sums A1, A2, A3 if non-NULL.
 */
float nmi_oracle_score_from_hist(const uint32_t *joint, const uint32_t *hist1, const uint32_t *hist2,
                                 int length, int mode, float *sums)
{
    float e1[NMI_BINS], e2[NMI_BINS], je_short[NMI_BINS], row[NMI_BINS];
    for (int b = 0; b < NMI_BINS; ++b) {
        e1[b] = nmi_oracle_bin_term(hist1[b], length);
        e2[b] = nmi_oracle_bin_term(hist2[b], length);
    }
    for (int d1 = 0; d1 < NMI_BINS; ++d1) {
        for (int d2 = 0; d2 < NMI_BINS; ++d2)
            row[d2] = nmi_oracle_bin_term(joint[d1 * NMI_BINS + d2], length);
        je_short[d1] = nmi_oracle_tree256(row);
    }
    float a1 = nmi_oracle_tree256(e1);
    float a2 = nmi_oracle_tree256(e2);
    float a3 = nmi_oracle_tree256(je_short);
    if (sums) {
        sums[0] = a1;
        sums[1] = a2;
        sums[2] = a3;
    }
    return nmi_oracle_score(a1, a2, a3, mode);
}

/* One candidate, CUDAF::NMIWithCuda_noMask, kernel.cu:49-114 (the arithmetic only). */
float nmi_oracle_eval_pair(const uint8_t *render, const uint8_t *warped, int width, int height, int shift,
                           int use_bg, int render_bottom_up, int mode)
{
    uint32_t *joint = (uint32_t *)malloc(NMI_JOINT * sizeof(uint32_t));
    uint32_t h1[NMI_BINS], h2[NMI_BINS];
    nmi_oracle_joint_hist(render, warped, width, height, shift, use_bg, render_bottom_up, joint, h1, h2);
    float s = nmi_oracle_score_from_hist(joint, h1, h2, width * height, mode, NULL);
    free(joint);
    return s;
}

/*
 * fp64 evaluation of the same quantity (no fp32 trees) -- used only to bound the fp32
 * rounding error of the reference's arithmetic in tests.
 */
double nmi_oracle_eval_pair_f64(const uint8_t *render, const uint8_t *warped, int width, int height, int shift,
                                int use_bg, int render_bottom_up, int mode)
{
    uint32_t *joint = (uint32_t *)malloc(NMI_JOINT * sizeof(uint32_t));
    uint32_t h1[NMI_BINS], h2[NMI_BINS];
    nmi_oracle_joint_hist(render, warped, width, height, shift, use_bg, render_bottom_up, joint, h1, h2);
    double len = (double)width * height, a1 = 0, a2 = 0, a3 = 0;
    for (int b = 0; b < NMI_BINS; ++b) {
        if (h1[b]) a1 += (h1[b] / len) * log2(h1[b] / len);
        if (h2[b]) a2 += (h2[b] / len) * log2(h2[b] / len);
    }
    for (int b = 0; b < NMI_JOINT; ++b)
        if (joint[b]) a3 += (joint[b] / len) * log2(joint[b] / len);
    free(joint);
    if (a1 == 0 && a2 == 0 && a3 == 0)
        return 0.0;
    if (mode == NMI_ORACLE_MODE_ENMI)
        return (a1 + a2) / a3;
    return 2.0 * (1.0 - a3 / (a1 + a2));
}

/*
 * Arg-max over the rating table, helperFunctions::find_max_elements,
 * Thirdparty/Localization/helperFunctions.cpp:50-103, with the caller's "[0]" pick
 * (src/Tracking.cc:1952-1953): max starts at 0, strict '>' ; the winner is the first cell
 * (lowest linear index in wz,wy,wx,sz,sy,sx order) equal to the max.  Returns -1 when no
 * cell equals the max (all cells negative or NaN: the reference would index an empty vector).
 */
int64_t nmi_oracle_find_max(const float *ratings, int64_t n, float *best_score)
{
    float max = 0;
    for (int64_t i = 0; i < n; ++i)
        if (ratings[i] > max)
            max = ratings[i];
    for (int64_t i = 0; i < n; ++i)
        if (ratings[i] == max) {
            if (best_score) *best_score = ratings[i];
            return i;
        }
    if (best_score) *best_score = 0;
    return -1;
}

/*
 * The candidate loop of Tracking::RelocalizeWithNMI, src/Tracking.cc:1879-1902, followed by the
 * arg-max (:1905,1952).  ratings[w*S + s] corresponds to rating[wZ][wY][wX][sZ][sY][sX] with
 * w = (wz*nWy+wy)*nWx+wx and s = (sz*nSy+sy)*nSx+sx.
 * threads <= 1: serial like the reference; otherwise OpenMP over candidates (CPU baseline).
 */
int64_t nmi_oracle_search_grid(const uint8_t *render_stack, int S, const uint8_t *warp_stack, int Wn, int width,
                               int height, int shift, int use_bg, int render_bottom_up, int mode, int threads,
                               float *ratings, float *best_score)
{
    size_t npix = (size_t)width * height;
    int64_t total = (int64_t)S * Wn;
    float *r = ratings ? ratings : (float *)malloc(total * sizeof(float));
#ifdef _OPENMP
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic) num_threads(threads)
#endif
    for (int64_t i = 0; i < total; ++i) {
        int w = (int)(i / S), s = (int)(i % S);
        r[i] = nmi_oracle_eval_pair(render_stack + s * npix, warp_stack + w * npix, width, height, shift, use_bg,
                                    render_bottom_up, mode);
    }
    int64_t idx = nmi_oracle_find_max(r, total, best_score);
    if (!ratings) free(r);
    return idx;
}

int nmi_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
