// relocalize_demo.cpp -- C++ host program on the drop-in boundary: what a maintainer of the reference would write.
//
// It drives the whole NMI refinement of one keyframe through the C++/C interface only:
//   * NmiSearchKernel (host/nmi_search_kernel.hpp)  -- the reference's grid descriptor class, same interface
//   * nmi_relocalize_with_strategy (include/nmi_host.h) -- Tracking::RelocalizeWithNMIStrategy (Tracking.cc:1987-2179)
//   * per level: render stack (here: a toy planar "renderer" on the host, uploaded), warp stack produced on the GPU
//     (nmi_warp_homographies + nmi_warp_stack = Image::calculateWarping), nmi_search_grid = the candidate loop + arg-max
//   * CUDAF::NMIWithCuda_noMask through host/cudaf_shim.hpp for one candidate, compared with the grid's rating table.
// The scene is a fronto-parallel textured plane at depth Z, so a camera translation (tx, ty) is a pixel shift of
// f*t/Z and a translation tz a zoom; the "camera frame" is the plane seen from a pose that is off by a known offset.
// The toy renderer only understands translations, so part A refines translation only (warp grid 1x1x1, the identity
// warp still produced on the device); part B is a one-level rotation search: a frame rotated about the optical axis
// against the warp stack of a 1x1x3 grid.  Exit code 0 iff both recover what was planted.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <vector>

#include "cudaf_shim.hpp"
#include "nmi_hip.h"
#include "nmi_host.h"
#include "nmi_rating.hpp"
#include "nmi_search_kernel.hpp"

#define CHECK_HIP(x)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));                 \
            return 2;                                                                      \
        }                                                                                  \
    } while (0)
#define CHECK_NMI(x)                                                                       \
    do {                                                                                   \
        int r_ = (x);                                                                      \
        if (r_ != NMI_OK) {                                                                \
            fprintf(stderr, "%s failed: %d (%s)\n", #x, r_, nmi_error_string(r_));         \
            return 2;                                                                      \
        }                                                                                  \
    } while (0)

namespace {

constexpr int W = 320, H = 240;
constexpr double FX = 145.0, FY = 145.0, CX = 158.5, CY = 121.6, DEPTH = 10.0;  // plane at Z = 10 m

struct Plane {  // smooth texture + seeded noise, larger than the frame so shifted views stay covered
    int w = W + 160, h = H + 160;
    std::vector<float> t;
    Plane() : t((size_t)w * h)
    {
        unsigned s = 12345u;
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                s = s * 1664525u + 1013904223u;
                const float n = ((s >> 8) & 0xFFFF) / 65535.0f - 0.5f;
                t[(size_t)y * w + x] = 128.0f + 45.0f * sinf(0.031f * x + 0.6f * sinf(0.017f * y)) + 40.0f * cosf(0.043f * y + 0.011f * x) +
                                       18.0f * sinf(0.11f * (x + y)) + 12.0f * n;
            }
    }
    float at(float x, float y) const  // bilinear
    {
        x += 80.0f, y += 80.0f;
        if (x < 0 || y < 0 || x >= w - 1 || y >= h - 1) return -1.0f;
        const int x0 = (int)x, y0 = (int)y;
        const float fx = x - x0, fy = y - y0;
        const float *p = &t[(size_t)y0 * w + x0];
        return (1 - fx) * (1 - fy) * p[0] + fx * (1 - fy) * p[1] + (1 - fx) * fy * p[w] + fx * fy * p[w + 1];
    }
};

// View of the plane from a camera displaced by (tx, ty, tz) metres in its own axes (x right, y down, z forward).
void view(const Plane &pl, float tx, float ty, float tz, float gamma, uint8_t background, bool bottom_up, uint8_t *out)
{
    const float z = (float)DEPTH - tz, s = z / (float)DEPTH;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const float u = (float)CX + ((float)x - (float)CX) * s + (float)FX * tx / (float)DEPTH;
            const float v = (float)CY + ((float)y - (float)CY) * s + (float)FY * ty / (float)DEPTH;
            float val = pl.at(u, v);
            uint8_t px = background;
            if (val >= 0.0f) {
                val = 255.0f * powf(fminf(fmaxf(val, 0.0f), 255.0f) / 255.0f, gamma);
                px = (uint8_t)lrintf(fminf(fmaxf(val, 0.0f), 255.0f));
            }
            out[(size_t)(bottom_up ? H - 1 - y : y) * W + x] = px;
        }
}

struct Demo {
    nmi_ctx *ctx = nullptr;
    Plane plane;
    uint8_t *d_frame = nullptr, *d_renders = nullptr, *d_warps = nullptr;
    float *d_ratings = nullptr;
    size_t cap_r = 0, cap_w = 0, cap_t = 0;
    std::vector<uint8_t> h_renders;
    std::vector<float> h_ratings;
    int evals = 0;
};

// The replacement for the body of Tracking::RelocalizeWithNMI (Tracking.cc:1871-1905): stacks for this grid around Twc,
// then one nmi_search_grid.  The toy renderer only understands camera translations; the camera axes are the columns of
// Twc (x right, y down = -up, z forward), matching rendering.hpp:668-694 up to the sign conventions documented there.
int eval_grid(void *user, const nmi_search_kernel *g, const float Twc[16], int64_t *best_index, float *best_score)
{
    Demo &d = *static_cast<Demo *>(user);
    const int S = g->num[0] * g->num[1] * g->num[2], Wn = g->num[3] * g->num[4] * g->num[5];
    const size_t npix = (size_t)W * H;
    if ((size_t)S * npix > d.cap_r) {
        if (d.d_renders) (void)hipFree(d.d_renders);
        if (hipMalloc((void **)&d.d_renders, (size_t)S * npix) != hipSuccess) return -10;
        d.cap_r = (size_t)S * npix;
    }
    if ((size_t)Wn * npix > d.cap_w) {
        if (d.d_warps) (void)hipFree(d.d_warps);
        if (hipMalloc((void **)&d.d_warps, (size_t)Wn * npix) != hipSuccess) return -10;
        d.cap_w = (size_t)Wn * npix;
    }
    if ((size_t)S * Wn > d.cap_t) {
        if (d.d_ratings) (void)hipFree(d.d_ratings);
        if (hipMalloc((void **)&d.d_ratings, (size_t)S * Wn * sizeof(float)) != hipSuccess) return -10;
        d.cap_t = (size_t)S * Wn;
    }
    // render stack: camera at Twc displaced by the translation of each grid cell
    d.h_renders.resize((size_t)S * npix);
    for (int sz = 0; sz < g->num[2]; ++sz)
        for (int sy = 0; sy < g->num[1]; ++sy)
            for (int sx = 0; sx < g->num[0]; ++sx) {
                float t[3];
                nmi_calculate_translation(Twc, g, sx, sy, sz, t);
                // world offset of the cell + world position of the camera, expressed in the identity-oriented toy world
                const float wx = Twc[3] + t[0], wy = Twc[7] + t[1], wz = Twc[11] + t[2];
                view(d.plane, wx, wy, wz, 0.7f, 255, true, &d.h_renders[(size_t)((sz * g->num[1] + sy) * g->num[0] + sx) * npix]);
            }
    if (hipMemcpy(d.d_renders, d.h_renders.data(), (size_t)S * npix, hipMemcpyHostToDevice) != hipSuccess) return -11;
    // warp stack on the device: Image::calculateWarping
    const double K[9] = {FX, 0, CX, 0, FY, CY, 0, 0, 1};
    std::vector<double> M((size_t)Wn * 9);
    const int32_t nw[3] = {g->num[3], g->num[4], g->num[5]};
    const float sw[3] = {g->step[3], g->step[4], g->step[5]};
    int rc = nmi_warp_homographies(K, nw, sw, M.data());
    if (rc != NMI_OK) return rc;
    if ((rc = nmi_warp_stack(d.ctx, d.d_frame, M.data(), Wn, d.d_warps)) != NMI_OK) return rc;
    rc = nmi_search_grid(d.ctx, d.d_renders, S, d.d_warps, Wn, d.d_ratings, best_index, best_score);
    ++d.evals;
    return rc;
}

}  // namespace

int main()
{
    Demo d;
    nmi_params p;
    CHECK_NMI(nmi_params_default(&p, W, H));
    CHECK_NMI(nmi_create(&p, &d.ctx));

    // ground truth: the frame was taken 0.30 m right, 0.18 m up(-y) and 0.35 m forward of where the tracker thinks it is
    const float truth[3] = {0.30f, -0.18f, 0.35f};
    std::vector<uint8_t> frame((size_t)W * H);
    view(d.plane, truth[0], truth[1], truth[2], 1.0f, 0, false, frame.data());
    CHECK_HIP(hipMalloc((void **)&d.d_frame, frame.size()));
    CHECK_HIP(hipMemcpy(d.d_frame, frame.data(), frame.size(), hipMemcpyHostToDevice));

    // the tracker's pose: camera at the origin, axes = identity (Twc = I), 3^6 grid with the steps of ETH_small.yaml:83-88
    NmiSearchKernel initial(3, 3, 3, 1, 1, 1, 0.2f, 0.2f, 0.5f, 0.02f, 0.02f, 0.05f);
    nmi_strategy_input in;
    memset(&in, 0, sizeof in);
    for (int i = 0; i < 4; ++i) in.Tcw[i * 5] = 1.0f;
    in.nmi_threshold = 0.1f;
    in.initial = initial.to_c();
    nmi_strategy_output out;
    nmi_properties props;
    nmi_properties_default(&props);
    CHECK_NMI(nmi_relocalize_with_strategy(&in, &props, eval_grid, &d, &out));

    for (int i = 0; i < out.iterations; ++i) std::cout << "NmiKernel:\t" << NmiSearchKernel(out.per_iteration[i]) << "\n";
    float Twc[16];
    nmi_mat4_inverse(out.Tcw, Twc);
    printf("relocalized=%d failed=%d iterations=%d stop=%d  recovered t = (%.3f, %.3f, %.3f)  truth (%.3f, %.3f, %.3f)  NMI %.5f\n",
           out.relocalized, out.failed, out.iterations, out.stop_reason, Twc[3], Twc[7], Twc[11], truth[0], truth[1], truth[2],
           out.kernel.nmi);

    bool ok = out.relocalized && !out.failed;
    ok = ok && fabsf(Twc[3] - truth[0]) <= 0.1f && fabsf(Twc[7] - truth[1]) <= 0.1f && fabsf(Twc[11] - truth[2]) <= 0.25f;
    ok = ok && out.iterations >= 2 && out.kernel.nmi > out.per_iteration[0].nmi;  // refinement improved the score

    // ---- part B: one rotation level.  The frame is rotated by -0.05 rad about the optical axis (made on the device with
    // the warp producer itself); against a 1x1x3 warp grid with step 0.05 the winner must be the cell that undoes it.
    NmiSearchKernel rotgrid(1, 1, 1, 1, 1, 3, 0.2f, 0.2f, 0.5f, 0.02f, 0.02f, 0.05f);
    const double K[9] = {FX, 0, CX, 0, FY, CY, 0, 0, 1};
    const int32_t one3[3] = {1, 1, 3};
    const float st3[3] = {0.02f, 0.02f, 0.05f};
    double M3[27];
    CHECK_NMI(nmi_warp_homographies(K, one3, st3, M3));  // cells: rz = -0.05, 0, +0.05
    uint8_t *d_rot = nullptr;
    CHECK_HIP(hipMalloc((void **)&d_rot, (size_t)W * H));
    CHECK_NMI(nmi_warp_stack(d.ctx, d.d_frame, M3 + 0, 1, d_rot));  // frame rotated by -0.05
    CHECK_NMI(nmi_synchronize(d.ctx));
    uint8_t *keep = d.d_frame;
    d.d_frame = d_rot;
    nmi_search_kernel rg = rotgrid.to_c();
    float T[16] = {1, 0, 0, truth[0], 0, 1, 0, truth[1], 0, 0, 1, truth[2], 0, 0, 0, 1};  // camera at the true position
    int64_t bi = -1;
    float bs = 0;
    if (eval_grid(&d, &rg, T, &bi, &bs) != 0) return 2;
    d.d_frame = keep;
    std::vector<float> table(3);
    CHECK_HIP(hipMemcpy(table.data(), d.d_ratings, 3 * sizeof(float), hipMemcpyDeviceToHost));
    printf("rotation level: ratings rz=-0.05: %.5f  rz=0: %.5f  rz=+0.05: %.5f  -> best cell %lld\n", table[0], table[1], table[2],
           (long long)bi);
    ok = ok && bi == 2 && table[2] > 1.5f * table[1] && table[2] > 1.5f * table[0];

    // one candidate through the reference's own entry point (shim): must equal the grid's rating for that cell
    CUDAF::RegisterRenderBuffer(42u, d.d_renders);
    float nmi_shim = -1.0f;
    CUDAF::NMIWithCuda_noMask((cv::cuda::PtrStep<unsigned char> *)(d.d_warps + (size_t)2 * W * H), SUC, MATCHING_NMI, W, H, &nmi_shim,
                              42u);
    printf("shim NMIWithCuda_noMask(render 0, warp 2) = %.7f, grid rating = %.7f\n", nmi_shim, table[2]);
    ok = ok && nmi_shim == table[2];

    // ---- part C: throughput of the UNCHANGED per-candidate call site (src/Tracking.cc:1886-1894) at the north-star frame
    // size: 640x480, one blocking CUDAF::NMIWithCuda_noMask per candidate, score on the host after every call.
    {
        const int w = 640, h = 480, n_r = 3, n_w = 9;
        std::vector<uint8_t> img((size_t)w * h * (n_r + n_w));
        unsigned s = 777u;
        for (size_t k = 0; k < img.size(); ++k) {
            s = s * 1664525u + 1013904223u;
            const size_t pix = k % ((size_t)w * h);
            const float base = 128.0f + 60.0f * sinf(0.02f * (float)(pix % w)) * cosf(0.015f * (float)(pix / w));
            img[k] = (uint8_t)fminf(fmaxf(base + (float)((s >> 16) & 31) - 16.0f + 3.0f * (float)(k / ((size_t)w * h)), 0.0f), 255.0f);
        }
        uint8_t *d_img = nullptr;
        CHECK_HIP(hipMalloc((void **)&d_img, img.size()));
        CHECK_HIP(hipMemcpy(d_img, img.data(), img.size(), hipMemcpyHostToDevice));
        for (int r = 0; r < n_r; ++r) CUDAF::RegisterRenderBuffer(100u + r, d_img + (size_t)r * w * h);
        std::vector<float> first(n_r * n_w), again(n_r * n_w);
        auto pass = [&](std::vector<float> &outv) {
            for (int r = 0; r < n_r; ++r)
                for (int v = 0; v < n_w; ++v)
                    CUDAF::NMIWithCuda_noMask((cv::cuda::PtrStep<unsigned char> *)(d_img + (size_t)(n_r + v) * w * h), SUC, MATCHING_NMI, w, h,
                                              &outv[r * n_w + v], 100u + r);
        };
        pass(first);  // warm-up (context creation for this frame size)
        const int reps = 150;
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < reps; ++k) pass(again);
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const double rate = (double)reps * n_r * n_w / dt;
        bool same = true, distinct = false;
        for (size_t k = 0; k < first.size(); ++k) same = same && first[k] == again[k] && first[k] > 0.0f && first[k] < 1.0f;
        for (size_t k = 1; k < first.size(); ++k) distinct = distinct || first[k] != first[0];
        printf("shim call site: %d blocking NMIWithCuda_noMask calls at %dx%d: %.1f us per call = %.0f evals/s (target 50000)\n",
               reps * n_r * n_w, w, h, dt / (reps * n_r * n_w) * 1e6, rate);
        printf("SHIM_EVALS_PER_S %.0f\n", rate);
        ok = ok && same && distinct;
        // The two lines BEHIND the call in the reference's loop, unchanged as well (src/Tracking.cc:1895, :1905): the score goes
        // into the six-level rating table, helperFunctions::find_max_elements picks the winner (host/nmi_rating.hpp: the table is
        // a view over the flat array nmi_search_grid writes).  Same winner as one nmi_search_grid call over the same stacks.
        {
            NmiSearchKernel kern(n_r, 1, 1, 3, 3, 1, 0.2f, 0.2f, 0.5f, 0.02f, 0.02f, 0.05f);
            NmiRatingTable rating(kern);
            for (int sX = 0; sX < kern.getNumSynthX(); sX++)
                for (int wX = 0; wX < kern.getNumWarpX(); wX++)
                    for (int wY = 0; wY < kern.getNumWarpY(); wY++) {
                        const int sY = 0, sZ = 0, wZ = 0, v = wY * kern.getNumWarpX() + wX;
                        float nmi = -1.0f;
                        CUDAF::NMIWithCuda_noMask((cv::cuda::PtrStep<unsigned char> *)(d_img + (size_t)(n_r + v) * w * h), SUC, MATCHING_NMI, w, h,
                                                  &nmi, 100u + sX);
                        rating[wZ][wY][wX][sZ][sY][sX] = nmi;
                    }
            std::vector<NmiSearchKernel> maxElements = helperFunctions::find_max_elements(rating, kern);
            nmi_params p640;
            CHECK_NMI(nmi_params_default(&p640, w, h));
            nmi_ctx *c640 = nullptr;
            CHECK_NMI(nmi_create(&p640, &c640));
            int64_t gi = -1;
            float gs = 0;
            float *d_tab = nullptr;
            CHECK_HIP(hipMalloc((void **)&d_tab, (size_t)n_r * n_w * sizeof(float)));
            CHECK_NMI(nmi_search_grid(c640, d_img, n_r, d_img + (size_t)n_r * w * h, n_w, d_tab, &gi, &gs));
            std::vector<float> grid_tab((size_t)n_r * n_w);
            CHECK_HIP(hipMemcpy(grid_tab.data(), d_tab, grid_tab.size() * sizeof(float), hipMemcpyDeviceToHost));
            bool same_table = true;
            for (size_t k = 0; k < grid_tab.size(); ++k) same_table = same_table && grid_tab[k] == rating.flat()[k];
            const bool same_winner = maxElements.size() >= 1 && maxElements[0].getNmi() == gs &&
                                     (int64_t)(maxElements[0].getBestWarpY() * 3 + maxElements[0].getBestWarpX()) * n_r + maxElements[0].getBestSynthX() == gi;
            printf("rating[wZ][wY][wX][sZ][sY][sX] + helperFunctions::find_max_elements: %zu winner(s), first = synth %d warp (%d, %d) NMI %.7f; "
                   "nmi_search_grid: index %lld NMI %.7f; tables %s\n", maxElements.size(), maxElements.empty() ? -1 : maxElements[0].getBestSynthX(),
                   maxElements.empty() ? -1 : maxElements[0].getBestWarpX(), maxElements.empty() ? -1 : maxElements[0].getBestWarpY(),
                   maxElements.empty() ? 0.0f : maxElements[0].getNmi(), (long long)gi, gs, same_table ? "equal" : "DIFFER");
            ok = ok && same_winner && same_table;
            (void)hipFree(d_tab);
            nmi_destroy(c640);
        }
        // the same loop with the two extra lines a host may add: BeginBatch() before the warp loop, Flush() after it
        std::vector<float> batched(n_r * n_w, -1.0f);
        auto pass_batched = [&](std::vector<float> &outv) {
            for (int r = 0; r < n_r; ++r) {
                CUDAF::BeginBatch();
                for (int v = 0; v < n_w; ++v)
                    CUDAF::NMIWithCuda_noMask((cv::cuda::PtrStep<unsigned char> *)(d_img + (size_t)(n_r + v) * w * h), SUC, MATCHING_NMI, w, h,
                                              &outv[r * n_w + v], 100u + r);
                CUDAF::Flush();
            }
        };
        pass_batched(batched);
        const auto t1 = std::chrono::steady_clock::now();
        for (int k = 0; k < reps; ++k) pass_batched(batched);
        const double dtb = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
        bool same_b = true;
        for (size_t k = 0; k < first.size(); ++k) same_b = same_b && first[k] == batched[k];
        printf("shim call site with BeginBatch / Flush around the %d-warp loop: %.1f us per candidate = %.0f evals/s\n", n_w,
               dtb / (reps * n_r * n_w) * 1e6, (double)reps * n_r * n_w / dtb);
        printf("SHIM_BATCHED_EVALS_PER_S %.0f\n", (double)reps * n_r * n_w / dtb);
        ok = ok && same_b;
        (void)hipFree(d_img);
    }
    (void)hipFree(d_rot);
    CUDAF::Shutdown();
    (void)hipFree(d.d_frame);
    (void)hipFree(d.d_renders);
    (void)hipFree(d.d_warps);
    (void)hipFree(d.d_ratings);
    nmi_destroy(d.ctx);
    printf("%s\n", ok ? "DEMO OK" : "DEMO FAILED");
    return ok ? 0 : 1;
}
