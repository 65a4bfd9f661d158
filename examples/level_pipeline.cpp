// level_pipeline.cpp -- C++ host program: keyframe refinement with EVERYTHING on the device.
//
// The map is a coloured point cloud (nmi_prop_RENDER 4) or, with --mesh, a textured triangle mesh (nmi_prop_RENDER 1, the
// reference's default, allProperties.hpp:41) in HBM, the camera frame is in HBM; per iteration of
// Tracking::RelocalizeWithNMIStrategy (Tracking.cc:1987-2179, here nmi_relocalize_with_strategy) the host computes 27 view
// matrices (Rendering::calculateTranslationCV + the MVP of rendering.hpp:196-202) and 27 homographies (image.cpp:76-107)
// and replays ONE captured HIP graph (nmi_level_run): renders, warps, the 729-candidate search, winner back.  Only
// matrices and the 8-byte winner cross PCIe.  Prints what was recovered and the rate in keyframes/s and levels/s --
// the native counterpart of `bench.py --config e2e`, without the Python between two levels.
// Exit code 0 iff the planted camera offset is recovered (lateral offset as seen in the image within 8 cm, depth within 20 cm).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "nmi_hip.h"
#include "nmi_host.h"
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

constexpr int W = 848, H = 480;  // Newer-College-shaped frames (BASELINE.json configs[3])
// Examples/Monocular/ETH_small.yaml:8-11 calibration (960x540) scaled to the frame
constexpr double FX = 435.04593205 * W / 960.0, FY = 435.04593205 * H / 540.0, CX = 475.55781765 * W / 960.0, CY = 274.7487729 * H / 540.0;
constexpr float DEPTH = 10.0f;

struct Pipeline {
    nmi_level *level = nullptr;
    nmi_render_params rp{};
    double in_run_s = 0.0;  // time spent inside nmi_level_run (parameters in -> winner out)
    int levels_run = 0;
};

// Replacement for the body of Tracking::RelocalizeWithNMI (Tracking.cc:1871-1905) with the producers on the device.
int eval_level(void *user, const nmi_search_kernel *g, const float Twc[16], int64_t *best_index, float *best_score)
{
    Pipeline &p = *static_cast<Pipeline *>(user);
    if (g->num[0] != 3 || g->num[1] != 3 || g->num[2] != 3 || g->num[3] != 3 || g->num[4] != 3 || g->num[5] != 3) return -20;
    // setupCam (ioData.cpp:177-197): position, a point ahead, the up vector, from the columns of Twc
    const float pos[3] = {Twc[3], Twc[7], Twc[11]};
    const float look[3] = {pos[0] + Twc[2], pos[1] + Twc[6], pos[2] + Twc[10]};
    const float up[3] = {Twc[1], Twc[5], Twc[9]};
    float mvps[27 * 16];
    for (int sz = 0; sz < 3; ++sz)
        for (int sy = 0; sy < 3; ++sy)
            for (int sx = 0; sx < 3; ++sx) {
                float t[3];
                int rc = nmi_calculate_translation(Twc, g, sx, sy, sz, t);
                if (rc != NMI_OK) return rc;
                if ((rc = nmi_render_mvp(&p.rp, pos, look, up, t, mvps + 16 * ((sz * 3 + sy) * 3 + sx))) != NMI_OK) return rc;
            }
    const double K[9] = {FX, 0, CX, 0, FY, CY, 0, 0, 1};
    const int32_t nw[3] = {3, 3, 3};
    const float sw[3] = {g->step[3], g->step[4], g->step[5]};
    double M[27 * 9];
    int rc = nmi_warp_homographies(K, nw, sw, M);
    if (rc != NMI_OK) return rc;
    ++p.levels_run;
    const auto t0 = std::chrono::steady_clock::now();
    rc = nmi_level_run(p.level, mvps, M, best_index, best_score);
    p.in_run_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

}  // namespace

// The map's relief and texture (shared by the point cloud and the mesh form).
static float surface_z(float u, float v) { return DEPTH + 3.0f * sinf(0.012f * u) * cosf(0.015f * v); }
static float surface_grey(float u, float v, float n)
{
    const float t = 128.0f + 45.0f * sinf(0.031f * u + 0.6f * sinf(0.017f * v)) + 40.0f * cosf(0.043f * v + 0.011f * u) +
                    18.0f * sinf(0.11f * (u + v)) + 10.0f * n;
    return fminf(fmaxf(t, 0.0f), 255.0f);
}

// --write-files DIR: the synthetic map as the files the reference's loaders read (OBJ + BMP, or XYZ + offset) and a settings file
// in the reference's YAML format that names them.  Numbers are printed with enough digits to come back bit for bit.
static bool write_files(const char *dir, bool mesh, int mesh_nx, int mesh_ny, const std::vector<float> &xyz, const std::vector<float> &attr,
                        const std::vector<uint8_t> &rgb, int tw, int th)
{
    const std::string d = std::string(dir) + "/";
    FILE *y = fopen((d + "settings.yaml").c_str(), "w");
    if (!y) return false;
    fprintf(y, "%%YAML:1.0\n\nCamera.fx: %.17g\nCamera.fy: %.17g\nCamera.cx: %.17g\nCamera.cy: %.17g\nCamera.Width: %d\nCamera.Height: %d\n\n", FX, FY, CX,
            CY, W, H);
    fprintf(y, "NMI.Treshold: 0.05\nNMI.SynthNumX: 3\nNMI.SynthNumY: 3\nNMI.SynthNumZ: 3\nNMI.WarpNumX: 3\nNMI.WarpNumY: 3\nNMI.WarpNumZ: 3\n");
    fprintf(y, "NMI.SynthStepX: 0.2\nNMI.SynthStepY: 0.2\nNMI.SynthStepZ: 0.5\nNMI.WarpStepX: 0.02\nNMI.WarpStepY: 0.02\nNMI.WarpStepZ: 0.05\n\n");
    fprintf(y, "NMI.Render.PointSize: 3.0\nNMI.Render.NearPlane: 5.0\nNMI.Render.FarPlane: 30.0\n");
    if (mesh)
        fprintf(y, "NMI.Render.Object: \"map.obj\"\nNMI.Render.Texture: \"map.bmp\"\n");
    else
        fprintf(y, "NMI.Render.Cloud: \"map.xyz\"\nNMI.Render.Offset: \"map.offset\"\n");
    fclose(y);
    if (mesh) {
        FILE *o = fopen((d + "map.obj").c_str(), "w");
        if (!o) return false;
        fprintf(o, "# %d x %d quads of the level_pipeline surface\n", mesh_nx, mesh_ny);
        // one v / vt per grid node, taken from the first corner that uses it; faces index them (1-based), as an exporter would write
        std::vector<int> node((size_t)(mesh_nx + 1) * (mesh_ny + 1), -1);
        std::vector<int> index(attr.size() / 2);
        int count = 0;
        for (size_t k = 0; k < attr.size() / 2; ++k) {
            const int i = (int)lrintf(attr[2 * k] * mesh_nx), j = (int)lrintf(attr[2 * k + 1] * mesh_ny);
            int &n = node[(size_t)j * (mesh_nx + 1) + i];
            if (n < 0) {
                n = ++count;
                fprintf(o, "v %.9g %.9g %.9g\nvt %.9g %.9g\n", xyz[3 * k], xyz[3 * k + 1], xyz[3 * k + 2], attr[2 * k], attr[2 * k + 1]);
            }
            index[k] = n;
        }
        for (size_t k = 0; k + 2 < index.size(); k += 3) fprintf(o, "f %d/%d %d/%d %d/%d\n", index[k], index[k], index[k + 1], index[k + 1], index[k + 2], index[k + 2]);
        fclose(o);
        FILE *b = fopen((d + "map.bmp").c_str(), "wb");
        if (!b) return false;
        unsigned char h[54] = {'B', 'M'};
        auto put = [&](int at, uint32_t v) { h[at] = v & 255, h[at + 1] = (v >> 8) & 255, h[at + 2] = (v >> 16) & 255, h[at + 3] = (v >> 24) & 255; };
        put(0x02, 54 + (uint32_t)rgb.size()), put(0x0A, 54), put(0x0E, 40), put(0x12, (uint32_t)tw), put(0x16, (uint32_t)th), put(0x1A, 1 | (24u << 16));
        put(0x22, (uint32_t)rgb.size());
        fwrite(h, 1, 54, b), fwrite(rgb.data(), 1, rgb.size(), b);
        fclose(b);
    } else {
        const double off[3] = {2683000.25, 1250000.5, 400.0};  // (a survey frame's size of numbers: what the offset file is for)
        FILE *f = fopen((d + "map.offset").c_str(), "w");
        if (!f) return false;
        fprintf(f, "%.17g %.17g %.17g\n", off[0], off[1], off[2]);
        fclose(f);
        f = fopen((d + "map.xyz").c_str(), "w");
        if (!f) return false;
        for (size_t k = 0; k < attr.size(); ++k)
            fprintf(f, "%.17g %.17g %.17g %.9g 0 0\n", (double)xyz[3 * k] + off[0], (double)xyz[3 * k + 1] + off[1], (double)xyz[3 * k + 2] + off[2],
                    attr[k] * 256.0f);
        fclose(f);
    }
    return true;
}

int main(int argc, char **argv)
{
    // usage: level_pipeline [keyframes] [--mesh [NXxNY]] [--density D] [--write-files DIR | --files DIR]
    //   --mesh: nmi_prop_RENDER 1, the reference's default render mode: the same surface as NX x NY quads = 2 NX NY textured
    //           triangles, default 300x200 = 120,000;  --density: points per pixel of a view along each axis (cloud; default 0.9)
    //   --write-files DIR: write the map and a settings file into DIR and stop (no GPU needed)
    //   --files DIR: take camera, grid, render parameters and the map from DIR/settings.yaml and the files it names
    //           (nmi_config_load, nmi_map_load_obj / _bmp / _xyz) instead of building them in memory
    int keyframes = 200, mesh_nx = 0, mesh_ny = 0;
    float density = 0.9f;
    const char *write_dir = nullptr, *read_dir = nullptr;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--mesh")) {
            mesh_nx = 300, mesh_ny = 200;
            if (i + 1 < argc && sscanf(argv[i + 1], "%dx%d", &mesh_nx, &mesh_ny) == 2) ++i;
        } else if (!strcmp(argv[i], "--density") && i + 1 < argc) {
            density = (float)atof(argv[++i]);
        } else if (!strcmp(argv[i], "--write-files") && i + 1 < argc) {
            write_dir = argv[++i];
        } else if (!strcmp(argv[i], "--files") && i + 1 < argc) {
            read_dir = argv[++i];
        } else {
            keyframes = atoi(argv[i]);
        }
    }
    bool mesh = mesh_nx > 0 && mesh_ny > 0;

    // The map: a textured, undulating surface about 10 m ahead, three frame-widths wide, ~0.8 points per pixel of a view
    // (3 M points).
    // (relief of +-3 m: on a flat wall a sideways step and a small turn of the camera move the image alike, and the 6-D search
    // could not tell them apart)
    std::vector<float> xyz, red;  // point cloud: positions [N][3] + red [N]; mesh: corners [3T][3] + uv [3T][2]
    std::vector<uint8_t> rgb;     // mesh: the texture, RGB8
    int tw = 0, th = 0;
    nmi_config cfg;
    memset(&cfg, 0, sizeof cfg);
    unsigned s = 2468u;
    if (read_dir) {
        const std::string d = std::string(read_dir) + "/";
        const int rc = nmi_config_load((d + "settings.yaml").c_str(), &cfg);
        if (rc != 0 || cfg.width != W || cfg.height != H) {
            fprintf(stderr, "settings.yaml: rc %d, %d x %d (this program is built for %d x %d)\n", rc, cfg.width, cfg.height, W, H);
            return 1;
        }
        auto at = [&](const char *name) { return name[0] == '/' ? std::string(name) : d + name; };
        mesh = cfg.render_object[0] != 0;
        float *a = nullptr, *b = nullptr;
        int64_t n = 0;
        if (mesh) {
            uint8_t *img = nullptr;
            int32_t w32 = 0, h32 = 0;
            if (nmi_map_load_obj(at(cfg.render_object).c_str(), &a, &b, &n) != 0 || nmi_map_load_bmp(at(cfg.render_texture).c_str(), &img, &w32, &h32) != 0) {
                fprintf(stderr, "cannot read the mesh / its texture\n");
                return 1;
            }
            xyz.assign(a, a + n * 3), red.assign(b, b + n * 2), rgb.assign(img, img + (size_t)w32 * h32 * 3);
            tw = w32, th = h32, mesh_nx = 0, mesh_ny = 0;
            nmi_map_free(img);
        } else {
            if (nmi_map_load_xyz(at(cfg.render_cloud).c_str(), at(cfg.render_offset).c_str(), &a, &b, nullptr, &n) != 0) {
                fprintf(stderr, "cannot read the cloud / its offset\n");
                return 1;
            }
            xyz.assign(a, a + n * 3), red.assign(b, b + n);
        }
        nmi_map_free(a), nmi_map_free(b);
        printf("map from %s: %lld %s\n", read_dir, (long long)(mesh ? n / 3 : n), mesh ? "triangles" : "points");
    } else if (!mesh) {
        const int nu = (int)(3 * W * density), nv = (int)(3 * H * density);
        xyz.resize((size_t)nu * nv * 3), red.resize((size_t)nu * nv);
        for (int j = 0; j < nv; ++j)
            for (int i = 0; i < nu; ++i) {
                const float u = -W + 3.0f * W * i / (nu - 1), v = -H + 3.0f * H * j / (nv - 1);
                const size_t k = (size_t)j * nu + i;
                const float z = surface_z(u, v);
                xyz[3 * k] = (u - (float)CX) / (float)FX * z;
                xyz[3 * k + 1] = (v - (float)CY) / (float)FY * z;
                xyz[3 * k + 2] = z;
                s = s * 1664525u + 1013904223u;
                red[k] = surface_grey(u, v, ((s >> 8) & 0xFFFF) / 65535.0f - 0.5f) / 256.0f;  // objloader.cpp:261: colour / 256
            }
    } else {
        // texture: the surface's grey values on a 2048 x 1024 raster (what loadBMP_custom would hand to glTexImage2D, texture.cpp:31-86)
        tw = 2048, th = 1024;
        rgb.resize((size_t)tw * th * 3);
        for (int j = 0; j < th; ++j)
            for (int i = 0; i < tw; ++i) {
                const float u = -W + 3.0f * W * (i + 0.5f) / tw, v = -H + 3.0f * H * (j + 0.5f) / th;
                s = s * 1664525u + 1013904223u;
                const uint8_t g = (uint8_t)lrintf(surface_grey(u, v, ((s >> 8) & 0xFFFF) / 65535.0f - 0.5f));
                rgb[((size_t)j * tw + i) * 3] = rgb[((size_t)j * tw + i) * 3 + 1] = rgb[((size_t)j * tw + i) * 3 + 2] = g;
            }
        xyz.resize((size_t)mesh_nx * mesh_ny * 6 * 3), red.resize((size_t)mesh_nx * mesh_ny * 6 * 2);
        size_t k = 0;
        auto corner = [&](int i, int j) {
            const float u = -W + 3.0f * W * i / mesh_nx, v = -H + 3.0f * H * j / mesh_ny, z = surface_z(u, v);
            xyz[3 * k] = (u - (float)CX) / (float)FX * z, xyz[3 * k + 1] = (v - (float)CY) / (float)FY * z, xyz[3 * k + 2] = z;
            red[2 * k] = (float)i / mesh_nx, red[2 * k + 1] = (float)j / mesh_ny;
            ++k;
        };
        for (int j = 0; j < mesh_ny; ++j)
            for (int i = 0; i < mesh_nx; ++i) {  // two triangles per quad, counter-clockwise as this camera (y down) sees them
                corner(i, j), corner(i + 1, j + 1), corner(i + 1, j);
                corner(i, j), corner(i, j + 1), corner(i + 1, j + 1);
            }
    }
    if (write_dir) {
        const bool ok = write_files(write_dir, mesh, mesh_nx, mesh_ny, xyz, red, rgb, tw, th);
        printf("%s\n", ok ? "FILES WRITTEN" : "FILES FAILED");
        return ok ? 0 : 1;
    }
    nmi_params prm;
    CHECK_NMI(nmi_params_default(&prm, W, H));
    nmi_ctx *ctx = nullptr;
    CHECK_NMI(nmi_create(&prm, &ctx));
    nmi_texture *tex = nullptr;
    if (mesh) CHECK_NMI(nmi_texture_create(ctx, rgb.data(), tw, th, &tex));
    const int64_t n_prims = mesh ? (int64_t)(xyz.size() / 9) : (int64_t)red.size();
    float *d_xyz = nullptr, *d_red = nullptr;
    uint8_t *d_frame = nullptr, *d_tmp = nullptr;
    CHECK_HIP(hipMalloc((void **)&d_xyz, xyz.size() * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_red, red.size() * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_frame, (size_t)W * H));
    CHECK_HIP(hipMalloc((void **)&d_tmp, (size_t)W * H));
    CHECK_HIP(hipMemcpy(d_xyz, xyz.data(), xyz.size() * sizeof(float), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_red, red.data(), red.size() * sizeof(float), hipMemcpyHostToDevice));

    Pipeline p;
    p.rp = nmi_render_params{FX, FY, CX, CY, 5.0f, 30.0f, 3.0f};
    if (read_dir) p.rp = nmi_render_params{cfg.fx, cfg.fy, cfg.cx, cfg.cy, cfg.render_near, cfg.render_far, cfg.render_point_size};

    // The tracker's pose: camera at the origin, ORB-SLAM axes (x right, y down, z forward) = world axes.
    float Twc0[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    // The frame was taken from a pose displaced by `truth` (world metres): render it there, turn it top-down, add sensor noise.
    const float truth[3] = {0.25f, -0.15f, 0.40f};
    {
        const float pos[3] = {truth[0], truth[1], truth[2]};
        const float look[3] = {pos[0] + Twc0[2], pos[1] + Twc0[6], pos[2] + Twc0[10]};
        const float up[3] = {Twc0[1], Twc0[5], Twc0[9]};
        const float zero[3] = {0, 0, 0};
        float mvp[16];
        CHECK_NMI(nmi_render_mvp(&p.rp, pos, look, up, zero, mvp));
        if (mesh)
            CHECK_NMI(nmi_render_mesh(ctx, d_xyz, d_red, n_prims, tex, mvp, 1, d_tmp));
        else
            CHECK_NMI(nmi_render_points(ctx, d_xyz, d_red, n_prims, mvp, 1, p.rp.point_size, d_tmp));
        CHECK_NMI(nmi_synchronize(ctx));
        std::vector<uint8_t> img((size_t)W * H), frame((size_t)W * H);
        CHECK_HIP(hipMemcpy(img.data(), d_tmp, img.size(), hipMemcpyDeviceToHost));
        unsigned q = 97531u;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float acc = 0.0f;  // sum of 4 uniforms: bell-shaped noise, sigma ~ 9 grey levels
                for (int k = 0; k < 4; ++k) {
                    q = q * 1664525u + 1013904223u;
                    acc += ((q >> 8) & 0xFFFF) / 65535.0f - 0.5f;
                }
                const float val = (float)img[(size_t)(H - 1 - y) * W + x] + 16.0f * acc;
                frame[(size_t)y * W + x] = (uint8_t)lrintf(fminf(fmaxf(val, 0.0f), 255.0f));
            }
        CHECK_HIP(hipMemcpy(d_frame, frame.data(), frame.size(), hipMemcpyHostToDevice));
    }
    if (mesh)
        CHECK_NMI(nmi_level_create_mesh(ctx, d_xyz, d_red, n_prims, tex, d_frame, 27, 27, &p.level));
    else
        CHECK_NMI(nmi_level_create(ctx, d_xyz, d_red, n_prims, d_frame, 27, 27, p.rp.point_size, &p.level));

    // 3^6 grid with the steps of ETH_small.yaml:83-88
    NmiSearchKernel initial(3, 3, 3, 3, 3, 3, 0.2f, 0.2f, 0.5f, 0.02f, 0.02f, 0.05f);
    nmi_strategy_input in;
    memset(&in, 0, sizeof in);
    CHECK_NMI(nmi_mat4_inverse(Twc0, in.Tcw));
    in.nmi_threshold = 0.05f;
    in.initial = initial.to_c();
    if (read_dir) in.nmi_threshold = cfg.nmi_threshold, in.initial = cfg.initial;
    nmi_properties props;
    nmi_properties_default(&props);
    nmi_strategy_output out;

    CHECK_NMI(nmi_relocalize_with_strategy(&in, &props, eval_level, &p, &out));  // warm-up + the checked run
    float Twc[16];
    CHECK_NMI(nmi_mat4_inverse(out.Tcw, Twc));
    for (int i = 0; i < out.iterations; ++i) {
        char line[256];
        nmi_sk_format(&out.per_iteration[i], line, sizeof line);
        printf("NmiKernel:\t%s\n", line);
    }
    printf("relocalized=%d failed=%d iterations=%d stop=%d  recovered t = (%.3f, %.3f, %.3f)  truth (%.3f, %.3f, %.3f)  NMI %.5f\n",
           out.relocalized, out.failed, out.iterations, out.stop_reason, Twc[3], Twc[7], Twc[11], truth[0], truth[1], truth[2],
           out.kernel.nmi);
    printf("recovered camera axes: x (%.4f %.4f %.4f)  y (%.4f %.4f %.4f)  z (%.4f %.4f %.4f)\n", Twc[0], Twc[4], Twc[8], Twc[1], Twc[5],
           Twc[9], Twc[2], Twc[6], Twc[10]);
    // At ~10 m a sideways step t and a turn by t / 10 m move the image almost alike (the relief separates them only
    // weakly), and the search is free to mix them: judge the lateral result by their sum, the offset an observer of the
    // image would infer.  The recovered optical axis is the third column of Twc.
    const float ex = Twc[3] + DEPTH * Twc[2], ey = Twc[7] + DEPTH * Twc[6];
    printf("lateral offset seen in the image: (%.3f, %.3f) m, truth (%.3f, %.3f)\n", ex, ey, truth[0], truth[1]);
    bool ok = out.relocalized && !out.failed && out.iterations >= 2;
    ok = ok && fabsf(ex - truth[0]) <= 0.08f && fabsf(ey - truth[1]) <= 0.08f && fabsf(Twc[11] - truth[2]) <= 0.2f;
    ok = ok && out.kernel.nmi > out.per_iteration[0].nmi;  // refinement improved the score

    p.levels_run = 0;
    p.in_run_s = 0.0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < keyframes; ++k) {
        nmi_strategy_output o;
        CHECK_NMI(nmi_relocalize_with_strategy(&in, &props, eval_level, &p, &o));
        ok = ok && o.kernel.best[0] == out.kernel.best[0] && o.kernel.nmi == out.kernel.nmi;  // deterministic
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%d keyframes, %d levels (27 renders of %lld %s + 27 warps + 729-candidate search each): %.1f keyframes/s, %.1f levels/s, "
           "%.3f ms per level\n",
           keyframes, p.levels_run, (long long)n_prims, mesh ? "textured triangles" : "points", keyframes / dt, p.levels_run / dt,
           dt / p.levels_run * 1e3);
    printf("per level: %.1f us inside nmi_level_run (parameters in -> winner out), %.1f us of host work between two calls (strategy, 27 view and 27 warp matrices)\n",
           p.in_run_s / p.levels_run * 1e6, (dt - p.in_run_s) / p.levels_run * 1e6);

    nmi_level_destroy(p.level);
    if (tex) nmi_texture_destroy(tex);
    (void)hipFree(d_xyz);
    (void)hipFree(d_red);
    (void)hipFree(d_frame);
    (void)hipFree(d_tmp);
    nmi_destroy(ctx);
    printf("%s\n", ok ? "PIPELINE OK" : "PIPELINE FAILED");
    return ok ? 0 : 1;
}
