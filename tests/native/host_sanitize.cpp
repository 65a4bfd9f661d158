// ASan/UBSan exercise of the host-side C++ (no GPU): YAML reader on hostile inputs, grid arithmetic, strategy state
// machine with odd callbacks.  Built and run by tests/test_host_sanitizers.py with -fsanitize=address,undefined.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>

#include "nmi_host.h"

static unsigned rng_state = 12345u;
static unsigned rnd() { return rng_state = rng_state * 1664525u + 1013904223u; }

static int eval_cb(void *user, const nmi_search_kernel *g, const float *Twc, int64_t *bi, float *bs)
{
    int *calls = (int *)user;
    ++*calls;
    const int64_t n = nmi_sk_candidates(g);
    if (n <= 0) return -7;
    *bi = (int64_t)(rnd() % (unsigned)n);
    *bs = (rnd() % 1000) / 1000.0f + Twc[3] * 0.0f;
    return 0;
}

int main()
{
    // 1. YAML reader: truncated / garbage / oversized inputs must fail cleanly, never crash or overrun
    const char *good =
        "%YAML:1.0\nCamera.fx: 1\nCamera.fy: 2\nCamera.cx: 3\nCamera.cy: 4\nCamera.Width: 64\nCamera.Height: 48\n"
        "NMI.SynthNumX:3\nNMI.SynthNumY: 3\nNMI.SynthNumZ: 3\nNMI.WarpNumX: 3\nNMI.WarpNumY: 3\nNMI.WarpNumZ: 3\n"
        "NMI.SynthStepX: 0.2\nNMI.SynthStepY: 0.2\nNMI.SynthStepZ: 0.5\nNMI.WarpStepX: 0.02\nNMI.WarpStepY: 0.02\nNMI.WarpStepZ: 0.05\n"
        "NMI.Init1: !!opencv-matrix\n   rows: 4\n   cols: 4\n   dt: f\n   data: [1,0,0,0,0,1,0,0,0,0,1,0,0,0,0,1]\n"
        "NMI.Render.Object: \"x.obj\"\n";
    nmi_config cfg;
    if (nmi_config_parse(good, strlen(good), &cfg) != 0 || cfg.width != 64 || !cfg.has_init1) return 1;
    const size_t glen = strlen(good);
    for (size_t cut = 0; cut <= glen; cut += 3) (void)nmi_config_parse(good, cut, &cfg);  // every truncation
    for (int it = 0; it < 2000; ++it) {  // random byte corruption
        std::string s(good);
        for (int k = 0; k < 1 + (int)(rnd() % 6); ++k) s[rnd() % s.size()] = (char)(rnd() & 0xFF);
        (void)nmi_config_parse(s.data(), s.size(), &cfg);
    }
    std::string longpath = std::string(good) + "NMI.Render.Cloud: \"" + std::string(5000, 'a') + "\"\n";
    if (nmi_config_parse(longpath.data(), longpath.size(), &cfg) != 0 || strlen(cfg.render_cloud) != 511) return 2;
    std::string bigmat = std::string(good) + "NMI.Init2: !!opencv-matrix\n rows: 4\n cols: 4\n dt: f\n data: [";
    for (int i = 0; i < 100000; ++i) bigmat += "1,";
    bigmat += "1]\n";
    if (nmi_config_parse(bigmat.data(), bigmat.size(), &cfg) == 0) return 3;  // wrong element count must be rejected

    // 2. grid arithmetic on degenerate descriptors
    nmi_search_kernel k;
    nmi_sk_init(&k);
    char buf[16];
    (void)nmi_sk_format(&k, buf, sizeof buf);   // truncating format must not overrun
    (void)nmi_sk_format(&k, buf, 0);
    if (nmi_sk_candidates(&k) != 0 || nmi_sk_set_best_from_index(&k, 0, 1.0f) == 0) return 4;
    int32_t idx6[6] = {0, 0, 0, 0, 0, 0};
    if (nmi_sk_linear_index(&k, idx6) != -1) return 5;
    nmi_sk_resize(&k, nullptr);

    // 3. strategy with random winners, zero / NaN / huge drift, singular pose
    for (int it = 0; it < 300; ++it) {
        nmi_strategy_input in;
        memset(&in, 0, sizeof in);
        for (int i = 0; i < 4; ++i) in.Tcw[i * 5] = 1.0f;
        in.Tcw[3] = (float)(rnd() % 100);
        nmi_sk_init(&in.initial);
        for (int a = 0; a < 6; ++a) {
            in.initial.num[a] = 1 + (int)(rnd() % 4);
            in.initial.step[a] = 0.001f * (float)(1 + rnd() % 300);
        }
        in.nmi_threshold = 0.1f;
        in.not_initialized = (int)(rnd() & 1);
        if (it % 3 == 0)
            for (int a = 0; a < 3; ++a) {
                in.distance_since_last[a] = (it % 9 == 0) ? NAN : (float)(rnd() % 2000) / 100.0f;
                in.rotation_since_last[a] = (float)(rnd() % 100) / 1000.0f;
            }
        nmi_strategy_output out;
        int calls = 0;
        const int rc = nmi_relocalize_with_strategy(&in, nullptr, eval_cb, &calls, &out);
        if (rc != 0 || out.iterations != calls || out.iterations > 4) return 6;
    }
    nmi_strategy_input sing;
    memset(&sing, 0, sizeof sing);  // all-zero pose: not invertible
    nmi_sk_init(&sing.initial);
    for (int a = 0; a < 6; ++a) sing.initial.num[a] = 1, sing.initial.step[a] = 0.1f;
    nmi_strategy_output out;
    int calls = 0;
    if (nmi_relocalize_with_strategy(&sing, nullptr, eval_cb, &calls, &out) == 0) return 7;
    float inv[16], zero[16] = {0};
    if (nmi_mat4_inverse(zero, inv) == 0) return 8;
    // 4. map files: every truncation and random corruption of a small OBJ / XYZ / BMP must load or fail cleanly
    {
        const char *dir = getenv("TMPDIR") ? getenv("TMPDIR") : "/tmp";
        const std::string path = std::string(dir) + "/nmi_host_sanitize_map.bin", off = std::string(dir) + "/nmi_host_sanitize_map.offset";
        auto put = [&](const std::string &name, const std::string &bytes) {
            FILE *f = fopen(name.c_str(), "wb");
            if (f) fwrite(bytes.data(), 1, bytes.size(), f), fclose(f);
        };
        const std::string obj = "# c\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nvn 0 0 1\nf 1/1 2/2 3/3\nf 3/3 2/2 1/1\n";
        const std::string xyz = "10 20 30 255 0 0\n11 21 31 128 64 32\n12 22 32 1 2 3\n";
        std::string bmp_w(54 + 4 * 4 * 3, '\x7f');
        bmp_w[0] = 'B', bmp_w[1] = 'M';
        for (int i = 2; i < 54; ++i) bmp_w[i] = 0;
        bmp_w[0x0A] = 54, bmp_w[0x12] = 4, bmp_w[0x16] = 4, bmp_w[0x1A] = 1, bmp_w[0x1C] = 24, bmp_w[0x22] = 48;
        const std::string bmp = bmp_w;
        put(off, "1 2 3\n");
        float *a = nullptr, *b = nullptr;
        uint8_t *img = nullptr;
        int64_t n = 0;
        int32_t w = 0, h = 0;
        auto try_all = [&](const std::string &bytes) {
            put(path, bytes);
            if (nmi_map_load_obj(path.c_str(), &a, &b, &n) == 0) nmi_map_free(a), nmi_map_free(b);
            if (nmi_map_load_xyz(path.c_str(), off.c_str(), &a, &b, nullptr, &n) == 0) nmi_map_free(a), nmi_map_free(b);
            if (nmi_map_load_bmp(path.c_str(), &img, &w, &h) == 0) nmi_map_free(img);
        };
        put(path, obj);
        if (nmi_map_load_obj(path.c_str(), &a, &b, &n) != 0 || n != 6) return 9;
        nmi_map_free(a), nmi_map_free(b);
        put(path, bmp);
        if (nmi_map_load_bmp(path.c_str(), &img, &w, &h) != 0 || w != 4 || h != 4) return 10;
        nmi_map_free(img);
        for (const std::string *src : {&obj, &xyz, &bmp}) {
            for (size_t cut = 0; cut <= src->size(); cut += 2) try_all(src->substr(0, cut));
            for (int it = 0; it < 300; ++it) {
                std::string s(*src);
                for (int k = 0; k < 1 + (int)(rnd() % 4); ++k) s[rnd() % s.size()] = (char)(rnd() & 0xFF);
                try_all(s);
            }
        }
        remove(path.c_str()), remove(off.c_str());
    }
    puts("host sanitize ok");
    return 0;
}
