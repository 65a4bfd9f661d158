// nmi_map.cpp -- the map files either side of the render producers, read the way the reference's loaders read them
// (declared in include/nmi_host.h; host only, no GPU, no OpenGL):
//   nmi_map_load_obj   loadOBJ        Thirdparty/Localization/objloader.cpp:140-224   -> per-corner xyz / uv arrays of nmi_render_mesh
//   nmi_map_load_xyz   loadXYZ        Thirdparty/Localization/objloader.cpp:226-264   -> xyz / red arrays of nmi_render_points
//   nmi_map_load_bmp   loadBMP_custom Thirdparty/Localization/texture.cpp:31-86       -> the RGB8 image of nmi_texture_create
// Same grammar and the same tolerances as those functions (they are fscanf / iostream loops; so are these); where the
// reference would read out of bounds or spin, these return an error instead (listed at each function).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <fstream>
#include <new>
#include <vector>

#include "nmi_host.h"

namespace {

template <typename T>
T *take(const std::vector<T> &v)
{
    T *p = static_cast<T *>(malloc(v.empty() ? sizeof(T) : v.size() * sizeof(T)));
    if (p && !v.empty()) memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}

struct File {
    FILE *f;
    explicit File(const char *path, const char *mode) : f(path ? fopen(path, mode) : nullptr) {}
    ~File()
    {
        if (f) fclose(f);
    }
};

}  // namespace

extern "C" {

void nmi_map_free(void *p) { free(p); }

// "v x y z", "vt u v", "f a/b c/d e/f" (1-based indices into the v / vt lists); any other first word: the line is skipped
// (objloader.cpp:158-197).  Every face corner becomes one vertex of the output, in file order (:199-214): the arrays
// glDrawArrays(GL_TRIANGLES) consumes.  -2: a face that is not three "position/texcoord" pairs (the reference gives up on the
// file too, :180-183); -3: an index outside the lists read so far in the file's whole (the reference reads out of bounds).
int nmi_map_load_obj(const char *path, float **xyz, float **uv, int64_t *n_vertices)
{
    if (!path || !xyz || !uv || !n_vertices) return -1;
    *xyz = nullptr, *uv = nullptr, *n_vertices = 0;
    File in(path, "r");
    if (!in.f) return -5;
    std::vector<float> pos, tex;
    std::vector<unsigned> ipos, itex;
    try {
        for (;;) {
            char word[128];
            if (fscanf(in.f, "%127s", word) == EOF) break;
            if (strcmp(word, "v") == 0) {
                float v[3] = {0, 0, 0};
                if (fscanf(in.f, "%f %f %f\n", &v[0], &v[1], &v[2]) == EOF) break;
                pos.insert(pos.end(), v, v + 3);
            } else if (strcmp(word, "vt") == 0) {
                float t[2] = {0, 0};
                if (fscanf(in.f, "%f %f\n", &t[0], &t[1]) == EOF) break;
                tex.insert(tex.end(), t, t + 2);
            } else if (strcmp(word, "f") == 0) {
                unsigned a[3], b[3];
                if (fscanf(in.f, "%u/%u %u/%u %u/%u\n", &a[0], &b[0], &a[1], &b[1], &a[2], &b[2]) != 6) return -2;
                ipos.insert(ipos.end(), a, a + 3);
                itex.insert(itex.end(), b, b + 3);
            } else {
                char rest[1000];  // any other keyword (comments, vn, usemtl, s, g ...): the remainder of its line is dropped
                if (!fgets(rest, sizeof rest, in.f)) break;
            }
        }
        std::vector<float> out_xyz, out_uv;
        out_xyz.reserve(ipos.size() * 3), out_uv.reserve(ipos.size() * 2);
        for (size_t i = 0; i < ipos.size(); ++i) {
            if (ipos[i] < 1 || (size_t)ipos[i] > pos.size() / 3 || itex[i] < 1 || (size_t)itex[i] > tex.size() / 2) return -3;
            out_xyz.insert(out_xyz.end(), pos.begin() + (size_t)(ipos[i] - 1) * 3, pos.begin() + (size_t)(ipos[i] - 1) * 3 + 3);
            out_uv.insert(out_uv.end(), tex.begin() + (size_t)(itex[i] - 1) * 2, tex.begin() + (size_t)(itex[i] - 1) * 2 + 2);
        }
        *xyz = take(out_xyz), *uv = take(out_uv);
        if (!*xyz || !*uv) {
            free(*xyz), free(*uv);
            *xyz = nullptr, *uv = nullptr;
            return -6;
        }
        *n_vertices = (int64_t)ipos.size();
    } catch (const std::bad_alloc &) {
        return -6;
    }
    return 0;
}

// offset file: three numbers; cloud file: "x y z r g b" per point, any white space between (objloader.cpp:233-262).  A
// position is read and shifted in double precision, then narrowed (:255-257); a colour is (1/256) * the file's value (:259) and
// only its red component reaches the render (GL_RED target, rendering.hpp:347): `red` [N] is what nmi_render_points takes,
// `rgb` [N][3] (optional) the whole scaled colour.  The reference's loop tests eof() before reading, so a file that ends in
// white space yields its last point twice; a second copy of a point cannot change a render (same fragment, same depth) and is
// not made here.  -2: a point with fewer than six numbers, or an offset file without three.
int nmi_map_load_xyz(const char *path, const char *offset_path, float **xyz, float **red, float **rgb, int64_t *n_points)
{
    if (!path || !offset_path || !xyz || !red || !n_points) return -1;
    *xyz = nullptr, *red = nullptr, *n_points = 0;
    if (rgb) *rgb = nullptr;
    try {
        std::ifstream off(offset_path);
        if (!off.is_open()) return -5;
        double ox, oy, oz;
        if (!(off >> ox >> oy >> oz)) return -2;
        std::ifstream in(path);
        if (!in.is_open()) return -5;
        std::vector<float> p, r, c;
        for (;;) {
            double x, y, z;
            float col[3];
            if (!(in >> x)) {
                if (in.eof()) break;  // white space (or nothing) after the last point
                return -2;
            }
            if (!(in >> y >> z >> col[0] >> col[1] >> col[2])) return -2;
            const float v[3] = {(float)(x - ox), (float)(y - oy), (float)(z - oz)};
            p.insert(p.end(), v, v + 3);
            for (float &k : col) k = (1.0f / 256.0f) * k;
            r.push_back(col[0]);
            if (rgb) c.insert(c.end(), col, col + 3);
        }
        *xyz = take(p), *red = take(r);
        if (rgb) *rgb = take(c);
        if (!*xyz || !*red || (rgb && !*rgb)) {
            free(*xyz), free(*red);
            *xyz = nullptr, *red = nullptr;
            if (rgb) free(*rgb), *rgb = nullptr;
            return -6;
        }
        *n_points = (int64_t)r.size();
    } catch (const std::bad_alloc &) {
        return -6;
    }
    return 0;
}

// A 24-bit uncompressed BMP: 54 header bytes ("BM"; compression at 0x1E = 0; bits per pixel at 0x1C = 24; width at 0x12,
// height at 0x16, image size at 0x22 -- 0 means width * height * 3), then the image bytes, which the reference reads from
// where the header ended whatever the header's data offset says (texture.cpp:49-79) and hands to glTexImage2D as GL_RGB,
// row 0 = v 0, rows unpadded.  The same bytes come back here: [height][width][3], byte 0 of a texel is the one the shader
// calls red.  -2: not such a file, or fewer image bytes than width * height * 3.
int nmi_map_load_bmp(const char *path, uint8_t **rgb, int32_t *width, int32_t *height)
{
    if (!path || !rgb || !width || !height) return -1;
    *rgb = nullptr, *width = 0, *height = 0;
    File in(path, "rb");
    if (!in.f) return -5;
    unsigned char h[54];
    if (fread(h, 1, 54, in.f) != 54 || h[0] != 'B' || h[1] != 'M') return -2;
    auto le32 = [&](int at) { return (uint32_t)h[at] | ((uint32_t)h[at + 1] << 8) | ((uint32_t)h[at + 2] << 16) | ((uint32_t)h[at + 3] << 24); };
    if (le32(0x1E) != 0 || le32(0x1C) != 24) return -2;  // (the reference reads the 16-bit depth and the 16 bits after it as one int, too)
    const uint32_t w = le32(0x12), ht = le32(0x16);
    uint64_t size = le32(0x22);
    if (w == 0 || ht == 0 || w > 32768 || ht > 32768) return -2;
    const uint64_t need = (uint64_t)w * ht * 3;
    if (size == 0) size = need;
    if (size < need) return -2;
    uint8_t *data = static_cast<uint8_t *>(malloc((size_t)need));
    if (!data) return -6;
    if (fread(data, 1, (size_t)need, in.f) != need) {
        free(data);
        return -2;
    }
    *rgb = data, *width = (int32_t)w, *height = (int32_t)ht;
    return 0;
}

}  // extern "C"
