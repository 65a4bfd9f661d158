// nmi_yaml.cpp -- OpenCV-free reader for the run-time configuration surface of the reference: the `Camera.*` and
// `NMI.*` keys that cv::FileStorage hands to NmiObjects / Tracking (Thirdparty/Localization/localization.cpp:131-253,
// src/Tracking.cc:150-159; example Examples/Monocular/ETH_small.yaml:8-24,62-96).
//
// Subset of cv::FileStorage YAML that those files use: "%YAML:1.0" header, '#' comments, flat `key: value` lines
// (the reference files also contain `key:value` without a blank, which OpenCV accepts), quoted strings, and
// `!!opencv-matrix` nodes with rows / cols / dt / data (data may span lines).
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "nmi_host.h"

namespace {

struct Node {
    std::string scalar;
    bool is_matrix = false;
    int rows = 0, cols = 0;
    std::vector<double> data;
};

std::string trim(const std::string &s)
{
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) ++a;
    while (b > a && isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}

std::string strip_comment(const std::string &line)
{
    bool in_quote = false;
    for (size_t i = 0; i < line.size(); ++i) {
        if (line[i] == '"') in_quote = !in_quote;
        if (line[i] == '#' && !in_quote) return line.substr(0, i);
    }
    return line;
}

std::string unquote(const std::string &v)
{
    if (v.size() >= 2 && ((v.front() == '"' && v.back() == '"') || (v.front() == '\'' && v.back() == '\''))) return v.substr(1, v.size() - 2);
    return v;
}

bool parse(const char *text, size_t len, std::map<std::string, Node> &out)
{
    std::vector<std::string> lines;
    {
        std::string cur;
        for (size_t i = 0; i < len; ++i) {
            if (text[i] == '\n') {
                lines.push_back(cur);
                cur.clear();
            } else if (text[i] != '\r') {
                cur.push_back(text[i]);
            }
        }
        lines.push_back(cur);
    }
    for (size_t li = 0; li < lines.size(); ++li) {
        std::string raw = strip_comment(lines[li]);
        if (li == 0 && raw.size() >= 3 && (unsigned char)raw[0] == 0xEF) raw = raw.substr(3);  // UTF-8 BOM
        const std::string line = trim(raw);
        if (line.empty() || line[0] == '%' || line == "---" || line == "...") continue;
        if (isspace((unsigned char)raw[0])) continue;  // indented: belongs to a node handled below
        const size_t colon = line.find(':');
        if (colon == std::string::npos) return false;
        const std::string key = trim(line.substr(0, colon));
        const std::string val = trim(line.substr(colon + 1));
        Node n;
        if (val.rfind("!!opencv-matrix", 0) == 0) {
            n.is_matrix = true;
            std::string blob;
            size_t lj = li + 1;
            for (; lj < lines.size(); ++lj) {
                const std::string r2 = strip_comment(lines[lj]);
                if (trim(r2).empty()) continue;
                if (!isspace((unsigned char)r2[0])) break;
                blob += " " + trim(r2);
            }
            li = lj - 1;
            auto field = [&](const char *name) -> std::string {
                const size_t p = blob.find(std::string(name) + ":");
                if (p == std::string::npos) return "";
                size_t q = p + strlen(name) + 1;
                while (q < blob.size() && isspace((unsigned char)blob[q])) ++q;
                if (q < blob.size() && blob[q] == '[') {
                    const size_t e = blob.find(']', q);
                    return e == std::string::npos ? "" : blob.substr(q + 1, e - q - 1);
                }
                size_t e = q;
                while (e < blob.size() && !isspace((unsigned char)blob[e])) ++e;
                return blob.substr(q, e - q);
            };
            n.rows = atoi(field("rows").c_str());
            n.cols = atoi(field("cols").c_str());
            std::string d = field("data");
            for (char &c : d)
                if (c == ',') c = ' ';
            const char *p = d.c_str();
            char *end = nullptr;
            for (;;) {
                const double v = strtod(p, &end);
                if (end == p) break;
                n.data.push_back(v);
                p = end;
            }
            if (n.rows <= 0 || n.cols <= 0 || (int)n.data.size() != n.rows * n.cols) return false;
        } else {
            n.scalar = unquote(val);
        }
        out[key] = n;
    }
    return true;
}

bool number(const std::map<std::string, Node> &m, const char *key, double &v)
{
    auto it = m.find(key);
    if (it == m.end() || it->second.is_matrix || it->second.scalar.empty()) return false;
    char *end = nullptr;
    v = strtod(it->second.scalar.c_str(), &end);
    return end != it->second.scalar.c_str();
}

void text_field(const std::map<std::string, Node> &m, const char *key, char *dst, size_t cap)
{
    dst[0] = 0;
    auto it = m.find(key);
    if (it == m.end() || it->second.is_matrix) return;
    snprintf(dst, cap, "%s", it->second.scalar.c_str());
}

bool matrix4(const std::map<std::string, Node> &m, const char *key, float out[16])
{
    auto it = m.find(key);
    if (it == m.end() || !it->second.is_matrix || it->second.rows != 4 || it->second.cols != 4) return false;
    for (int i = 0; i < 16; ++i) out[i] = (float)it->second.data[i];
    return true;
}

}  // namespace

extern "C" {

int nmi_config_parse(const char *text, size_t len, nmi_config *out)
{
    if (!text || !out) return -1;
    memset(out, 0, sizeof *out);
    std::map<std::string, Node> m;
    if (!parse(text, len, m)) return -2;
    double v;
    // Camera.* : localization.cpp:135-136,159-168 (renderer size, K)
    if (!number(m, "Camera.Width", v)) return -3;
    out->width = (int32_t)v;
    if (!number(m, "Camera.Height", v)) return -3;
    out->height = (int32_t)v;
    if (!number(m, "Camera.fx", out->fx) || !number(m, "Camera.fy", out->fy) || !number(m, "Camera.cx", out->cx) ||
        !number(m, "Camera.cy", out->cy))
        return -3;
    // NMI.* grid: localization.cpp:213-253.  cv::FileNode -> int / float conversions of the reference.
    static const char *num_keys[6] = {"NMI.SynthNumX", "NMI.SynthNumY", "NMI.SynthNumZ", "NMI.WarpNumX", "NMI.WarpNumY", "NMI.WarpNumZ"};
    static const char *step_keys[6] = {"NMI.SynthStepX", "NMI.SynthStepY", "NMI.SynthStepZ", "NMI.WarpStepX", "NMI.WarpStepY", "NMI.WarpStepZ"};
    nmi_sk_init(&out->initial);
    for (int a = 0; a < 6; ++a) {
        if (!number(m, num_keys[a], v)) return -4;
        out->initial.num[a] = (int32_t)v;
        if (!number(m, step_keys[a], v)) return -4;
        out->initial.step[a] = (float)v;
    }
    // Tracking.cc:152-157
    out->has_init1 = matrix4(m, "NMI.Init1", out->init1) ? 1 : 0;
    out->has_init2 = matrix4(m, "NMI.Init2", out->init2) ? 1 : 0;
    if (number(m, "NMI.Offset", v)) out->init_offset = (int32_t)v;
    if (number(m, "NMI.Treshold", v)) out->nmi_threshold = (float)v;
    // NMI.Render.*: localization.cpp:133,146-157 (inputs of the render-stack producer; carried, not interpreted here)
    if (number(m, "NMI.Render.PointSize", v)) out->render_point_size = (float)v;
    if (number(m, "NMI.Render.NearPlane", v)) out->render_near = (float)v;
    if (number(m, "NMI.Render.FarPlane", v)) out->render_far = (float)v;
    text_field(m, "NMI.Render.Object", out->render_object, sizeof out->render_object);
    text_field(m, "NMI.Render.Texture", out->render_texture, sizeof out->render_texture);
    text_field(m, "NMI.Render.Cloud", out->render_cloud, sizeof out->render_cloud);
    text_field(m, "NMI.Render.Offset", out->render_offset, sizeof out->render_offset);
    return 0;
}

int nmi_config_load(const char *yaml_path, nmi_config *out)
{
    if (!yaml_path || !out) return -1;
    FILE *f = fopen(yaml_path, "rb");
    if (!f) return -5;
    std::string text;
    char buf[4096];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
    fclose(f);
    return nmi_config_parse(text.data(), text.size(), out);
}

}  // extern "C"
