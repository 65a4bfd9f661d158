// nmi_driver.cpp -- host-side search driver behind include/nmi_host.h.
//
// Plain C++ (no GPU, no OpenCV): the grid descriptor arithmetic, the arg-max rule, the pose update and the
// coarse-to-fine strategy state machine that surround the scoring kernel in the reference.  Each function
// names the reference lines whose behaviour it reproduces (paths relative to the reference repository root).
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "nmi_config.hpp"
#include "nmi_host.h"

namespace {

enum { SX = 0, SY = 1, SZ = 2, WX = 3, WY = 4, WZ = 5 };

struct Mat4 {
    float m[16];
    float &at(int r, int c) { return m[r * 4 + c]; }
    float at(int r, int c) const { return m[r * 4 + c]; }
};

Mat4 identity()
{
    Mat4 r;
    memset(r.m, 0, sizeof r.m);
    r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f;
    return r;
}

Mat4 mul(const Mat4 &a, const Mat4 &b)
{
    Mat4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = 0.0f;
            for (int k = 0; k < 4; ++k) s += a.at(i, k) * b.at(k, j);
            r.at(i, j) = s;
        }
    return r;
}

// General 4x4 inverse (Gauss-Jordan, partial pivoting, in double) -- cv::Mat::inv() of the reference call sites
// (Tracking.cc:1977,1981) is a general LU inverse as well; poses are rigid, so both agree to float rounding.
bool inverse(const Mat4 &a, Mat4 &out)
{
    double w[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            w[i][j] = a.at(i, j);
            w[i][j + 4] = i == j ? 1.0 : 0.0;
        }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r)
            if (fabs(w[r][c]) > fabs(w[piv][c])) piv = r;
        if (w[piv][c] == 0.0) return false;
        if (piv != c)
            for (int j = 0; j < 8; ++j) {
                const double t = w[c][j];
                w[c][j] = w[piv][j];
                w[piv][j] = t;
            }
        const double d = w[c][c];
        for (int j = 0; j < 8; ++j) w[c][j] /= d;
        for (int r = 0; r < 4; ++r)
            if (r != c) {
                const double f = w[r][c];
                if (f != 0.0)
                    for (int j = 0; j < 8; ++j) w[r][j] -= f * w[c][j];
            }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) out.at(i, j) = (float)w[i][j + 4];
    return true;
}

bool on_border(const nmi_search_kernel *k, int a) { return (k->best[a] == k->num[a] - 1 || k->best[a] == 0) && k->num[a] > 1; }

}  // namespace

extern "C" {

void nmi_properties_default(nmi_properties *p)
{
    if (!p) return;
    p->max_iteration_count = nmi_prop_MAX_ITERATION_COUNT;
    p->reloc_frequency = nmi_prop_RELOC_FREQUENCY;
    p->step_factor = nmi_prop_STEPFACTOR;
    p->min_kernel_rotation = nmi_prop_MIN_KERNEL_ROTATION;
    p->min_kernel_translation = nmi_prop_MIN_KERNEL_TRANSLATION;
    p->use_bg = nmi_prop_BG ? 1 : 0;
}

// NmiSearchKernel::NmiSearchKernel(), nmiSearchKernel.cpp:35-38 / reset() :153-158.
void nmi_sk_init(nmi_search_kernel *k)
{
    for (int a = 0; a < 6; ++a) {
        k->num[a] = -1;
        k->step[a] = -1.0f;
        k->best[a] = -1;
    }
    k->nmi = 0.0f;
}
void nmi_sk_reset(nmi_search_kernel *k) { nmi_sk_init(k); }

// isMiddle(), nmiSearchKernel.cpp:99-102: every best index equals n/2 (integer division).
int nmi_sk_is_middle(const nmi_search_kernel *k)
{
    for (int a = 0; a < 6; ++a)
        if (k->best[a] != k->num[a] / 2) return 0;
    return 1;
}

// resizeKernel(), nmiSearchKernel.cpp:104-141: the step of an axis shrinks by STEPFACTOR unless the best
// cell sits on the border of an axis with more than one cell; an axis whose step fell under the minimum
// (translation 0.005 m, rotation 0.001 rad; the comparison is float-vs-double as in the reference) collapses to 1.
void nmi_sk_resize(nmi_search_kernel *k, const nmi_properties *props)
{
    nmi_properties d;
    if (!props) {
        nmi_properties_default(&d);
        props = &d;
    }
    for (int a = 0; a < 6; ++a)
        if (!on_border(k, a)) k->step[a] *= props->step_factor;
    for (int a = 0; a < 6; ++a) {
        const double min_step = a < 3 ? props->min_kernel_translation : props->min_kernel_rotation;
        if (k->step[a] < min_step) k->num[a] = 1;
    }
}

int64_t nmi_sk_candidates(const nmi_search_kernel *k)
{
    int64_t n = 1;
    for (int a = 0; a < 6; ++a) n *= k->num[a] > 0 ? k->num[a] : 0;
    return n;
}

// operator<<, nmiSearchKernel.cpp:183-195: fixed, precision 5, "sX: %2d/%1d: %6f;\t sY: ..." and the NMI last.
int nmi_sk_format(const nmi_search_kernel *k, char *buf, size_t cap)
{
    static const char *names[6] = {"sX", "sY", "sZ", "rX", "rY", "rZ"};
    size_t n = 0;
    for (int a = 0; a < 6; ++a) {
        const int w = snprintf(buf + n, n < cap ? cap - n : 0, "%s%s: %2d/%1d: %6.5f", a ? ";\t " : "", names[a], k->best[a],
                               k->num[a], (double)k->step[a]);
        if (w < 0) return -1;
        n += (size_t)w;
    }
    const int w = snprintf(buf + n, n < cap ? cap - n : 0, ";\t NMI: %.5f", (double)k->nmi);
    if (w < 0) return -1;
    return (int)(n + (size_t)w);
}

// Scan order of find_max_elements (helperFunctions.cpp:53-64): wz, wy, wx, sz, sy, sx; sx fastest.
int64_t nmi_sk_linear_index(const nmi_search_kernel *k, const int32_t idx6[6])
{
    static const int order[6] = {WZ, WY, WX, SZ, SY, SX};
    int64_t lin = 0;
    for (int o = 0; o < 6; ++o) {
        const int a = order[o];
        if (idx6[a] < 0 || idx6[a] >= k->num[a]) return -1;
        lin = lin * k->num[a] + idx6[a];
    }
    return lin;
}

int nmi_sk_set_best_from_index(nmi_search_kernel *k, int64_t linear_index, float score)
{
    static const int order[6] = {SX, SY, SZ, WX, WY, WZ};  // fastest axis first
    if (linear_index < 0 || linear_index >= nmi_sk_candidates(k)) return -1;
    for (int o = 0; o < 6; ++o) {
        const int a = order[o];
        k->best[a] = (int32_t)(linear_index % k->num[a]);
        linear_index /= k->num[a];
    }
    k->nmi = score;  // Tracking.cc:1952-1953
    return 0;
}

// helperFunctions::find_max_elements, helperFunctions.cpp:50-103: pass 1 max from 0 with strict '>', pass 2 every
// cell equal to the max in scan order.  The caller uses element 0 (Tracking.cc:1952).
int64_t nmi_find_max_elements(const float *ratings, int64_t n, int64_t *ties, int64_t cap, float *max_value)
{
    float mx = 0.0f;
    for (int64_t i = 0; i < n; ++i)
        if (ratings[i] > mx) mx = ratings[i];
    int64_t count = 0;
    for (int64_t i = 0; i < n; ++i)
        if (ratings[i] == mx) {
            if (ties && count < cap) ties[count] = i;
            ++count;
        }
    if (max_value) *max_value = mx;
    return count;
}

int nmi_mat4_inverse(const float m[16], float out[16])
{
    Mat4 a, r;
    memcpy(a.m, m, sizeof a.m);
    if (!inverse(a, r)) return -1;
    memcpy(out, r.m, sizeof r.m);
    return 0;
}

// Rendering::calculateTranslationCV, rendering.hpp:668-694, with the camera set from Twc as setupCam does
// (ioData.cpp:177-197: position = translation column, view direction = third column, up = second column):
//   dir_y = up / |up|, dir_z = -view / |view|, dir_x = dir_y rotated by -90 degrees about dir_z,
//   translation = sum over axes of (index - (n-1)/2) * step * dir.
int nmi_calculate_translation(const float Twc[16], const nmi_search_kernel *k, int32_t sx, int32_t sy, int32_t sz,
                              float out_xyz[3])
{
    if (!Twc || !k || !out_xyz) return -1;
    const float up[3] = {Twc[1], Twc[5], Twc[9]};
    const float view[3] = {Twc[2], Twc[6], Twc[10]};
    float len = sqrtf(up[0] * up[0] + up[1] * up[1] + up[2] * up[2]);
    const float dy[3] = {up[0] / len, up[1] / len, up[2] / len};
    len = sqrtf(view[0] * view[0] + view[1] * view[1] + view[2] * view[2]);
    const float dz[3] = {-view[0] / len, -view[1] / len, -view[2] / len};
    // glm::rotate(dir_y, radians(-90), dir_z): Rodrigues about the (normalised) axis dz
    float alen = sqrtf(dz[0] * dz[0] + dz[1] * dz[1] + dz[2] * dz[2]);
    const float ax[3] = {dz[0] / alen, dz[1] / alen, dz[2] / alen};
    const float ang = -90.0f * 0.01745329251994329576923690768489f;
    const float c = cosf(ang), s = sinf(ang);
    const float dot = ax[0] * dy[0] + ax[1] * dy[1] + ax[2] * dy[2];
    const float cr[3] = {ax[1] * dy[2] - ax[2] * dy[1], ax[2] * dy[0] - ax[0] * dy[2], ax[0] * dy[1] - ax[1] * dy[0]};
    float dx[3];
    for (int i = 0; i < 3; ++i) dx[i] = dy[i] * c + cr[i] * s + ax[i] * dot * (1.0f - c);
    const float ox = ((float)k->num[SX] - 1.0f) / 2.0f, oy = ((float)k->num[SY] - 1.0f) / 2.0f,
                oz = ((float)k->num[SZ] - 1.0f) / 2.0f;
    for (int i = 0; i < 3; ++i)
        out_xyz[i] = ((float)sx - ox) * k->step[SX] * dx[i] + ((float)sy - oy) * k->step[SY] * dy[i] +
                     ((float)sz - oz) * k->step[SZ] * dz[i];
    return 0;
}

// Tracking::CalculateNMIRelocalization, Tracking.cc:2374-2419: rot_a = (best_a - n_a/2) * stepRad_a with integer
// n/2, R = Rz*Ry*Rx, newLoc = Twc * [R|0], then the translation of the best render cell is added to the last column.
int nmi_calculate_relocalization(const float Twc[16], const nmi_search_kernel *k, float out_Twc[16])
{
    if (!Twc || !k || !out_Twc) return -1;
    const float rx = (float)(k->best[WX] - k->num[WX] / 2) * k->step[WX];
    const float ry = (float)(k->best[WY] - k->num[WY] / 2) * k->step[WY];
    const float rz = (float)(k->best[WZ] - k->num[WZ] / 2) * k->step[WZ];
    Mat4 Rx = identity(), Ry = identity(), Rz = identity(), T;
    Rx.at(1, 1) = cosf(rx), Rx.at(1, 2) = -sinf(rx), Rx.at(2, 1) = sinf(rx), Rx.at(2, 2) = cosf(rx);
    Ry.at(0, 0) = cosf(ry), Ry.at(0, 2) = sinf(ry), Ry.at(2, 0) = -sinf(ry), Ry.at(2, 2) = cosf(ry);
    Rz.at(0, 0) = cosf(rz), Rz.at(0, 1) = -sinf(rz), Rz.at(1, 0) = sinf(rz), Rz.at(1, 1) = cosf(rz);
    memcpy(T.m, Twc, sizeof T.m);
    const Mat4 n = mul(T, mul(Rz, mul(Ry, Rx)));
    float t[3];
    nmi_calculate_translation(Twc, k, k->best[SX], k->best[SY], k->best[SZ], t);
    memcpy(out_Twc, n.m, sizeof n.m);
    out_Twc[3] += t[0];
    out_Twc[7] += t[1];
    out_Twc[11] += t[2];
    return 0;
}

// Tracking::RelocalizeWithNMIStrategy, Tracking.cc:1987-2179.
int nmi_relocalize_with_strategy(const nmi_strategy_input *in, const nmi_properties *props_in, nmi_eval_grid_fn eval_grid,
                                 void *user, nmi_strategy_output *out)
{
    if (!in || !eval_grid || !out) return -1;
    nmi_properties props;
    if (props_in)
        props = *props_in;
    else
        nmi_properties_default(&props);
    if (props.max_iteration_count > NMI_STRATEGY_MAX_ITER) return -1;
    memset(out, 0, sizeof *out);

    nmi_search_kernel cur, last;
    nmi_sk_reset(&cur);   // :1997
    nmi_sk_reset(&last);  // :1998
    const nmi_search_kernel &init = in->initial;

    // ---- grid seeding (:2001-2070) ----
    if (in->distance_since_last[0] > 0.0f) {
        // 2 % of the path / rotation accumulated since the last NMI fix (:2004-2010); axes whose step is below the
        // minimum get a single cell (:2014-2043), the others the YAML counts.
        for (int a = 0; a < 6; ++a) {
            const float travelled = a < 3 ? in->distance_since_last[a] : in->rotation_since_last[a - 3];
            cur.step[a] = travelled * 0.02;
            const double min_step = a < 3 ? props.min_kernel_translation : props.min_kernel_rotation;
            cur.num[a] = cur.step[a] < min_step ? 1 : init.num[a];
        }
    } else if (in->not_initialized) {
        cur = init;  // :2055-2063: 5x5x5 renders during initialisation, YAML warps and steps
        nmi_sk_reset(&last);
        cur.num[SX] = cur.num[SY] = cur.num[SZ] = 5;
        for (int a = 0; a < 6; ++a) cur.best[a] = -1;
        cur.nmi = 0.0f;
    } else {
        for (int a = 0; a < 6; ++a) {  // :2066: setKernel(InitialNmiKernel) copies counts and steps only
            cur.num[a] = init.num[a];
            cur.step[a] = init.step[a];
        }
    }

    Mat4 Tcw, TcwSave, TcwSaveLast;
    memcpy(Tcw.m, in->Tcw, sizeof Tcw.m);
    TcwSave = Tcw;      // :2082-2085
    TcwSaveLast = Tcw;  // :2087

    int i = 0, under = 0, iterations = 0, stop = 0;
    for (;;) {
        ++i;
        if (i > props.max_iteration_count) {  // :2092
            stop = 0;
            break;
        }
        // RelocalizeWithNMI (:1851-1985): score the grid around the current pose, take the winner, move the pose.
        Mat4 Twc;
        if (!inverse(Tcw, Twc)) return -2;
        int64_t best_index = -1;
        float best_score = 0.0f;
        const int rc = eval_grid(user, &cur, Twc.m, &best_index, &best_score);
        if (rc != 0) return rc;
        if (nmi_sk_set_best_from_index(&cur, best_index, best_score) != 0) return -3;  // :1952-1953
        Mat4 newLoc;
        nmi_calculate_relocalization(Twc.m, &cur, newLoc.m);  // :1956
        if (!inverse(newLoc, Tcw)) return -2;                 // :1977 / :1981  SetPose(newLoc.inv())
        if (iterations < NMI_STRATEGY_MAX_ITER) out->per_iteration[iterations] = cur;
        ++iterations;

        if (i > 1 && nmi_sk_is_middle(&cur)) {  // :2108-2110
            stop = 1;
            break;
        }
        if (i > 1) {  // :2112-2121
            if ((double)(cur.nmi / last.nmi) < 1.001) {
                if (under > 0) {
                    stop = 2;
                    break;
                }
                ++under;
            } else {
                under = 0;
            }
        }
        last = cur;                  // NMIobjectsReInitialization, localization.cpp:410-420
        nmi_sk_resize(&cur, &props);
        TcwSaveLast = Tcw;           // :2126-2129
    }

    // ---- accept / reject (:2134-2168) ----
    if (cur.nmi < last.nmi) {
        Tcw = TcwSaveLast;
        out->reverted_to_previous = 1;
    }
    const double base = 5.0;
    const double dist = sqrt((double)in->distance_since_last[0] * in->distance_since_last[0] +
                             (double)in->distance_since_last[1] * in->distance_since_last[1] +
                             (double)in->distance_since_last[2] * in->distance_since_last[2]);
    double thr;
    if (dist < base) {
        thr = in->nmi_threshold;
    } else {
        thr = in->nmi_threshold * (base / dist);
        if (thr < in->nmi_threshold / 2) thr = in->nmi_threshold / 2;
    }
    out->relocalized = iterations > 0 ? 1 : 0;  // SetNMIRelocalized(true) in every RelocalizeWithNMI call (:1978,1982)
    if (cur.nmi < thr) {
        Tcw = TcwSave;
        out->relocalized = 0;
        out->failed = 1;
    }
    memcpy(out->Tcw, Tcw.m, sizeof Tcw.m);
    out->iterations = iterations;
    out->stop_reason = stop;
    out->nmi_threshold_used = (float)thr;
    out->kernel = cur;
    out->last_kernel = last;
    return 0;
}

}  // extern "C"
