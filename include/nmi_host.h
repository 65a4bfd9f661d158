/*
 * nmi_host.h -- C ABI of the host-side search driver in libnmi_hip.so (no GPU work in here).
 *
 * These entry points restate, as plain C structs and functions, the host logic that surrounds the
 * scoring kernel in the reference (paths relative to the reference repository root):
 *   NmiSearchKernel                       Thirdparty/Localization/nmiSearchKernel.hpp:25-86, .cpp:25-195
 *   helperFunctions::find_max_elements    Thirdparty/Localization/helperFunctions.cpp:50-103
 *   Tracking::CalculateNMIRelocalization  src/Tracking.cc:2374-2419
 *   Rendering::calculateTranslationCV     Thirdparty/Localization/rendering.hpp:668-694
 *   Tracking::RelocalizeWithNMIStrategy   src/Tracking.cc:1987-2179
 * C++ callers can use the same-named classes of orbslam2_nmi_amd/host/ (nmi_search_kernel.hpp, nmi_driver.hpp)
 * directly; this header is what a ctypes / cgo / JNI binding would bind.
 *
 * Matrices are float[16], row-major 4x4, the layout of the reference's CV_32F cv::Mat poses.
 */
#ifndef NMI_HOST_H
#define NMI_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Axis order everywhere: 0 synthX, 1 synthY, 2 synthZ (translation / renders), 3 warpX, 4 warpY, 5 warpZ (rotation / warps). */
typedef struct nmi_search_kernel {
    int32_t num[6];  /* numSynthX..numWarpZ   (nmiSearchKernel.hpp:29) */
    float step[6];   /* stepX..stepZ [m], stepRadX..stepRadZ [rad] (:30) */
    int32_t best[6]; /* bestSynthX..bestWarpZ (:35) */
    float nmi;       /* NMI                   (:32) */
} nmi_search_kernel;

/* Compile-time properties of Thirdparty/Localization/allProperties.hpp:27-50 as run-time values. */
typedef struct nmi_properties {
    int32_t max_iteration_count;   /* nmi_prop_MAX_ITERATION_COUNT 4      */
    int32_t reloc_frequency;       /* nmi_prop_RELOC_FREQUENCY 2          */
    float step_factor;             /* nmi_prop_STEPFACTOR 0.5f            */
    double min_kernel_rotation;    /* nmi_prop_MIN_KERNEL_ROTATION 0.001  */
    double min_kernel_translation; /* nmi_prop_MIN_KERNEL_TRANSLATION 0.005 */
    int32_t use_bg;                /* nmi_prop_BG true                    */
} nmi_properties;
void nmi_properties_default(nmi_properties *p);

/* NmiSearchKernel member functions (nmiSearchKernel.cpp). */
void nmi_sk_init(nmi_search_kernel *k);                   /* default ctor: everything -1, NMI 0 (:35-38) */
void nmi_sk_reset(nmi_search_kernel *k);                  /* reset() (:153-158) */
int nmi_sk_is_middle(const nmi_search_kernel *k);         /* isMiddle() (:99-102), integer n/2 */
void nmi_sk_resize(nmi_search_kernel *k, const nmi_properties *props); /* resizeKernel() (:104-141) */
int64_t nmi_sk_candidates(const nmi_search_kernel *k);    /* product of the six counts */
int nmi_sk_format(const nmi_search_kernel *k, char *buf, size_t cap); /* operator<< (:183-195); returns length */

/* Linear rating index <-> best indices.  index = ((((wz*nWy+wy)*nWx+wx)*nSz+sz)*nSy+sy)*nSx+sx, the scan order of
 * find_max_elements (helperFunctions.cpp:53-64) and the layout of nmi_search_grid's rating table. */
int64_t nmi_sk_linear_index(const nmi_search_kernel *k, const int32_t idx6[6]);
int nmi_sk_set_best_from_index(nmi_search_kernel *k, int64_t linear_index, float score);

/* Host arg-max with the reference's rule; writes up to `cap` tie indices in scan order, returns the tie count
 * (0 when no cell equals the maximum: every cell negative or NaN). */
int64_t nmi_find_max_elements(const float *ratings, int64_t n, int64_t *ties, int64_t cap, float *max_value);

/* Rendering::calculateTranslationCV for camera pose Twc and grid cell (sx, sy, sz) of `k`. */
int nmi_calculate_translation(const float Twc[16], const nmi_search_kernel *k, int32_t sx, int32_t sy, int32_t sz,
                              float out_xyz[3]);
/* Tracking::CalculateNMIRelocalization: refined Twc from the best indices of `k`. */
int nmi_calculate_relocalization(const float Twc[16], const nmi_search_kernel *k, float out_Twc[16]);
int nmi_mat4_inverse(const float m[16], float out[16]);

/*
 * Scores one candidate grid centred on Twc: the replacement for the body of Tracking::RelocalizeWithNMI
 * (src/Tracking.cc:1871-1905): produce the render stack for grid->num[0..2] / step[0..2] around Twc and the warp
 * stack for num[3..5] / step[3..5], run nmi_search_grid, return the winner.  Return 0 on success.
 */
typedef int (*nmi_eval_grid_fn)(void *user, const nmi_search_kernel *grid, const float Twc[16], int64_t *best_index,
                                float *best_score);

typedef struct nmi_strategy_input {
    float Tcw[16];                 /* pose of the frame / keyframe before the search (GetPose()) */
    float distance_since_last[3];  /* mDistanceSinceLastNMI  (Tracking.cc:651-653) */
    float rotation_since_last[3];  /* mRotationSinceLastNMI  (Tracking.cc:661)     */
    int32_t not_initialized;       /* mState == NOT_INITIALIZED (Tracking.cc:2055)  */
    float nmi_threshold;           /* NMI.Treshold -> mfNmiInitTresholf (Tracking.cc:157) */
    nmi_search_kernel initial;     /* InitialNmiKernel from the YAML NMI.* keys (localization.cpp:185-253) */
} nmi_strategy_input;

#define NMI_STRATEGY_MAX_ITER 16
typedef struct nmi_strategy_output {
    float Tcw[16];       /* pose after the search (restored to the input on failure) */
    int32_t relocalized; /* SetNMIRelocalized */
    int32_t failed;      /* SetNMIFailed      */
    int32_t iterations;  /* calls of eval_grid */
    int32_t stop_reason; /* 0 iteration cap, 1 best in the middle, 2 gain below 0.1 % twice */
    int32_t reverted_to_previous; /* final NMI below the previous iterate: pose of the previous iterate kept */
    float nmi_threshold_used;
    nmi_search_kernel kernel;      /* NmiKernel at exit     */
    nmi_search_kernel last_kernel; /* LastNmiKernel at exit */
    nmi_search_kernel per_iteration[NMI_STRATEGY_MAX_ITER]; /* NmiKernel after each eval (the _log.txt lines) */
} nmi_strategy_output;

/* Tracking::RelocalizeWithNMIStrategy (src/Tracking.cc:1987-2179) as a pure state machine over eval_grid. */
int nmi_relocalize_with_strategy(const nmi_strategy_input *in, const nmi_properties *props, nmi_eval_grid_fn eval_grid,
                                 void *user, nmi_strategy_output *out);

/*
 * Run-time configuration surface: the Camera.* / NMI.* keys of the reference's YAML settings file, read without OpenCV
 * (cv::FileStorage consumers: Thirdparty/Localization/localization.cpp:131-253, src/Tracking.cc:150-159;
 * Examples/Monocular/ETH_small.yaml:8-24,62-96).  Returns 0, or <0: -2 syntax, -3 Camera.* missing, -4 NMI grid key
 * missing, -5 file not readable.
 */
typedef struct nmi_config {
    int32_t width, height;            /* Camera.Width / Camera.Height  -> nmi_params.width / height            */
    double fx, fy, cx, cy;            /* Camera.fx..cy                 -> K of nmi_warp_homographies           */
    nmi_search_kernel initial;        /* NMI.SynthNum[XYZ], NMI.WarpNum[XYZ], NMI.SynthStep[XYZ], NMI.WarpStep[XYZ] */
    float nmi_threshold;              /* NMI.Treshold (sic)            -> nmi_strategy_input.nmi_threshold     */
    int32_t init_offset;              /* NMI.Offset: frame id of the second initialisation pose               */
    int32_t has_init1, has_init2;
    float init1[16], init2[16];       /* NMI.Init1 / NMI.Init2, 4x4 row-major (Tracking.cc:152-159)            */
    float render_point_size, render_near, render_far; /* NMI.Render.PointSize / NearPlane / FarPlane          */
    char render_object[512], render_texture[512], render_cloud[512], render_offset[512]; /* NMI.Render.* paths */
} nmi_config;
int nmi_config_parse(const char *text, size_t len, nmi_config *out);
int nmi_config_load(const char *yaml_path, nmi_config *out);

/*
 * Map files (the paths of nmi_config.render_object / render_texture / render_cloud / render_offset), read with the grammar and
 * the tolerances of the reference's loaders, into the host arrays the render producers of nmi_hip.h take after an upload
 * (and, once, nmi_sort_triangles / nmi_sort_points):
 *   nmi_map_load_obj  loadOBJ, objloader.cpp:140-224: "v", "vt", "f a/b c/d e/f" lines -> one vertex per face corner:
 *                     xyz float [n_vertices][3], uv float [n_vertices][2] (n_vertices = 3 * triangles) -- nmi_render_mesh
 *   nmi_map_load_xyz  loadXYZ, objloader.cpp:226-264: "x y z r g b" per point, minus the three numbers of the offset file
 *                     (in double precision), colour * 1/256 -> xyz float [n][3], red float [n] -- nmi_render_points;
 *                     rgb (may be NULL) float [n][3], the whole scaled colour
 *   nmi_map_load_bmp  loadBMP_custom, texture.cpp:31-86: 24-bit uncompressed BMP -> uint8 [height][width][3] in file order
 *                     (row 0 = v 0) -- nmi_texture_create
 * Buffers are malloc'ed; release each with nmi_map_free.  Returns 0, or <0: -1 argument, -2 not in the format (an OBJ face
 * that is not three position/texcoord pairs, a point with fewer than six numbers, a BMP that is not 24-bit uncompressed or is
 * shorter than its header says), -3 OBJ index outside the file's lists, -5 file not readable, -6 out of memory.
 */
int nmi_map_load_obj(const char *path, float **xyz, float **uv, int64_t *n_vertices);
int nmi_map_load_xyz(const char *path, const char *offset_path, float **xyz, float **red, float **rgb, int64_t *n_points);
int nmi_map_load_bmp(const char *path, uint8_t **rgb, int32_t *width, int32_t *height);
void nmi_map_free(void *p);

#ifdef __cplusplus
}
#endif
#endif /* NMI_HOST_H */
