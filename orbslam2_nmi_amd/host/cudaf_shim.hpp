// cudaf_shim.hpp -- source-compatible stand-in for Thirdparty/CUDA_Functions/kernel.cuh of the reference.
//
// Including this header instead of "kernel.cuh" lets the reference call site
//     CUDAF::NMIWithCuda_noMask((cv::cuda::PtrStep<unsigned char>*)image.data, SUC, MATCHING_NMI,
//                               width, height, &rating[...], renderedTexture);          (src/Tracking.cc:1886-1894)
// compile unchanged and run on libnmi_hip.so.  Differences the host has to provide for:
//   * the render arrives as an OpenGL texture name in the reference (cudaGraphicsGLRegisterImage per call,
//     kernel.cu:53-56).  Here the caller registers, once per render, the linear device buffer that holds the
//     texels of that texture:  CUDAF::RegisterRenderBuffer(textureName, devicePtr)   (see INTEGRATION.md for the
//     hipGraphicsGLRegisterImage + hipMemcpy2DFromArray lines that fill it);
//   * errors abort the process like checkCudaErrors does (kernel.cu:53-113), after printing the library's message.
// The first argument is a raw device pointer in disguise in the reference too (caller casts GpuMat::data,
// Tracking.cc:1887; callee casts back, kernel.cu:79); NMI_mode and MatchingMode are ignored there (kernel.cu:49)
// and here: the score form comes from the ENMI / SUC macros (kernel.cuh:22-23, NMI.cu:344,352).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <map>
#include <utility>
#include <vector>

#include "nmi_config.hpp"
#include "nmi_hip.h"

namespace cv {
namespace cuda {
template <typename T>
struct PtrStep;  // only ever used as an opaque pointer type by this call (Tracking.cc:1887)
}
}  // namespace cv

namespace CUDAF {

namespace detail {
struct State {
    std::map<unsigned int, const uint8_t *> renders;          // GL texture name -> linear device copy
    std::map<std::pair<int, int>, nmi_ctx *> contexts;        // one persistent workspace per frame size
    // BeginBatch() .. Flush(): calls are recorded here and scored together
    bool batching = false;
    int batch_w = 0, batch_h = 0;
    std::vector<const uint8_t *> batch_renders, batch_warps;
    std::vector<float *> batch_out;
    std::vector<float> batch_scores;
    // No destructor on purpose: this object dies during static destruction, possibly after the HIP runtime has shut
    // down; call CUDAF::Shutdown() to release the workspaces earlier (the reference frees per call, kernel.cu:103-113).
};
inline State &state()
{
    static State s;
    return s;
}
inline void die(int rc, const char *what, nmi_ctx *ctx)
{
    fprintf(stderr, "CUDAF shim: %s failed: %d (%s) %s\n", what, rc, nmi_error_string(rc), nmi_last_error_detail(ctx));
    exit(EXIT_FAILURE);  // checkCudaErrors semantics
}
inline nmi_ctx *context(int width, int height)
{
    auto &ctxs = state().contexts;
    auto it = ctxs.find({width, height});
    if (it != ctxs.end()) return it->second;
    nmi_params p;
    nmi_params_default(&p, width, height);
    p.mode = SUC ? NMI_MODE_SUC : NMI_MODE_ENMI;  // macro-selected in the reference (NMI.cu:344,352)
    if (!ENMI && !SUC) p.mode = NMI_MODE_SUC;
    p.use_bg = nmi_prop_BG ? 1 : 0;               // allProperties.hpp:38
    p.render_bottom_up = 1;                       // NMI.cu:82
    nmi_ctx *ctx = nullptr;
    const int rc = nmi_create(&p, &ctx);
    if (rc != NMI_OK) die(rc, "nmi_create", nullptr);
    ctxs[{width, height}] = ctx;
    return ctx;
}
}  // namespace detail

// Tell the shim where the texels of GL texture `syntGL` live on the device (uint8, width*height, bottom-up rows
// exactly as glReadPixels / the mapped cudaArray would give them).
inline void RegisterRenderBuffer(unsigned int syntGL, const unsigned char *d_render) { detail::state().renders[syntGL] = d_render; }
inline void UnregisterRenderBuffer(unsigned int syntGL) { detail::state().renders.erase(syntGL); }

// Releases every workspace the shim created (optional; safe to call at any point where no search is running).
inline void Shutdown()
{
    for (auto &kv : detail::state().contexts) nmi_destroy(kv.second);
    detail::state().contexts.clear();
    detail::state().renders.clear();
}

// Optional batching for hosts that can add two lines around the inner loop of Tracking::RelocalizeWithNMI
// (src/Tracking.cc:1883-1894):
//     CUDAF::BeginBatch();
//     for (wX..) for (wY..) for (wZ..) CUDAF::NMIWithCuda_noMask(..., &rating[wZ][wY][wX][sZ][sY][sX], renderedTexture);
//     CUDAF::Flush();        // before the next renderToTextureOnGPU overwrites the texture
// Between the two, NMIWithCuda_noMask only records its arguments; Flush scores all recorded candidates with one
// nmi_eval_pairs call (one launch for the 27 warps of a default grid) and writes every *NMI.  Same values as unbatched.
inline void BeginBatch()
{
    detail::State &st = detail::state();
    st.batching = true;
    st.batch_renders.clear();
    st.batch_warps.clear();
    st.batch_out.clear();
}
inline void Flush()
{
    detail::State &st = detail::state();
    st.batching = false;
    const int n = (int)st.batch_out.size();
    if (n == 0) return;
    nmi_ctx *ctx = detail::context(st.batch_w, st.batch_h);
    st.batch_scores.resize((size_t)n);
    const int rc = nmi_eval_pairs(ctx, st.batch_renders.data(), st.batch_warps.data(), n, st.batch_scores.data());
    if (rc != NMI_OK) detail::die(rc, "nmi_eval_pairs", ctx);
    for (int i = 0; i < n; ++i) *st.batch_out[(size_t)i] = st.batch_scores[(size_t)i];
    st.batch_renders.clear();
    st.batch_warps.clear();
    st.batch_out.clear();
}

// Identical signature to kernel.cuh:37.
inline void NMIWithCuda_noMask(cv::cuda::PtrStep<unsigned char> *d_Warped, int /*NMI_mode*/, int /*MatchingMode*/, int width,
                               int height, float *NMI, unsigned int syntGL)
{
    auto it = detail::state().renders.find(syntGL);
    if (it == detail::state().renders.end()) detail::die(NMI_ERR_INVALID_ARGUMENT, "lookup of the render buffer", nullptr);
    detail::State &st = detail::state();
    if (st.batching) {
        if (!st.batch_out.empty() && (width != st.batch_w || height != st.batch_h)) Flush(), st.batching = true;  // one frame size per batch
        st.batch_w = width;
        st.batch_h = height;
        st.batch_renders.push_back(it->second);
        st.batch_warps.push_back(reinterpret_cast<const uint8_t *>(d_Warped));
        st.batch_out.push_back(NMI);
        return;
    }
    nmi_ctx *ctx = detail::context(width, height);
    const int rc = nmi_eval_pair(ctx, it->second, reinterpret_cast<const uint8_t *>(d_Warped), NMI);
    if (rc != NMI_OK) detail::die(rc, "nmi_eval_pair", ctx);
}

}  // namespace CUDAF
