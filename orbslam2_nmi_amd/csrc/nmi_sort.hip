// nmi_sort.hip -- puts a map (point cloud or triangle soup) into Morton order, on the device.
//
// The renderers' results do not depend on the order of the primitives (the depth test is a minimum), their speed does,
// twice over: culling works on runs of neighbours in memory (64 points, 256 triangles: a run that is compact in space is
// outside most views as a whole), and the anchors of neighbouring points land in neighbouring words of the depth buffer,
// so a wavefront's 64 atomics touch a few cache lines instead of 64.  A 3 M-point cloud rendered into 27 views at
// 848x480: 96 us in scan order, 406 us shuffled, 116 us shuffled and then sorted here (tools/cloud_order_time.py).
// A map loader calls this once (loadXYZ / loadOBJ give file order, objloader.cpp:140-264).
//
// Key = 30-bit Morton code of the primitive's position (a triangle's centroid) in the bounding box of the finite
// positions; primitives with a non-finite coordinate go last.  rocPRIM's radix sort orders (key, index) pairs; a gather
// writes the records.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "nmi_kernels.h"

namespace nmi {

namespace {

// order-preserving map float -> uint32 (for atomicMin / atomicMax on floats)
__device__ __forceinline__ uint32_t ordered(float f)
{
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float unordered(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

// position of record i: a point, or the centroid of 3 consecutive vertices
__device__ __forceinline__ void position(const float *__restrict__ xyz, long long i, int verts, float &x, float &y, float &z)
{
    const float *p = xyz + (size_t)i * 3 * verts;
    x = p[0], y = p[1], z = p[2];
    for (int v = 1; v < verts; ++v) x += p[3 * v], y += p[3 * v + 1], z += p[3 * v + 2];
    if (verts > 1) x /= (float)verts, y /= (float)verts, z /= (float)verts;
}

__global__ __launch_bounds__(256) void nmi_box_kernel(const float *__restrict__ xyz, long long n, int verts, uint32_t *box /*[6]: min xyz, max xyz (ordered)*/)
{
    uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float p[3];
        position(xyz, i, verts, p[0], p[1], p[2]);
        if (!(fabsf(p[0]) <= 3.0e38f && fabsf(p[1]) <= 3.0e38f && fabsf(p[2]) <= 3.0e38f)) continue;  // NaN / inf
        for (int k = 0; k < 3; ++k) {
            const uint32_t o = ordered(p[k]);
            lo[k] = min(lo[k], o);
            hi[k] = max(hi[k], o);
        }
    }
    for (int k = 0; k < 3; ++k) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = min(lo[k], (uint32_t)__shfl_xor((int)lo[k], off, 64));
            hi[k] = max(hi[k], (uint32_t)__shfl_xor((int)hi[k], off, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&box[k], lo[k]);
            atomicMax(&box[3 + k], hi[k]);
        }
    }
}

__device__ __forceinline__ uint32_t spread3(uint32_t v)  // 10 bits -> every third bit
{
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ __launch_bounds__(256) void nmi_morton_kernel(const float *__restrict__ xyz, long long n, int verts, const uint32_t *__restrict__ box,
                                                         uint32_t *__restrict__ keys, uint32_t *__restrict__ index)
{
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    float p[3];
    position(xyz, i, verts, p[0], p[1], p[2]);
    uint32_t code = 0x40000000u;  // after every finite one
    if (fabsf(p[0]) <= 3.0e38f && fabsf(p[1]) <= 3.0e38f && fabsf(p[2]) <= 3.0e38f) {
        code = 0;
        for (int k = 0; k < 3; ++k) {
            const float lo = unordered(box[k]), hi = unordered(box[3 + k]);
            const float span = hi - lo;
            float t = span > 0.0f ? (p[k] - lo) / span : 0.0f;
            t = fminf(fmaxf(t, 0.0f), 1.0f);
            code |= spread3((uint32_t)(t * 1023.0f)) << k;
        }
    }
    keys[i] = code;
    index[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void nmi_gather_kernel(const uint32_t *__restrict__ index, long long n, const float *__restrict__ a, int na,
                                                         float *__restrict__ a_out, const float *__restrict__ b, int nb, float *__restrict__ b_out)
{
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t src = index[i];
    for (int k = 0; k < na; ++k) a_out[(size_t)i * na + k] = a[src * na + k];
    for (int k = 0; k < nb; ++k) b_out[(size_t)i * nb + k] = b[src * nb + k];
}

}  // namespace

// Records i = 0..n-1: `a` holds na floats per record, the first 3 * verts of which are its vertices; `b` holds nb floats per
// record (may be null with nb = 0).  Everything on `stream`; returns after the stream has drained (temporary storage).
hipError_t sort_records_morton(const float *a, int na, int verts, const float *b, int nb, long long n, float *a_out, float *b_out,
                               hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    if (n >= (1ll << 32)) return hipErrorInvalidValue;
    uint32_t *box = nullptr, *keys = nullptr, *keys2 = nullptr, *idx = nullptr, *idx2 = nullptr;
    void *temp = nullptr;
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t r) {
        if (e == hipSuccess) e = r;
        return e == hipSuccess;
    };
    const size_t nb4 = (size_t)n * sizeof(uint32_t);
    ok(hipMalloc((void **)&box, 6 * sizeof(uint32_t))) && ok(hipMalloc((void **)&keys, nb4)) && ok(hipMalloc((void **)&keys2, nb4)) &&
        ok(hipMalloc((void **)&idx, nb4)) && ok(hipMalloc((void **)&idx2, nb4));
    if (e == hipSuccess) {
        const uint32_t init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
        ok(hipMemcpyAsync(box, init, sizeof init, hipMemcpyHostToDevice, stream));
        ok(hipStreamSynchronize(stream));  // `init` lives on this stack
    }
    if (e == hipSuccess) {
        const unsigned blocks = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(nmi_box_kernel, dim3(blocks < 2048 ? blocks : 2048), dim3(256), 0, stream, a, n, verts, box);
        hipLaunchKernelGGL(nmi_morton_kernel, dim3(blocks), dim3(256), 0, stream, a, n, verts, box, keys, idx);
        ok(hipGetLastError());
        size_t temp_bytes = 0;
        ok(rocprim::radix_sort_pairs(nullptr, temp_bytes, keys, keys2, idx, idx2, (size_t)n, 0, 31, stream));
        ok(hipMalloc(&temp, temp_bytes ? temp_bytes : 16));
        if (e == hipSuccess) ok(rocprim::radix_sort_pairs(temp, temp_bytes, keys, keys2, idx, idx2, (size_t)n, 0, 31, stream));
        if (e == hipSuccess) {
            hipLaunchKernelGGL(nmi_gather_kernel, dim3(blocks), dim3(256), 0, stream, idx2, n, a, na, a_out, b, b ? nb : 0, b_out);
            ok(hipGetLastError());
        }
        const hipError_t es = hipStreamSynchronize(stream);
        ok(es);
    }
    for (void *p : {(void *)box, (void *)keys, (void *)keys2, (void *)idx, (void *)idx2, temp})
        if (p) (void)hipFree(p);
    return e;
}

}  // namespace nmi
