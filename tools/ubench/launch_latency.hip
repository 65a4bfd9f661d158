// launch_latency.hip -- round trip of one blocking launch (hipLaunchKernel -> the kernel's last lane stores a word to pinned
// host memory -> the host sees it), for kernels with and without scratch (private segment: asked for only, or used), with little and much LDS, 32 and
// 256 workgroups of 1024 lanes: what an nmi_eval_pair call pays besides its kernel's work.
// hipcc --offload-arch=gfx950 -O2 -o launch_latency launch_latency.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <chrono>

template <int SCRATCH, int LDS_BYTES>
__global__ __launch_bounds__(1024) void k(unsigned long long *post, unsigned long long seq, unsigned int *ticket, int idx)
{
    __shared__ uint32_t lds[LDS_BYTES / 4];
    if (LDS_BYTES > 4 && threadIdx.x == 0) lds[idx & 3] = idx;
    if (SCRATCH == 1 || (SCRATCH == 2 && idx == 12345)) {  // 2: the kernel asks for a private segment but never touches it
        volatile uint32_t a[16];
        for (int i = 0; i < 16; ++i) a[i] = i + idx;
        if (a[idx & 15] == 0xDEAD) post[1] = 1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int n = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (n == gridDim.x - 1) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(post, seq + lds[LDS_BYTES > 4 ? (idx & 3) : 0] * 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <int SCRATCH, int LDS_BYTES>
static void run(const char *name, int grid, hipStream_t st, unsigned long long *post, unsigned int *ticket)
{
    unsigned long long seq = 1;
    auto once = [&]() {
        ++seq;
        hipLaunchKernelGGL((k<SCRATCH, LDS_BYTES>), dim3(grid), dim3(1024), 0, st, post, seq, ticket, 3);
        while (__atomic_load_n(post, __ATOMIC_ACQUIRE) != seq) {}
    };
    for (int i = 0; i < 200; ++i) once();
    const int n = 2000;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) once();
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
    printf("%-34s grid %3d: %6.2f us per blocking launch\n", name, grid, us);
}

int main()
{
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    unsigned long long *post;
    hipHostMalloc((void **)&post, 64, hipHostMallocCoherent | hipHostMallocMapped);
    post[0] = post[1] = 0;
    unsigned int *ticket;
    hipMalloc((void **)&ticket, 4);
    hipMemset(ticket, 0, 4);
    hipDeviceSynchronize();
    for (int grid : {1, 32, 256}) {
        run<0, 4>("no scratch, no LDS", grid, st, post, ticket);
        run<2, 4>("scratch asked for, untouched", grid, st, post, ticket);
        run<1, 4>("scratch written and read", grid, st, post, ticket);
        run<0, 86016>("no scratch, 84 KiB LDS", grid, st, post, ticket);
        run<0, 155648>("no scratch, 152 KiB LDS", grid, st, post, ticket);
        run<2, 155648>("scratch untouched, 152 KiB LDS", grid, st, post, ticket);
    }
    return 0;
}
