// wait_value.hip -- can the launch latency of a blocking call be taken off its critical path by enqueueing the kernel AHEAD,
// behind a hipStreamWaitValue64 that the host releases with one store when the call's arguments are known?
// Measures, for 32 workgroups of 1024 empty lanes: (a) plain launch -> post seen by the host; (b) kernel pre-enqueued behind
// a wait: release store -> post seen; (c) sustained call period with two calls enqueued ahead.
// hipcc --offload-arch=gfx950 -O2 -o wait_value wait_value.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <chrono>

__global__ __launch_bounds__(1024) void k(unsigned long long *post, const unsigned long long *args, unsigned int *ticket)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int n = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (n == gridDim.x - 1) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long seq = __hip_atomic_load(args, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // the "arguments", read at run time
            __hip_atomic_store(post, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
    int can = 0;
    CHECK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned long long *post, *args;
    CHECK(hipHostMalloc((void **)&post, 64, hipHostMallocCoherent | hipHostMallocMapped));
    CHECK(hipHostMalloc((void **)&args, 64, hipHostMallocCoherent | hipHostMallocMapped));
    *post = *args = 0;
    unsigned int *ticket;
    CHECK(hipMalloc((void **)&ticket, 4));
    CHECK(hipMemset(ticket, 0, 4));
    uint64_t *sig = nullptr;
    CHECK(hipExtMallocWithFlags((void **)&sig, 8, hipMallocSignalMemory));
    *sig = 0;  // signal memory is host-accessible
    CHECK(hipDeviceSynchronize());
    const int grid = 32, n = 2000;
    using clk = std::chrono::steady_clock;
    auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    unsigned long long seq = 0;
    // (a) plain
    for (int i = 0; i < 200 + n; ++i) {
        static clk::time_point t0;
        if (i == 200) t0 = clk::now();
        ++seq;
        __atomic_store_n(args, seq, __ATOMIC_RELEASE);
        hipLaunchKernelGGL(k, dim3(grid), dim3(1024), 0, st, post, args, ticket);
        while (__atomic_load_n(post, __ATOMIC_ACQUIRE) != seq) {}
        if (i == 200 + n - 1) printf("(a) launch -> post, per call:                          %6.2f us\n", us(t0, clk::now()) / n);
    }
    // (b) pre-enqueued behind a wait; the enqueue itself is outside the timed part
    double acc = 0;
    for (int i = 0; i < 200 + n; ++i) {
        ++seq;
        CHECK(hipStreamWaitValue64(st, sig, seq, hipStreamWaitValueGte, ~0ull));
        hipLaunchKernelGGL(k, dim3(grid), dim3(1024), 0, st, post, args, ticket);
        for (volatile int spin = 0; spin < 20000; ++spin) {}  // let the queue reach the wait
        const auto t0 = clk::now();
        __atomic_store_n(args, seq, __ATOMIC_RELEASE);
        __atomic_store_n(sig, seq, __ATOMIC_RELEASE);
        while (__atomic_load_n(post, __ATOMIC_ACQUIRE) != seq) {}
        if (i >= 200) acc += us(t0, clk::now());
    }
    printf("(b) release store -> post (kernel enqueued ahead):      %6.2f us\n", acc / n);
    // (c) sustained: two calls enqueued ahead; each iteration releases one and enqueues another
    unsigned long long enq = seq;
    for (int d = 0; d < 2; ++d) {
        ++enq;
        CHECK(hipStreamWaitValue64(st, sig, enq, hipStreamWaitValueGte, ~0ull));
        hipLaunchKernelGGL(k, dim3(grid), dim3(1024), 0, st, post, args, ticket);
    }
    clk::time_point t0;
    for (int i = 0; i < 200 + n; ++i) {
        if (i == 200) t0 = clk::now();
        ++seq;
        __atomic_store_n(args, seq, __ATOMIC_RELEASE);
        __atomic_store_n(sig, seq, __ATOMIC_RELEASE);
        ++enq;  // enqueue the call after next while this one runs
        CHECK(hipStreamWaitValue64(st, sig, enq, hipStreamWaitValueGte, ~0ull));
        hipLaunchKernelGGL(k, dim3(grid), dim3(1024), 0, st, post, args, ticket);
        while (__atomic_load_n(post, __ATOMIC_ACQUIRE) != seq) {}
    }
    printf("(c) sustained, two calls enqueued ahead, per call:      %6.2f us\n", us(t0, clk::now()) / n);
    // drain the two waiting kernels
    __atomic_store_n(sig, enq, __ATOMIC_RELEASE);
    CHECK(hipStreamSynchronize(st));
    return 0;
}
