// micro-benchmark: cost of a grid-wide barrier (256 persistent workgroups, device-scope atomics) on MI355X
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}
// two-level barrier: workgroups of an XCD (blockIdx & 7) meet on their own counter, the last one of each XCD meets
// the other XCDs on the global counter and then releases its XCD through a per-XCD generation flag
__device__ __forceinline__ void grid_barrier2(unsigned* ws, unsigned gen, int nosleep) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned xcd = blockIdx.x & 7, per = gridDim.x >> 3;
        unsigned* xc = ws + 32 * (1 + xcd);          // separate 128-byte lines
        unsigned* xf = ws + 32 * (9 + xcd);
        const unsigned old = __hip_atomic_fetch_add(xc, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == gen * per - 1) {                  // last of this XCD
            __hip_atomic_fetch_add(ws, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(ws, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gen * 8) { if (!nosleep) __builtin_amdgcn_s_sleep(1); }
            __hip_atomic_store(xf, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(xf, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gen) { if (!nosleep) __builtin_amdgcn_s_sleep(1); }
        }
    }
    __syncthreads();
}
__global__ __launch_bounds__(256) void k2(unsigned* ws, int nbar, float* data, int work, int nosleep) {
    float acc = 0.f;
    for (int b = 0; b < nbar; ++b) {
        for (int i = 0; i < work; ++i) acc += data[(blockIdx.x * 256 + threadIdx.x + i * 65536) & 0xfffff];
        grid_barrier2(ws, (unsigned)(b + 1), nosleep);
    }
    if (acc == 123.f) data[0] = acc;
}
__global__ __launch_bounds__(256) void k(unsigned* counter, int nbar, float* data, int work) {
    float acc = 0.f;
    for (int b = 0; b < nbar; ++b) {
        for (int i = 0; i < work; ++i) acc += data[(blockIdx.x * 256 + threadIdx.x + i * 65536) & 0xfffff];
        grid_barrier(counter, (unsigned)(b + 1) * gridDim.x);
    }
    if (acc == 123.f) data[0] = acc;
}
int main() {
    unsigned* c; float* d;
    hipMalloc(&c, 4); hipMalloc(&d, 4 << 20); hipMemset(d, 0, 4 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int work : {0, 4}) for (int nbar : {1, 101, 1001}) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipMemset(c, 0, 4);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, c, nbar, d, work);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("work %d nbar %4d: %.1f us total, %.2f us per barrier\n", work, nbar, best * 1e3, best * 1e3 / nbar);
    }
    unsigned* ws; hipMalloc(&ws, 4096);
    for (int nosleep : {0, 1}) for (int nbar : {101, 1001}) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipMemset(ws, 0, 4096);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k2, dim3(256), dim3(256), 0, 0, ws, nbar, d, 0, nosleep);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("two-level nosleep %d nbar %4d: %.1f us total, %.2f us per barrier\n", nosleep, nbar, best * 1e3, best * 1e3 / nbar);
    }
    return 0;
}
