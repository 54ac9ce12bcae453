// Rate of fp32 global atomic adds to DISTINCT addresses in the access pattern a single-pass attention backward would use for
// its dQ accumulation (encoder shape: 640 (batch, head) pairs x 1536 query rows x 64 floats, 12 key-block workgroups per
// pair, every workgroup adds one [64 q][64 d] tile per query tile: 755 M float atomics per launch).  The 120 G/s quoted in
// DESIGN section 13 came from LayerNorm's dgamma/dbeta atomics, where 256 workgroups hit the SAME 3 840 addresses.
//   mode 0: agent-scope atomicAdd, XCD-aware workgroup order (all 12 writers of a pair on one XCD)
//   mode 1: workgroup-scope atomic (no sc1: executes in the XCD's L2), XCD-aware order
//   mode 2: agent-scope, plain order (a pair's writers sprayed over the 8 XCDs)
//   mode 3: as 0 with ~2.5 us of dependent ALU work between query tiles (the spacing a real kernel has)
// Every mode verifies that each element received exactly 12 adds.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o tools/atomic_rate_probe tools/atomic_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define PAIRS 640
#define ROWS 1536
#define NKB 12

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* acc, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hi = lane >> 5, ln = lane & 31;
    const int n = gridDim.x, L = blockIdx.x;
    const int xcd = L & 7, slot = L >> 3, q = n >> 3, r = n & 7;
    const int logical = (MODE == 2) ? L : xcd * q + min(xcd, r) + slot;
    const int pair = logical / NKB;
    const int qb = wave >> 1, db = wave & 1;
    float* base = acc + (long)pair * ROWS * 64 + db * 32 + ln;
    float x = 1.0f + lane * 1e-9f;
    for (int qt = 0; qt < ROWS / 64; ++qt) {
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int row = qt * 64 + qb * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * hi;
            float* p = base + (long)row * 64;
            if (MODE == 1) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else atomicAdd(p, 1.0f);
        }
        if (MODE == 3) {
            for (int i = 0; i < 600; ++i) x = x * 1.0000001f + 1e-7f;
        }
    }
    if (x == 123.f) sink[0] = x;
}

template <int MODE>
static void run(float* acc, float* sink, const char* what) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t bytes = (size_t)PAIRS * ROWS * 64 * 4;
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipMemset(acc, 0, bytes);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<MODE>, dim3(PAIRS * NKB), dim3(256), 0, 0, acc, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    std::vector<float> h((size_t)PAIRS * ROWS * 64);
    hipMemcpy(h.data(), acc, bytes, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (float v : h) bad += (v != 12.0f);
    const double n_at = (double)PAIRS * NKB * (ROWS / 64) * 4 * 16 * 64;
    printf("mode %d (%s): %.3f ms  %.0f G float atomics/s  %.2f TB/s of 4-byte adds  wrong elements: %zu\n", MODE, what, best,
           n_at / best / 1e6, n_at * 4 / best / 1e9, bad);
}

int main() {
    float *acc, *sink;
    hipMalloc(&acc, (size_t)PAIRS * ROWS * 64 * 4); hipMalloc(&sink, 64);
    run<0>(acc, sink, "agent scope, XCD-aware order");
    run<1>(acc, sink, "workgroup scope, XCD-aware order");
    run<2>(acc, sink, "agent scope, plain order");
    run<3>(acc, sink, "agent scope, XCD-aware, spaced");
    return 0;
}
