// Emulates a communication kernel that keeps some CUs busy: `nwg` workgroups of 256 threads spin for `usec` microseconds.
// Built as a tiny shared library used by tools/comm_contention.py (experiment, not part of libdwamd.so).
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(256) void hog_kernel(long long ticks, float* sink) {
    const long long t0 = wall_clock64();
    float acc = 0.f;
    while (wall_clock64() - t0 < ticks) { acc += 1.f; __builtin_amdgcn_s_sleep(8); }
    if (acc < 0.f) sink[0] = acc;
}
extern "C" int hog_launch(int nwg, int usec, float* sink, void* stream) {
    // wall_clock64 ticks at 100 MHz on gfx9
    hipLaunchKernelGGL(hog_kernel, dim3(nwg), dim3(256), 65536, (hipStream_t)stream, (long long)usec * 100, sink);
    return (int)hipGetLastError();
}
