// How long a wave waits for the acknowledgement of its epilogue-sized store burst (vmcnt retires in order and counts stores
// on gfx950: a wave that has stored cannot complete a wait for a later load before this).  One 512-thread workgroup per CU;
// every storing wave writes 16 KiB (8 x 16-byte stores per lane, row-contiguous like the GEMM epilogue), then s_waitcnt vmcnt(0).
//   mode 0: all 8 waves of every CU store at once (the lockstep tile round of the persistent GEMM: 128 KiB per CU)
//   mode 1: waves 0-3 store (64 KiB per CU), waves 4-7 idle
//   mode 2: waves 0-3 store while waves 4-7 stream operand DMA (global -> LDS, 64 KiB per ~1.5 us per CU) the whole time
//   mode 3: as 2, and only every second CU's storers store in this round (the other half stores half a period later)
// Reported: microseconds from the first store issue to the completed wait, per storing wave (mean / p50 / p95 / max).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/store_ack_probe tools/store_ack_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) void lds_void_t;

__global__ __launch_bounds__(512) void probe(char* out, const char* src, long long* stamps, int mode, int rounds) {
    __shared__ __attribute__((aligned(1024))) char lds[65536];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cu = blockIdx.x;
    const bool storer = mode == 0 ? true : wave < 4;
    const bool loader = mode >= 2 && wave >= 4;
    for (int r = 0; r < rounds; ++r) {
        __syncthreads();
        if (loader) {
            // ~4.5 us of DMA: 3 x 64 KiB per CU (16 pieces of 1 KiB per loader wave per 64 KiB)
            for (int rep = 0; rep < 3; ++rep) {
                for (int i = 0; i < 16; ++i) {
                    const char* g = src + ((long)((cu * 3 + rep) & 1023) * 65536 + ((wave - 4) * 16 + i) * 1024 + lane * 16);
                    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lds_void_t*)(lds + ((wave - 4) * 16 + i) * 1024));
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(m0v), "v"(g) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        const bool active = storer && (mode != 3 || ((cu ^ r) & 1) == 0);
        if (active) {
            u32x4 v = {(unsigned)lane, (unsigned)wave, (unsigned)cu, (unsigned)r};
            // 8 rows of 1 KiB per instruction-group: lane -> 16 bytes, 64 lanes contiguous; wave's 16 KiB region per round slot
            char* base = out + ((long)cu * 8 + wave) * 16384 + ((long)(r & 7) * 256 * 8 * 16384);
            const long long t0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
            for (int i = 0; i < 16; ++i) *(u32x4*)(base + i * 1024 + lane * 16) = v;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const long long t1 = __builtin_amdgcn_s_memrealtime();
            if (lane == 0) stamps[((long)r * 256 + cu) * 8 + wave] = t1 - t0;
        } else if (storer && lane == 0) stamps[((long)r * 256 + cu) * 8 + wave] = -1;
    }
}

int main() {
    const int rounds = 24;
    char *out, *src; long long* stamps;
    hipMalloc(&out, 8L * 256 * 8 * 16384); hipMalloc(&src, 1024L * 65536); hipMalloc(&stamps, (long)rounds * 256 * 8 * 8);
    hipMemset(src, 1, 1024L * 65536);
    std::vector<long long> h((size_t)rounds * 256 * 8);
    for (int mode = 0; mode < 4; ++mode) {
        hipMemset(stamps, 0xff, (long)rounds * 256 * 8 * 8);
        hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, out, src, stamps, mode, rounds);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> us;
        for (int r = 8; r < rounds; ++r)            // skip the first rounds (cold)
            for (int i = 0; i < 256 * 8; ++i) { long long t = h[(size_t)r * 256 * 8 + i]; if (t > 0) us.push_back(t * 0.01); }
        std::sort(us.begin(), us.end());
        double mean = 0; for (double x : us) mean += x; mean /= us.size();
        printf("mode %d: %zu samples  mean %.2f us  p50 %.2f  p95 %.2f  max %.2f\n", mode, us.size(), mean, us[us.size() / 2],
               us[(size_t)(us.size() * 0.95)], us.back());
    }
    return 0;
}
