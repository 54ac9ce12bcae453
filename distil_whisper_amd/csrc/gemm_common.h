// Shared pieces of the bf16 MFMA GEMM kernel (gemm.hip): parameter block, k-major fragment fetch, fused epilogue.
#pragma once
#include "common.h"
#include "../../include/dwamd.h"

struct GemmP {
    const bf16* a;
    const bf16* b;
    void* c;
    const float* bias;
    bf16* z_out;
    const bf16* zgrad;
    const void* r;
    long lda, ldb, ldc, ldz, ldzg, ldr;
    int m, n, k;
    int act, c_dtype, r_dtype, r_row_mod, round_res;
    int tiles_n, nwg;      // output tiles
    int strip;             // rasterisation strip width in tiles (see the kernel)
    int split_k, atomic;   // K slices per tile (grid = nwg * split_k); atomic: C (f32) += v with atomics
    int vec;               // all epilogue pointers / leading dimensions allow 4-wide vector access
    long slice_stride;     // split-K without atomics: slice ks stores its partial tile at c + ks * slice_stride
    // decode-step fusions (skinny-M kernel only, see gemm_skinny.hip)
    const void* ln_x;      // A = bf16(LayerNorm(ln_x)) built on load (a is ignored); f32 or bf16 per ln_x_dtype
    const float* ln_g;
    const float* ln_b;
    long ld_lnx;
    int ln_x_dtype;
    float ln_eps;
    bf16* kv_out;          // output columns >= kv_split of row m go to the K/V cache instead of C:
    long kv_ld;            //   kv_out[((m / kv_rpb) * kv_pitch + kv_row0 + m % kv_rpb) * kv_ld + (n - kv_split)]
    int kv_split, kv_rpb, kv_pitch, kv_row0;
    int stage_next;        // debug key 11; bit 4 (16): the software-pipelined kernels skip the epilogue (profiling: tools/gemm_overhead.py)
    int zg_f16;            // z_out / zgrad hold gelu'(z) in fp16 instead of z in bf16 (see DwGemm.z_is_gelu_grad)
    int* sched;            // persistent kernels: 9 device counters of this stream (dynamic job hand-out), or null
    int stagger;           // debug key 12: start offsets of the persistent workgroups (gemm_wp.h), 0 = none
    float* colsum;         // f32 [n] += column sums of the stored C (DwGemm.colsum_out), or null
    long long* trace;      // debug keys 13 / 14 (low / high half of a device pointer): per-workgroup phase timestamps of
                           // the first 8 tiles (gemm_wp.h DW_TRACE; tools/gemm_phase_trace.py), null = off
};

// Persistent workgroups: the grid holds at most one workgroup per CU; each walks a list of output-tile jobs.
// The stores of tile i are still draining while the operand loads and MFMAs of tile i+1 run (a wave cannot
// retire before its stores are acknowledged, so one-tile workgroups expose the whole write burst).
// XCD-aware: the dispatcher places block b on XCD b%8; every XCD owns a contiguous range of the job list.
__device__ __forceinline__ void gemm_job_range(const GemmP& p, int& job_first, int& job_count, int& job_step) {
    const int total = p.nwg * p.split_k;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, local = bid >> 3;
    const int q = total >> 3, r = total & 7;
    if ((int)gridDim.x == total) {           // one job per workgroup (small problems)
        job_first = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
        job_count = 1;
        job_step = 1;
    } else {                                 // gridDim.x is a multiple of 8: gridDim.x / 8 workgroups per XCD
        const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        const int cnt = q + (xcd < r ? 1 : 0);
        job_step = gridDim.x >> 3;
        job_first = start + local;
        job_count = local < cnt ? (cnt - local + job_step - 1) / job_step : 0;
    }
}

// Job stream of a persistent workgroup.  Static mode (sched == null, or one job per workgroup): the k-th job of local
// workgroup l of an XCD is job l + k * (workgroups per XCD) of the XCD's range.  Dynamic mode: the workgroups of an XCD
// draw consecutive jobs of that range from a device counter (one agent-scope atomic per tile, requested a whole tile
// ahead).  The order in which an XCD's CUs walk the strip is the same, but a workgroup that starts late -- because a
// communication kernel or a kernel of another stream holds its CU -- no longer owns a fixed share of the tiles: with 8
// of the 256 CUs taken the step went from 441 to 595 ms with the static hand-out (the 8 late workgroups run their 15
// tiles after everybody else has finished), 441 -> ~455 ms with the dynamic one (tools/comm_contention.py).  The last
// workgroup to leave resets the counters for the next launch on the stream.
struct GemmJobs {
    int start, cnt;        // this XCD's job range [start, start + cnt)
    int step, local;       // static mode
    int cur;               // offset of the current job in the range (>= cnt: none left)
    int iter;
    bool dynamic;
};
__device__ __forceinline__ int gemm_jobs_fetch(const GemmP& p) {
    return __hip_atomic_fetch_add(p.sched + (blockIdx.x & 7), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// `slot`: two ints of LDS.  Returns with J.cur = the first job of this workgroup (one barrier in dynamic mode).
__device__ __forceinline__ void gemm_jobs_begin(const GemmP& p, GemmJobs& J, int* slot) {
    const int total = p.nwg * p.split_k;
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int q = total >> 3, r = total & 7;
    J.start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    J.iter = 0;
    if ((int)gridDim.x == total) {           // one job per workgroup (small problems)
        J.cnt = q + (xcd < r ? 1 : 0);
        J.local = bid >> 3; J.step = J.cnt > 0 ? J.cnt : 1; J.cur = J.local; J.dynamic = false;
        J.cnt = J.local < J.cnt ? J.local + 1 : 0;      // exactly one job: offsets [local, local + 1)
        return;
    }
    J.cnt = q + (xcd < r ? 1 : 0);
    J.step = gridDim.x >> 3;
    J.local = bid >> 3;
    J.dynamic = p.sched != nullptr;
    if (J.dynamic) {
        if (threadIdx.x == 0) slot[0] = gemm_jobs_fetch(p);
        __syncthreads();
        J.cur = slot[0];
    } else {
        J.cur = J.local;
    }
}
// at the top of a tile: request the job after this one (dynamic mode; the answer is read a whole tile later)
__device__ __forceinline__ void gemm_jobs_prefetch(const GemmP& p, GemmJobs& J, int* slot) {
    if (J.dynamic && threadIdx.x == 0) slot[(J.iter + 1) & 1] = gemm_jobs_fetch(p);
}
// after the barrier that ends a tile
__device__ __forceinline__ void gemm_jobs_advance(GemmJobs& J, const int* slot) {
    ++J.iter;
    J.cur = J.dynamic ? slot[J.iter & 1] : J.cur + J.step;
}
__device__ __forceinline__ void gemm_jobs_end(const GemmP& p, const GemmJobs& J) {
    if (J.dynamic && threadIdx.x == 0) {
        const int done = __hip_atomic_fetch_add(p.sched + 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == (int)gridDim.x - 1) {
#pragma unroll
            for (int i = 0; i < 9; ++i) __hip_atomic_store(p.sched + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// job id -> (row tile, column tile, K slice).  Slice-major (neighbouring blocks share operand panels in L2), then
// tile rasterisation: the 32 CUs of an XCD walk column strips of `sw` output tiles (strip-major, then down M),
// so a strip of B (sw x BN x K, <= ~2.6 MB for K = 1280) stays resident in the XCD's 4 MiB L2 while A streams
// through once per strip and every A panel is shared by sw concurrently running workgroups.
__device__ __forceinline__ void gemm_job_decode(const GemmP& p, int id, int& tm, int& tn, int& ks) {
    ks = id / p.nwg;
    id -= ks * p.nwg;
    const int tiles_m = p.nwg / p.tiles_n;
    const int nstrips = (p.tiles_n + p.strip - 1) / p.strip;
    const int sw = (p.tiles_n + nstrips - 1) / nstrips;          // balanced strip width
    const int per_strip = sw * tiles_m;
    int strip = id / per_strip;
    int within = id - strip * per_strip;
    int width = sw;
    const int full = p.tiles_n - (nstrips - 1) * sw;             // width of the last (possibly narrower) strip
    if (strip >= nstrips - 1) {                                  // ids past the full strips belong to the last
        strip = nstrips - 1;
        within = id - strip * per_strip;
        width = full;
    }
    tm = within / width;
    tn = strip * sw + (within - tm * width);
}

// Rasterisation strip width (in 256-wide column tiles) for the persistent 256x256 kernels: as many B tiles as fit a
// budget of the XCD's 4 MiB L2 (g_gemm_strip_budget, units of 512 KiB), the whole row when fewer than 2 fit or the row is
// not wider than that.  `override` > 0 (dw_debug_set key 1) forces a width.
extern int g_gemm_strip_budget;
// CUs the persistent 256-tile kernels occupy (multiple of 8, default all 256; dw_debug_set key 9).  With a communication
// kernel resident on some CUs a 256-workgroup grid would need a second round for the workgroups that did not fit.
extern int g_gemm_cus;
inline int gemm_strip_width(int k, int tiles_n, int override_) {
    if (override_ > 0) return override_;
    const long tile_bytes = 256L * k * 2;
    int sw = (int)(((long)g_gemm_strip_budget << 19) / tile_bytes);
    if (sw < 2 || sw >= tiles_n) sw = tiles_n;
    return sw;
}

// transposed (k-major) tile [64][BX]: fragment X^T[i = x + ...][k-slots] for one 16-deep k step
template <int BX>
__device__ __forceinline__ bf16x8 frag_kmajor(const char* tile, int x, int kk, int lane) {
    constexpr int RB = BX * 2;
    const int g = lane >> 4, p = lane & 15;
    const int col = x + ((g & 1) << 4) + ((p & 3) << 2);
    const int k0 = kk * 16 + ((g >> 1) << 3) + (p >> 2);
    const int ls = col >> 3;
    const int inb = (p & 1) << 3;
    const int sw = ((p >> 2) & 3) << 2;  // (krow & 3) << 2 ; krow & 3 == p >> 2 for both reads
    const char* a0 = tile + k0 * RB + ((ls ^ sw) << 4) + inb;
    const char* a1 = a0 + 4 * RB;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)a0);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)a1);
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}


// One epilogue row group: 4 consecutive output columns of one row, all fused element-wise work of the GEMM (bias, Z
// store, GELU, GELU' with the prefetched z, residual add with the prefetched r, C store).  `v` holds the fp32
// accumulator values; the caller guarantees the 4 columns are in range and every pointer allows 4-wide accesses.
// `cdst` / `zdst`: addresses of the 4 output elements in C (element size per c_dtype) and in z_out (2-byte elements).
// F >= 0: the epilogue flavour is a compile-time bit set (EPI_*) and every test below folds away; F < 0: the tests read the
// parameter block at run time (general walk, rare flavours).  The unrolled interior walk is instantiated per flavour and
// the flavour is chosen ONCE per tile (gemm_epilogue): the run-time version inlined 32 times per wave tile put the GELU
// arithmetic, both residual forms and every store variant into each of the 32 row groups -- ~16 000 instructions
// (>100 KB) of which a plain epilogue executed ~400, scattered over the whole range behind 600 branches: more than the
// 64 KiB instruction cache two CUs share, re-fetched from L2 every tile.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
enum { EPI_BIAS = 1, EPI_ZBF16 = 2, EPI_GELU = 4, EPI_STOREG = 8, EPI_ZG16 = 16, EPI_ZGBF = 32, EPI_RES = 64, EPI_RES_F32 = 128,
       EPI_ROUND = 256, EPI_OUT_F32 = 512, EPI_COLSUM = 1024 };
__device__ __forceinline__ int gemm_epi_flavour(const GemmP& p) {
    return (p.bias ? EPI_BIAS : 0) | ((p.z_out && !p.zg_f16) ? EPI_ZBF16 : 0) | (p.act == 1 ? EPI_GELU : 0) |
           ((p.z_out && p.zg_f16) ? EPI_STOREG : 0) | ((p.zgrad && p.zg_f16) ? EPI_ZG16 : 0) |
           ((p.zgrad && !p.zg_f16) ? EPI_ZGBF : 0) | (p.r ? EPI_RES : 0) | ((p.r && p.r_dtype == DW_F32) ? EPI_RES_F32 : 0) |
           ((p.r && p.round_res) ? EPI_ROUND : 0) | (p.c_dtype == DW_F32 ? EPI_OUT_F32 : 0) | (p.colsum ? EPI_COLSUM : 0);
}
// Stores of the 2-byte outputs (C in bf16, z_out) carry the non-temporal hint.  An output of hundreds of MB written through L2 /
// the Infinity Cache the default way evicts the operand panels the other workgroups are still reading: sustained, the fc1 shape
// (N = 5120) gains 4-13 %, fc2 1 %, qkv loses 1 % (tools/gemm_w4_probe.py); the whole step -1.35 % in one process against a
// build with -DDW_EPI_NT=0 (tools/ab_step.py, tools/build_variant_lib.sh).  fp32 outputs (the residual stream, the weight-gradient
// slabs) are read again by the next kernel and keep the default policy (NT there: +0.3 %); so do LayerNorm's and the attention
// kernels' outputs (NT: +0.5 % / +2.9 % -- their consumers run right behind them).
// Compile time on purpose: with a per-launch flag both store forms sit in every walk and the larger image costs more than the
// hint returns (and `if (nt) __builtin_nontemporal_store(..) else *p = v` is merged into ONE plain store: the hint is metadata).
#ifndef DW_EPI_NT
#define DW_EPI_NT 1
#endif
template <class T>
__device__ __forceinline__ void gemm_store_out(T* dst, const T& v) {
#if DW_EPI_NT
    __builtin_nontemporal_store(v, dst);
#else
    *dst = v;
#endif
}
template <int F = -1>
__device__ __forceinline__ void gemm_epi_vec4(const GemmP& p, float (&v)[4], const f32x4& b4, bool plain, bool have_side,
                                              const bf16x4& zs, const f32x4& rs, char* cdst, char* zdst, float (&cs)[4]) {
    constexpr bool RT = F < 0;
    const bool any = RT ? !plain : (F & ~(EPI_OUT_F32 | EPI_COLSUM)) != 0;
    const bool colsum = RT ? p.colsum != nullptr : bool(F & EPI_COLSUM);
    if (any) {
        if (RT || (F & EPI_BIAS)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += b4[e];
        }
        if (RT ? (p.z_out && !p.zg_f16) : bool(F & EPI_ZBF16)) {
            bf16x4 z4;
#pragma unroll
            for (int e = 0; e < 4; ++e) z4[e] = f2bf(v[e]);
            gemm_store_out((bf16x4*)zdst, z4);
        }
        if (RT ? p.act == 1 : bool(F & EPI_GELU)) {
            const bool store_g = RT ? (p.z_out && p.zg_f16) : bool(F & EPI_STOREG);
            f16x4 g4;
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                f32x2 vin; vin[0] = v[e]; vin[1] = v[e + 1];
                const f32x2 x2 = round_bf16_pair(vin);
                f32x2 y2, s2, pdf;
                gelu_y_s_pdf2(x2, y2, s2, pdf);             // one exp + one rcp per element serve gelu AND gelu'
                v[e] = y2[0]; v[e + 1] = y2[1];
                if (store_g) {
                    const f32x2 g2 = gelu_cdf_from_s2(x2, s2) + x2 * pdf;
                    g4[e] = (_Float16)g2[0]; g4[e + 1] = (_Float16)g2[1];
                }
            }
            if (store_g) gemm_store_out((f16x4*)zdst, g4);
        }
        if (RT ? have_side : (F & (EPI_ZG16 | EPI_ZGBF | EPI_RES)) != 0) {
            if (RT ? (p.zgrad && p.zg_f16) : bool(F & EPI_ZG16)) {    // the forward stored gelu'(z) (fp16 bits in the bf16-typed slots)
                const f16x4 g4 = __builtin_bit_cast(f16x4, zs);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= (float)g4[e];
            } else if (RT ? p.zgrad != nullptr : bool(F & EPI_ZGBF)) {
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    f32x2 x2; x2[0] = bf2f(zs[e]); x2[1] = bf2f(zs[e + 1]);
                    const f32x2 g2 = gelu_grad_fast2(x2);
                    v[e] *= g2[0]; v[e + 1] *= g2[1];
                }
            }
            if (RT ? p.r != nullptr : bool(F & EPI_RES)) {
                float rv[4];
                if (RT ? p.r_dtype == DW_F32 : bool(F & EPI_RES_F32)) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) rv[e] = rs[e];
                } else {
                    const unsigned u0 = __float_as_uint(rs[0]), u1 = __float_as_uint(rs[1]);
                    rv[0] = __uint_as_float(u0 << 16); rv[1] = __uint_as_float(u0 & 0xffff0000u);
                    rv[2] = __uint_as_float(u1 << 16); rv[3] = __uint_as_float(u1 & 0xffff0000u);
                }
                const bool rnd = RT ? p.round_res != 0 : bool(F & EPI_ROUND);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (rnd ? round_bf16(v[e]) : v[e]) + rv[e];
            }
        }
    }
    if (RT ? p.c_dtype == DW_F32 : bool(F & EPI_OUT_F32)) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = v[e];
        *(f32x4*)cdst = o;
        if (colsum) {
#pragma unroll
            for (int e = 0; e < 4; ++e) cs[e] += v[e];
        }
    } else {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
        gemm_store_out((bf16x4*)cdst, o);
        if (colsum) {                              // (the values as stored: what a column sum over C would read)
#pragma unroll
            for (int e = 0; e < 4; ++e) cs[e] += bf2f(o[e]);
        }
    }
}

// The lanes of a wave that own the same 4 columns in the row-major walks (lane % LPR equal) combine their partial column
// sums; one of them adds the result to DwGemm.colsum_out.
template <int LPR>
__device__ __forceinline__ void gemm_colsum_flush(const GemmP& p, float (&cs)[4], int lane, int n) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) cs[e] += __shfl_xor(cs[e], o);
    }
    if (lane < LPR) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (n + e < p.n) atomicAdd(p.colsum + n + e, cs[e]);
    }
}

struct GemmNoHook { __device__ __forceinline__ void operator()() const {} };
// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a release/acquire fence too: hipcc puts
// `s_waitcnt vmcnt(0)` in front of its s_barrier, which waits for EVERY outstanding vector-memory operation of the wave --
// the epilogue's stores and the operand DMA already in flight for the next tile.  Nothing in these kernels communicates
// through global memory inside a workgroup, so the barriers around the LDS patches / operand buffers need lgkmcnt only.
__device__ __forceinline__ void gemm_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// `hook` runs once per wave after the epilogue's leading loads are issued and before its first barrier (the software
// pipelined kernels request the next job's first operand tile there).
// SWZ: the wave's patch is 32 rows x TN floats WITHOUT padding, its 16-byte slots XOR-swizzled by (row & 15) instead
// (conflict-free for the transposing b128 writes -- 16 consecutive rows hit 16 different slots -- and for the row-major
// b128 reads -- a row's 16 slots are a permutation): 8 waves x 8 KiB = exactly one 64 KiB operand buffer, so the
// software-pipelined kernels can keep the patches in buffer 1 while the NEXT tile's first operand tile is already
// arriving in buffer 0 (gemm_wp.h).
// LAY: accumulator layout.  32: f32x16 acc[FM][FN] of v_mfma_f32_32x32x16_bf16 (lane & 31 = row of the 32-row slab, register r =
// column (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the 32-column block).  16: f32x4 acc[2 FM][2 FN] of v_mfma_f32_16x16x32_bf16
// (lane & 15 = row of the 16-row block, register r = column 4 (lane >> 4) + r of the 16-column block).  Either way a lane holds
// FN * 4 "quads" (4 consecutive columns of one row) per 32-row slab; quad q of slab i sits at row qrow(q) + lane part, column
// qcol(q) + lane part of the slab.
template <int FM, int FN, int TN, int PFDIST = 0, class Hook = GemmNoHook, bool SWZ = false, int LAY = 32, int JOFF = 0, class Acc = f32x16[FM][FN]>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, Acc& acc, char* smem, int wave, int lane,
                                              int m0_, int wm0_, int n0_, int wn0_, int ks_, Hook hook = Hook(),
                                              const float* lds_bias = nullptr, long long* tslot = nullptr) {
    // tslot (phase trace, tools/gemm_phase_trace.py; thread 0 only, or null): [5] behind the leading barrier, [6] first slab
    // transposed and its stores issued
    // lds_bias: the tile's bias slice (BN floats from column n0) staged in LDS by the caller at the start of the tile
    // (interior tiles only).  A bias read from global memory here is an ordinary load whose vmcnt wait also drains every
    // operand DMA the caller has in flight for the NEXT tile (the counter retires in order).
    static_assert(!SWZ || TN == 64, "swizzled patch: 16 slots of 16 bytes per row");
#ifndef DW_EPI_LAUNDER
#define DW_EPI_LAUNDER 1
#endif
    // Every lane-dependent value of the walks below (patch addresses under their swizzle, lane offsets of C / z / residual)
    // is a function of `lane` alone, i.e. invariant over the persistent workgroup's tile loop: hoisted in front of it they are
    // live across the K loop, where every register is taken, and get SPILLED -- and a scratch reload inside a walk is a VMEM
    // load whose vmcnt(0) wait also waits for the acknowledgement of the global store issued just before it (the fp32-residual
    // walk of the 320-row kernel did that once per row group: 100 B of scratch, 163 scratch instructions).  An opaque
    // redefinition of `lane` per tile keeps the few instructions that derive them inside the epilogue.
    if constexpr (DW_EPI_LAUNDER != 0) asm volatile("" : "+v"(lane));
    // tile and wave coordinates are the same in every lane: say so (the job index comes out of an LDS slot, which the
    // compiler must otherwise treat as a per-lane value, and every row address would be 64-bit vector arithmetic)
    const int m0 = __builtin_amdgcn_readfirstlane(m0_), wm0 = __builtin_amdgcn_readfirstlane(wm0_);
    const int n0 = __builtin_amdgcn_readfirstlane(n0_), wn0 = __builtin_amdgcn_readfirstlane(wn0_);
    const int ks = __builtin_amdgcn_readfirstlane(ks_);
    // ---- epilogue ----
    // The MFMAs were issued as (B-fragment, A-fragment), so each 32x32 accumulator holds the TRANSPOSED output tile:
    // lane&31 = output row, register r = output column (r&3) + 8*(r>>2) + 4*(lane>>5).  Every wave turns its
    // accumulators into row-major order through a private LDS patch (32 rows x TN fp32, padded stride: conflict-free
    // b128 writes), then walks it 4 rows x TN columns per instruction: every global access of the epilogue (C and Z
    // stores, residual and GELU' loads, atomics) is 4-wide per lane and contiguous along the row across 16 lanes.
    // (compile-time accumulator indices only: a runtime-indexed accumulator array would be demoted to scratch)
    //
    // Three copies of the walk, chosen per wave (wave-uniform):
    //   * interior wave tiles (every row and column in range, 4-wide accesses allowed, no atomics) run a fully
    //     unrolled branch-free walk, with or without prefetched side inputs;
    //   * everything else (ragged edges, unaligned pointers, atomic split-K) runs ONE looped copy of the general
    //     per-element code.  Keeping the general code out of the unrolled walk is what keeps the kernel's
    //     instruction footprint small: unrolled, it was >90 % of a 600 KB kernel image and the 4-wave kernel spent
    //     ~40 us per tile fetching instructions.
    constexpr int PLD = SWZ ? TN : TN + 4;         // patch row stride in floats (TN = 64 -> 68: 8 rows x 4 banks; SWZ: 64)
    constexpr int LPR = TN / 4;                    // lanes per row in the row-major walk
    constexpr int RPI = 64 / LPR;                  // rows per instruction
    constexpr int NIT = 32 / RPI;                  // row groups of a 32-row slab
    float* patch = (float*)smem + wave * (32 * PLD);
    const int hi = lane >> 5, ln = lane & 31;
    // quad accessors (see LAY above)
    const int lrow = LAY == 32 ? ln : (lane & 15);                   // lane part of a quad's row within the slab
    const int lcol = LAY == 32 ? hi * 4 : (lane >> 4) * 4;           // lane part of a quad's first column within the wave tile
    auto qrow = [](int q) constexpr -> int { return LAY == 32 ? 0 : (q / (FN * 2)) * 16; };
    auto qcol = [](int q) constexpr -> int { return LAY == 32 ? (q / 4) * 32 + (q % 4) * 8 : (q % (FN * 2)) * 16; };
    auto quad = [&](auto ic, auto qc) __attribute__((always_inline)) -> f32x4 {
        constexpr int i = decltype(ic)::value, q = decltype(qc)::value;
        f32x4 v4;
        if constexpr (LAY == 32) {
            constexpr int j = q / 4, g = q % 4;
            v4[0] = acc[i][j][g * 4 + 0]; v4[1] = acc[i][j][g * 4 + 1]; v4[2] = acc[i][j][g * 4 + 2]; v4[3] = acc[i][j][g * 4 + 3];
        } else {
            constexpr int h = q / (FN * 2), j = q % (FN * 2);
            v4 = acc[2 * i + h][j + JOFF];        // (JOFF: first 16-column block of this call within a wider wave tile)
        }
        return v4;
    };
    const bool plain = !p.bias && !p.z_out && p.act == 0 && !p.zgrad && !p.r;
    float* const cf = (float*)p.c + (long)ks * p.slice_stride;  // (fp32 outputs only; slice_stride = 0 otherwise)
    const int pr = lane / LPR, pc = (lane % LPR) * 4;
    const int n = n0 + wn0 + pc;
    const bool n_in = n < p.n;
    const bool full = p.vec && n + 3 < p.n;
    const bool interior = p.vec && !p.atomic && p.r_row_mod <= 0 && n0 + wn0 + TN <= p.n && m0 + wm0 + FM * 32 <= p.m;
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && lds_bias && interior) b4 = *(const f32x4*)(lds_bias + wn0 + pc);
    else if (p.bias && n_in) {
        if (full) b4 = *(const f32x4*)(p.bias + n);
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (n + e < p.n) b4[e] = p.bias[n + e];
        }
    }
    auto to_patch = [&](auto ic) __attribute__((always_inline)) {
        // (the FN * 4 write addresses are re-derived per slab from an opaque copy of the lane index: held across the slabs, two
        // of them were spilled and reloaded behind a vmcnt(0) that waited for the previous slab's stores)
        int l2 = lane;
        if constexpr (DW_EPI_LAUNDER != 0) asm volatile("" : "+v"(l2));
        const int lrow2 = LAY == 32 ? (l2 & 31) : (l2 & 15), lcol2 = LAY == 32 ? (l2 >> 5) * 4 : (l2 >> 4) * 4;
        static_for<0, FN * 4>([&](auto qc) __attribute__((always_inline)) {
            constexpr int q = decltype(qc)::value;
            const f32x4 v4 = quad(ic, qc);
            const int row = lrow2 + qrow(q), col = lcol2 + qcol(q);
            if constexpr (SWZ) *(f32x4*)(patch + row * PLD + (((col >> 2) ^ (row & 15)) << 2)) = v4;
            else *(f32x4*)(patch + row * PLD + col) = v4;
        });
        // (wave-private patch: the compiler's lgkmcnt wait orders these LDS writes before the reads that follow)
    };

    if (interior) {
        // Side inputs of the epilogue (z for GELU', residual r) come from HBM.  They are fetched into registers PFD
        // row groups ahead (the slot consumed at a row group is re-issued right away for the group PFD further on):
        // otherwise every row group is a serial load -> use -> store chain (the loads cannot be hoisted above the
        // previous group's stores) and the whole memory latency is exposed 32 times per tile with the matrix pipe
        // idle.  The prefetch distance is a whole slab by default (2 waves per SIMD); callers shorten it when four
        // waves share a SIMD (128 registers per lane, and the other waves cover more of the latency).
        const bool pf_z = p.zgrad != nullptr, pf_r = p.r != nullptr;
        constexpr int PFD = PFDIST > 0 ? (PFDIST < NIT ? PFDIST : NIT) : NIT;
        // Addresses: every access of the walk is (wave-uniform row-group origin) + (a per-lane offset that does not
        // change over the walk).  Written that way -- a uniform 64-bit base plus an unsigned 32-bit lane offset -- the
        // compiler keeps the bases in scalar registers and the loads / stores use the SGPR-base addressing form: one
        // lane-offset register per matrix (C, z_out, zgrad, r) instead of a 64-bit row * stride product per row group
        // (which was most of the epilogue's VALU work and register pressure).
        const int es_c = p.c_dtype == DW_F32 ? 4 : 2;
        const int es_r = p.r_dtype == DW_F32 ? 4 : 2;
        const long row_u = (long)(m0 + wm0), col_u = (long)(n0 + wn0);              // wave-uniform tile origin
        char* const c_u = (p.c_dtype == DW_F32 ? (char*)cf : (char*)p.c) + (row_u * p.ldc + col_u) * es_c;
        char* const z_u = (char*)p.z_out + (row_u * p.ldz + col_u) * 2;
        const char* const zg_u = (const char*)p.zgrad + (row_u * p.ldzg + col_u) * 2;
        const char* const r_u = (const char*)p.r + (row_u * p.ldr + col_u) * es_r;
        const unsigned l_c = (unsigned)((pr * (int)p.ldc + pc) * es_c);
        const unsigned l_z = (unsigned)((pr * (int)p.ldz + pc) * 2);
        const unsigned l_zg = (unsigned)((pr * (int)p.ldzg + pc) * 2);
        const unsigned l_r = (unsigned)((pr * (int)p.ldr + pc) * es_r);
        bf16x4 zq[PFD];
        f32x4 rq[PFD];                             // fp32 residual: 4 values; bf16 residual: raw bits in [0], [1]
        float cs[4] = {0.f, 0.f, 0.f, 0.f};        // this lane's 4 columns summed over the wave tile's rows (EPI_COLSUM)
        // One copy of the slab walk per epilogue flavour (F >= 0: compile-time; -1: run-time tests, everything else).
        // (run-time form: F = -1 with side inputs, F = -2 without -- two copies, so that the copy without them does not
        // carry the prefetch registers through the GELU arithmetic)
        auto walk = [&](auto fc) __attribute__((always_inline)) {
            constexpr int F = decltype(fc)::value;
            constexpr bool RT = F < 0;
            constexpr bool side = RT ? F == -1 : (F & (EPI_ZG16 | EPI_ZGBF | EPI_RES)) != 0;
            const bool want_z = RT ? pf_z : (F & (EPI_ZG16 | EPI_ZGBF)) != 0;
            const bool want_r = RT ? pf_r : (F & EPI_RES) != 0;
            const bool r_f32 = RT ? p.r_dtype == DW_F32 : (F & EPI_RES_F32) != 0;
            const int esc = RT ? es_c : ((F & EPI_OUT_F32) ? 4 : 2), esr = RT ? es_r : ((F & EPI_RES_F32) ? 4 : 2);
            auto side_load = [&](auto gc) __attribute__((always_inline)) {   // gc: linear row-group index = slab * NIT + group
                constexpr int gi = decltype(gc)::value;
                constexpr int slot = gi % PFD;
                constexpr long rg = (gi / NIT) * 32 + (gi % NIT) * RPI;              // first row of the row group in the wave tile
#ifndef DW_EPI_NT_SIDE
#define DW_EPI_NT_SIDE 1       // 1 (default since round 6): gelu'(z) -- 491 MB per dX-of-fc2 launch, read once -- is loaded with the non-temporal hint:
                               // 338.7 -> 337.2 ms per step against a variant build in one process (tools/ab_keys.py `lib`); 2: the residual, 3: both (3 = 1 within noise)
#endif
                if (want_z) {
                    if constexpr ((DW_EPI_NT_SIDE & 1) != 0) zq[slot] = __builtin_nontemporal_load((const bf16x4*)(zg_u + rg * p.ldzg * 2 + l_zg));
                    else zq[slot] = *(const bf16x4*)(zg_u + rg * p.ldzg * 2 + l_zg);
                }
                if (want_r) {
                    const char* src = r_u + rg * p.ldr * esr + l_r;
                    if (r_f32) {
                        if constexpr ((DW_EPI_NT_SIDE & 2) != 0) rq[slot] = __builtin_nontemporal_load((const f32x4*)src);
                        else rq[slot] = *(const f32x4*)src;
                    }
                    else {
                        const f32x2 t = *(const f32x2*)src;
                        rq[slot][0] = t[0]; rq[slot][1] = t[1];
                    }
                }
            };
            if constexpr (side) static_for<0, PFD>([&](auto gc) __attribute__((always_inline)) { side_load(gc); });
            hook();
            if constexpr (SWZ) gemm_lds_barrier(); else __syncthreads();   // every wave is done reading the operand tiles
            if (tslot) tslot[5] = (long long)__builtin_amdgcn_s_memrealtime();
            static_for<0, FM>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                if (i == 1 && tslot) tslot[6] = (long long)__builtin_amdgcn_s_memrealtime();
                to_patch(ic);
#ifndef DW_EPI_READS_FIRST
#define DW_EPI_READS_FIRST 1
#endif
#ifndef DW_EPI_RG5
#define DW_EPI_RG5 2
#endif
                // (8-wave kernels: all row groups of the slab are read back from the patch first -- the slab's 32 accumulator
                // registers have just died -- instead of one read -> lgkmcnt(0) -> arithmetic -> store round per row group with
                // the LDS latency exposed every time)
                constexpr bool RF = SWZ && DW_EPI_READS_FIRST != 0;
                constexpr int RG = FM == 5 ? DW_EPI_RG5 : NIT;     // (the 320-row kernel has no 32 free registers: a part of a slab at a time)
                f32x4 a4s[RF ? RG : 1];
                static_for<0, NIT>([&](auto itc) __attribute__((always_inline)) {
                    constexpr int it = decltype(itc)::value;
                    if constexpr (RF && it % RG == 0) {
                        static_for<0, RG>([&](auto jc) __attribute__((always_inline)) {
                            constexpr int j = decltype(jc)::value;
                            const int rlj = (it + j) * RPI + pr;
                            a4s[j] = *(const f32x4*)(patch + rlj * PLD + (((pc >> 2) ^ (rlj & 15)) << 2));
                        });
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    const int rl = it * RPI + pr;
                    constexpr long rg = i * 32 + it * RPI;
                    f32x4 a4;
                    if constexpr (RF) a4 = a4s[it % RG];
                    else a4 = *(const f32x4*)(patch + rl * PLD + (SWZ ? (((pc >> 2) ^ (rl & 15)) << 2) : pc));
                    bf16x4 zs;
                    f32x4 rs;
                    if constexpr (side) {
                        constexpr int gi = i * NIT + it;
                        zs = zq[gi % PFD]; rs = rq[gi % PFD];
                        if constexpr (gi + PFD < FM * NIT) side_load(std::integral_constant<int, gi + PFD>{});
                    }
                    float v[4] = {a4[0], a4[1], a4[2], a4[3]};
                    gemm_epi_vec4<(F < 0 ? -1 : F)>(p, v, b4, plain, side, zs, rs, c_u + rg * p.ldc * esc + l_c, z_u + rg * p.ldz * 2 + l_z, cs);
                });
            });
            if (RT ? p.colsum != nullptr : (F & EPI_COLSUM) != 0) gemm_colsum_flush<LPR>(p, cs, lane, n);
        };
        // ---- flavours WITHOUT side inputs and with bf16 output (plain, bias, bias + GELU [+ stored gelu']): the element-wise
        // work is done on the accumulators where they are (lane = output row, register = output column: the bias is one
        // value per register, GELU is per element), and only the 2-byte RESULTS go through the wave's LDS patch to become
        // row-major -- half the patch bytes of the fp32 walk above, written with ds_write_b64 instead of ds_write_b128 (the
        // LDS write path, ~79 B/clk per CU, was ~60 % of a slab's 0.74 us), read back 8 columns per lane, stored with half as
        // many (16-byte) global stores.  Same arithmetic per element as the fp32 walk: results are bit-identical.
        // Patch: 32 rows x 136 B per wave (8 B of padding per row instead of a swizzle: the transposing b64 writes of 16
        // consecutive rows land on 16 different bank pairs, and every LDS address is ONE lane register + an immediate);
        // gelu'(z) (fp16), when it is stored, goes through the same patch after the outputs.
        auto walk16 = [&](auto fc) __attribute__((always_inline)) {
            constexpr int F = decltype(fc)::value;
            static_assert((F & ~(EPI_BIAS | EPI_GELU | EPI_STOREG)) == 0 && TN == 64, "walk16: bf16 output, no side inputs");
            constexpr bool HAS_G = (F & EPI_STOREG) != 0;
            constexpr int PRS = 136;                  // patch row stride in bytes
            char* const bp = (char*)smem + wave * 8192;
            // The bias slice is read from LDS through an LDS-typed pointer.  (It used to be `lds_bias ? lds_bias + .. : p.bias + ..`:
            // a pointer that may be LDS or global is a GENERIC pointer, every read through it a flat_load, and a flat load's result
            // is waited for with vmcnt(0) AND lgkmcnt(0) -- eight times per slab the walk stopped until the previous slab's global
            // stores had been acknowledged.  Tiles without the staged slice take the fp32 walk: see ok16.)
            typedef __attribute__((address_space(3))) const f32x4 lds_cf32x4_t;
            typedef __attribute__((address_space(3))) const float lds_cfloat_t;
            lds_cfloat_t* const bsrc = (lds_cfloat_t*)lds_bias + wn0 + lcol;
            hook();
            gemm_lds_barrier();                       // every wave is done reading the operand tiles
            if (tslot) tslot[5] = (long long)__builtin_amdgcn_s_memrealtime();
            const int rr = lane >> 3, sc = lane & 7;  // row-major side: 8 rows per instruction, 8 lanes x 16 B per row
            char* const wr = bp + lrow * PRS + lcol * 2;      // transposing side: + qrow(q) * PRS + qcol(q) * 2
            const char* const rd = bp + rr * PRS + sc * 16;   // row-major side:   + it * 8 * PRS (+ 8)
            const unsigned l_c16 = (unsigned)((rr * (int)p.ldc + sc * 8) * 2);
            const unsigned l_z16 = (unsigned)((rr * (int)p.ldz + sc * 8) * 2);
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
            auto flush = [&](char* dst_u, long ld, unsigned l16, auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                u32x4 o4[4];           // (all four reads first: one LDS latency per slab instead of four read -> wait -> store rounds)
                static_for<0, 4>([&](auto itc) __attribute__((always_inline)) {
                    constexpr int it = decltype(itc)::value;
                    const u32x2 lo = *(const u32x2*)(rd + it * 8 * PRS), hi2 = *(const u32x2*)(rd + it * 8 * PRS + 8);
                    o4[it][0] = lo[0]; o4[it][1] = lo[1]; o4[it][2] = hi2[0]; o4[it][3] = hi2[1];
                });
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, 4>([&](auto itc) __attribute__((always_inline)) {
                    constexpr int it = decltype(itc)::value;
                    constexpr long rg = i * 32 + it * 8;
#ifdef DW_EPI_ABLATE      // (profiling builds: key 11 bit 32 = the accumulator-side walk without its global stores)
                    if (!(p.stage_next & 32) || o4[it][0] == 0x12345678u)
#endif
                    gemm_store_out((u32x4*)(dst_u + rg * ld * 2 + l16), o4[it]);
                });
            };
            static_for<0, FM>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                if (i == 1 && tslot) tslot[6] = (long long)__builtin_amdgcn_s_memrealtime();
                f16x4 gq[HAS_G ? FN * 4 : 1];
                // (quads in groups of four: their four bias reads are in flight together -- one LDS latency per group, not per quad)
                static_for<0, FN>([&](auto gc) __attribute__((always_inline)) {
                f32x4 bb[4];
                if constexpr ((F & EPI_BIAS) != 0)
                    static_for<0, 4>([&](auto kc) __attribute__((always_inline)) {
                        constexpr int k = decltype(kc)::value;
                        bb[k] = *(lds_cf32x4_t*)(bsrc + qcol(decltype(gc)::value * 4 + k));
                    });
                static_for<0, 4>([&](auto kc) __attribute__((always_inline)) {
                    constexpr int q = decltype(gc)::value * 4 + decltype(kc)::value;
                    std::integral_constant<int, q> qc;
                    // (explicit pairs: left to the vectorizer the four values are paired (1, 2) -- two moves, a packed and two scalar
                    // adds, three conversions, a permute and an align per quad instead of two packed adds and two packed conversions)
                    const f32x4 a4 = quad(ic, qc);
                    f32x2 p[2];
                    p[0][0] = a4[0]; p[0][1] = a4[1]; p[1][0] = a4[2]; p[1][1] = a4[3];
                    if constexpr ((F & EPI_BIAS) != 0) {
                        const f32x4 b4t = bb[decltype(kc)::value];
                        f32x2 b0, b1;
                        b0[0] = b4t[0]; b0[1] = b4t[1]; b1[0] = b4t[2]; b1[1] = b4t[3];
                        p[0] += b0; p[1] += b1;
                    }
                    if constexpr ((F & EPI_GELU) != 0) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const f32x2 x2 = round_bf16_pair(p[h]);
                            f32x2 s2, pdf;
                            gelu_y_s_pdf2(x2, p[h], s2, pdf);
                            if constexpr (HAS_G) {
                                const f32x2 g2 = gelu_cdf_from_s2(x2, s2) + x2 * pdf;
                                gq[q][2 * h] = (_Float16)g2[0]; gq[q][2 * h + 1] = (_Float16)g2[1];
                            }
                        }
                    }
                    const bf16x2 r0 = __builtin_convertvector(p[0], bf16x2), r1 = __builtin_convertvector(p[1], bf16x2);
                    u32x2 o;
                    o[0] = __builtin_bit_cast(unsigned, r0); o[1] = __builtin_bit_cast(unsigned, r1);
                    *(u32x2*)(wr + qrow(q) * PRS + qcol(q) * 2) = o;
                });
                });
                // (wave-private patch: the compiler's lgkmcnt waits order its LDS writes and reads)
                flush(c_u, p.ldc, l_c16, ic);
                if constexpr (HAS_G) {
                    static_for<0, FN * 4>([&](auto qc) __attribute__((always_inline)) {
                        constexpr int q = decltype(qc)::value;
                        *(f16x4*)(wr + qrow(q) * PRS + qcol(q) * 2) = gq[q];
                    });
                    flush(z_u, p.ldz, l_z16, ic);
                }
            });
        };
#ifndef DW_EPI16
#define DW_EPI16 1
#endif
        // 16-byte accesses of C (and z_out): pointer and leading dimension multiples of 8 elements
        const bool ok16 = DW_EPI16 && SWZ && p.c_dtype != DW_F32 && ((uintptr_t)p.c & 15) == 0 && (p.ldc & 7) == 0 && (!p.bias || lds_bias) &&
                          (!p.z_out || (((uintptr_t)p.z_out & 15) == 0 && (p.ldz & 7) == 0));
#define DW_EPI_CASE16(F)                                                                             \
    case (F):                                                                                        \
        if constexpr (SWZ) { if (ok16) { walk16(std::integral_constant<int, (F)>{}); return; } }      \
        walk(std::integral_constant<int, (F)>{}); return
#define DW_EPI_CASE(F) case (F): walk(std::integral_constant<int, (F)>{}); return
        if constexpr (FM == 4 || FM == 2) {
            // 256-row kernels (8 waves: FM = 4; 16 waves and the 128-tile variant: FM = 2): one compact walk per flavour the step uses; any other flavour (the conv stem's GEMMs: two
            // launches per step) takes the looped general walk below -- there is NO unrolled run-time copy in the image.
            switch (gemm_epi_flavour(p)) {
                DW_EPI_CASE16(0);                                                              // dX GEMMs, LM head
                DW_EPI_CASE16(EPI_BIAS);                                                       // QKV / Q / KV projections
                DW_EPI_CASE16(EPI_BIAS | EPI_GELU);                                            // teacher fc1
                DW_EPI_CASE(EPI_BIAS | EPI_GELU | EPI_STOREG);                               // student fc1 (keeps gelu'(z); two outputs: the fp32 walk is 1 % faster; 0.5 % per step with non-temporal stores)
                DW_EPI_CASE(EPI_BIAS | EPI_RES | EPI_RES_F32 | EPI_ROUND | EPI_OUT_F32);      // student out-proj / fc2
                DW_EPI_CASE(EPI_BIAS | EPI_RES | EPI_ROUND);                                  // teacher out-proj / fc2
                DW_EPI_CASE(EPI_ZG16);                                                        // dX of fc2 (x gelu'(z))
                DW_EPI_CASE(EPI_ZG16 | EPI_COLSUM);                                           // ... + fc1.bias gradient
                DW_EPI_CASE(EPI_RES | EPI_RES_F32 | EPI_ROUND | EPI_OUT_F32);                 // d(encoder output) accumulation
                DW_EPI_CASE(EPI_OUT_F32);                                                     // fp32 partial slabs of the dW GEMMs
                default: break;
            }
        } else {
            // (the 320-row tile -- FM = 5, 160 accumulator registers -- and the 16-wave kernels keep the run-time walk for
            // the flavours with side inputs: with one walk per flavour the 320-row kernel's register allocation falls
            // apart, 763 spills against 12)
            if constexpr (FM == 5) {
                switch (gemm_epi_flavour(p)) {
                    DW_EPI_CASE16(0);
                    DW_EPI_CASE16(EPI_BIAS);
                    DW_EPI_CASE16(EPI_BIAS | EPI_GELU);
                    DW_EPI_CASE(EPI_BIAS | EPI_GELU | EPI_STOREG);
                    DW_EPI_CASE(EPI_BIAS | EPI_RES | EPI_RES_F32 | EPI_ROUND | EPI_OUT_F32);
                    DW_EPI_CASE(EPI_BIAS | EPI_RES | EPI_ROUND);
                    default: break;
                }
            }
            if (pf_z || pf_r) walk(std::integral_constant<int, -1>{});
            else walk(std::integral_constant<int, -2>{});
            return;
        }
#undef DW_EPI_CASE
#undef DW_EPI_CASE16
    }

    // ---- general walk (ragged tile edges, unaligned pointers, atomic accumulation): looped, loads at use ----
    hook();
    if constexpr (SWZ) gemm_lds_barrier(); else __syncthreads();
    float gcs[4] = {0.f, 0.f, 0.f, 0.f};
    static_for<0, FM>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        to_patch(ic);
#pragma unroll 1
        for (int it = 0; it < NIT; ++it) {
            const int rl = it * RPI + pr;
            const int m = m0 + wm0 + i * 32 + rl;
            const f32x4 a4 = *(const f32x4*)(patch + rl * PLD + (SWZ ? (((pc >> 2) ^ (rl & 15)) << 2) : pc));
            if (m >= p.m || !n_in) continue;
            float v[4] = {a4[0], a4[1], a4[2], a4[3]};
            if (p.atomic) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < p.n) atomicAdd(cf + (long)m * p.ldc + n + e, v[e]);
                continue;
            }
            const int rr = p.r_row_mod > 0 ? (m % p.r_row_mod) : m;
            if (full) {
                bf16x4 zs = {};
                f32x4 rs = {0.f, 0.f, 0.f, 0.f};
                if (p.zgrad) zs = *(const bf16x4*)(p.zgrad + (long)m * p.ldzg + n);
                if (p.r) {
                    if (p.r_dtype == DW_F32) rs = *(const f32x4*)((const float*)p.r + (long)rr * p.ldr + n);
                    else {
                        const f32x2 t = *(const f32x2*)((const bf16*)p.r + (long)rr * p.ldr + n);
                        rs[0] = t[0]; rs[1] = t[1];
                    }
                }
                gemm_epi_vec4(p, v, b4, plain, true, zs, rs,
                              p.c_dtype == DW_F32 ? (char*)(cf + (long)m * p.ldc + n) : (char*)((bf16*)p.c + (long)m * p.ldc + n),
                              (char*)(p.z_out + (long)m * p.ldz + n), gcs);
            } else {
                // ragged / unaligned columns: scalar path
#pragma unroll 1
                for (int e = 0; e < 4; ++e) {
                    const int nn = n + e;
                    if (nn >= p.n) continue;
                    float x = v[e];
                    if (!plain) {
                        x += b4[e];
                        if (p.z_out && !p.zg_f16) p.z_out[(long)m * p.ldz + nn] = f2bf(x);
                        if (p.act == 1) {
                            const float xr = round_bf16(x);
                            if (p.z_out && p.zg_f16) ((_Float16*)p.z_out)[(long)m * p.ldz + nn] = (_Float16)gelu_grad_fast(xr);
                            x = gelu_fast(xr);
                        }
                        if (p.zgrad && p.zg_f16) x *= (float)((const _Float16*)p.zgrad)[(long)m * p.ldzg + nn];
                        else if (p.zgrad) x *= gelu_grad_fast(bf2f(p.zgrad[(long)m * p.ldzg + nn]));
                        if (p.r) {
                            const float rv = p.r_dtype == DW_F32 ? ((const float*)p.r)[(long)rr * p.ldr + nn]
                                                                 : bf2f(((const bf16*)p.r)[(long)rr * p.ldr + nn]);
                            x = (p.round_res ? round_bf16(x) : x) + rv;
                        }
                    }
                    if (p.c_dtype == DW_F32) cf[(long)m * p.ldc + nn] = x;
                    else { ((bf16*)p.c)[(long)m * p.ldc + nn] = f2bf(x); x = round_bf16(x); }
                    if (p.colsum) {                  // (compile-time indices: a run-time index would demote gcs to scratch)
                        if (e == 0) gcs[0] += x; else if (e == 1) gcs[1] += x; else if (e == 2) gcs[2] += x; else gcs[3] += x;
                    }
                }
            }
        }
    });
    if (p.colsum) gemm_colsum_flush<LPR>(p, gcs, lane, n);
}
