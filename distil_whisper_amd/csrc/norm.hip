// LayerNorm forward / backward for gfx950 (HBM-bound; one 64-lane wave per row, 16-byte loads, fp32 statistics).
// Replaces nn.LayerNorm(eps=1e-5) of the reference model (TF:modeling_whisper.py:371,377,434,443,446,573,682).
// Under the reference's bf16 autocast LayerNorm runs in fp32 on the (fp32 student / bf16 teacher) residual stream and
// its output is cast to bf16 by the next Linear; here the bf16 cast is fused into the store.
#include "common.h"
#include "../../include/dwamd.h"

#define LN_MAXV 8  // up to 8 float4 per lane -> cols <= 2048
#ifndef DW_NT_LN_IN
#define DW_NT_LN_IN 1      // 1 (default): the forward kernels' row loads are non-temporal; 2: + the backward kernels' x and dy loads; 0: off.
                           // Same-process A/B of variant builds on the full step (tools/ab_keys.py `lib`): 344.8 -> 343.9 and 343.1 -> 342.5 ms with 1
                           // (the rows are read once here; kept out of the caches they leave more room for the GEMM operands); 2 and the same
                           // hint in AdamW (-DDW_NT_ADAMW) add nothing measurable.
#endif

template <bool XBF>
__device__ __forceinline__ f32x4 ld4b(const void* x, long off) {       // backward kernels: x and dy are read once
    if constexpr (DW_NT_LN_IN >= 2) {
        if (XBF) {
            const bf16x4 t = __builtin_nontemporal_load((const bf16x4*)((const bf16*)x + off));
            f32x4 r; r[0] = bf2f(t[0]); r[1] = bf2f(t[1]); r[2] = bf2f(t[2]); r[3] = bf2f(t[3]);
            return r;
        } else return __builtin_nontemporal_load((const f32x4*)((const float*)x + off));
    } else {
        if (XBF) {
            const bf16x4 t = *(const bf16x4*)((const bf16*)x + off);
            f32x4 r; r[0] = bf2f(t[0]); r[1] = bf2f(t[1]); r[2] = bf2f(t[2]); r[3] = bf2f(t[3]);
            return r;
        } else return *(const f32x4*)((const float*)x + off);
    }
}
template <bool XBF>
__device__ __forceinline__ f32x4 ld4(const void* x, long off) {
    if (XBF) {
        const bf16x4 t = *(const bf16x4*)((const bf16*)x + off);
        f32x4 r; r[0] = bf2f(t[0]); r[1] = bf2f(t[1]); r[2] = bf2f(t[2]); r[3] = bf2f(t[3]);
        return r;
    } else {
        return *(const f32x4*)((const float*)x + off);
    }
}

// Output stores.  -DDW_NT_LN=n marks those of level <= n non-temporal (1: y, 2: + the fp32 residual gradient, 3: + its bf16 copy).
#ifndef DW_NT_LN
#define DW_NT_LN 0
#endif
// Input loads of the forward kernels.  -DDW_NT_LN_IN=1 marks the row loads non-temporal (the fp32 residual stream / the teacher's bf16
// stream is read ONCE by this kernel; the student's is read again only in the backward, hundreds of MB of other traffic later).
template <class T>
__device__ __forceinline__ T ln_load_in(const T* src) {
    if constexpr (DW_NT_LN_IN != 0) return __builtin_nontemporal_load(src);
    else return *src;
}
template <int LEVEL, class T>
__device__ __forceinline__ void ln_store(T* dst, const T& v) {
    if constexpr (LEVEL <= DW_NT_LN) __builtin_nontemporal_store(v, dst);
    else *dst = v;
}

template <bool XBF, int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const void* x, const float* gamma, const float* beta, bf16* y,
                                                     float* mean, float* rstd, int rows, int cols, float eps, long ldx, long ldy) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvec = cols >> 2;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int idx = lane + 64 * j;
        if (idx < nvec) {
            v[j] = ld4<XBF>(x, (long)row * ldx + idx * 4);
            s += v[j][0] + v[j][1] + v[j][2] + v[j][3];
        }
    }
    s = wave_sum(s);
    const float mu = s / cols;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int idx = lane + 64 * j;
        if (idx < nvec) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[j][e] - mu; q += d * d; }
        }
    }
    q = wave_sum(q);
    const float rs = rsqrtf(q / cols + eps);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int idx = lane + 64 * j;
        if (idx < nvec) {
            const f32x4 g = *(const f32x4*)(gamma + idx * 4);
            const f32x4 bt = *(const f32x4*)(beta + idx * 4);
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f2bf((v[j][e] - mu) * rs * g[e] + bt[e]);
            ln_store<1>((bf16x4*)(y + (long)row * ldy + idx * 4), o);
        }
    }
    if (lane == 0 && mean) { mean[row] = mu; rstd[row] = rs; }
}

// Persistent form of the fp32-input forward (the student's residual stream, 6 B per element): a resident grid walks the rows
// and every wave requests its NEXT row before it reduces the current one -- one-row waves (5 KiB each) spend a third of their
// life being launched and retired, and have nothing in flight while they reduce (dw_debug_set key 21 bit 0).
template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_persist_kernel(const float* x, const float* gamma, const float* beta, bf16* y,
                                                             float* mean, float* rstd, int rows, int cols, float eps, long ldx, long ldy) {
    const int lane = threadIdx.x & 63;
    const int nvec = cols >> 2;
    const int stride = gridDim.x * 4;
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    f32x4 v[NV], nx[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int idx = lane + 64 * j;
        if (idx < nvec) v[j] = ln_load_in((const f32x4*)(x + (long)row * ldx + idx * 4));
    }
    for (; row < rows; row += stride) {
        const int nrow = row + stride;
        if (nrow < rows) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int idx = lane + 64 * j;
                if (idx < nvec) nx[j] = ln_load_in((const f32x4*)(x + (long)nrow * ldx + idx * 4));
            }
        }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j)
            if (lane + 64 * j < nvec) s += v[j][0] + v[j][1] + v[j][2] + v[j][3];
        s = wave_sum(s);
        const float mu = s / cols;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j)
            if (lane + 64 * j < nvec) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[j][e] - mu; q += d * d; }
            }
        q = wave_sum(q);
        const float rs = rsqrtf(q / cols + eps);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int idx = lane + 64 * j;
            if (idx < nvec) {
                const f32x4 g = *(const f32x4*)(gamma + idx * 4);
                const f32x4 bt = *(const f32x4*)(beta + idx * 4);
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = f2bf((v[j][e] - mu) * rs * g[e] + bt[e]);
                ln_store<1>((bf16x4*)(y + (long)row * ldy + idx * 4), o);
            }
        }
        if (lane == 0 && mean) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] = nx[j];
    }
}

// bf16 input (the teacher's residual stream) with 16-byte accesses: 8 elements per lane per vector.  With 8-byte loads the
// row of a bf16 stream is half as many bytes per request as an fp32 row and the kernel ran at 3.0 TB/s against 4.5 for
// fp32 input (tools/stream_kernels_bench.py).
template <int NV8>
__global__ __launch_bounds__(256) void ln_fwd_bf16x8_kernel(const bf16* x, const float* gamma, const float* beta, bf16* y,
                                                            float* mean, float* rstd, int rows, int cols, float eps, long ldx, long ldy) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvec = cols >> 3;
    float v[NV8][8];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV8; ++j) {
        const int idx = lane + 64 * j;
        if (idx < nvec) {
            const bf16x8 t = ln_load_in((const bf16x8*)(x + (long)row * ldx + idx * 8));
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[j][e] = bf2f(t[e]); s += v[j][e]; }
        }
    }
    s = wave_sum(s);
    const float mu = s / cols;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV8; ++j) {
        const int idx = lane + 64 * j;
        if (idx < nvec) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[j][e] - mu; q += d * d; }
        }
    }
    q = wave_sum(q);
    const float rs = rsqrtf(q / cols + eps);
#pragma unroll
    for (int j = 0; j < NV8; ++j) {
        const int idx = lane + 64 * j;
        if (idx < nvec) {
            const f32x4 g0 = *(const f32x4*)(gamma + idx * 8), g1 = *(const f32x4*)(gamma + idx * 8 + 4);
            const f32x4 b0 = *(const f32x4*)(beta + idx * 8), b1 = *(const f32x4*)(beta + idx * 8 + 4);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = f2bf((v[j][e] - mu) * rs * g0[e] + b0[e]);
                o[e + 4] = f2bf((v[j][e + 4] - mu) * rs * g1[e] + b1[e]);
            }
            ln_store<1>((bf16x8*)(y + (long)row * ldy + idx * 8), o);
        }
    }
    if (lane == 0 && mean) { mean[row] = mu; rstd[row] = rs; }
}

template <bool XBF, int NV, int NW>
__global__ __launch_bounds__(NW * 64) void ln_bwd_kernel(const bf16* dy, const void* x, const float* mean,
                                                     const float* rstd, const float* gamma, float* dres,
                                                     int accumulate, float* dgamma, float* dbeta, bf16* dres_lowp,
                                                     float* dres_colsum, int rows, int cols, long lddy, long ldx, long lddres,
                                                         long ldlowp) {
    __shared__ float red[3 * NW * 512];  // [dgamma|dbeta|colsum][wave][512-column window]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nvec = cols >> 2;
    f32x4 gm[NV], ag[NV], ab[NV], ac[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int idx = lane + 64 * j;
#pragma unroll
        for (int e = 0; e < 4; ++e) { ag[j][e] = 0.f; ab[j][e] = 0.f; gm[j][e] = 0.f; ac[j][e] = 0.f; }
        if (idx < nvec) gm[j] = *(const f32x4*)(gamma + idx * 4);
    }
    for (int row = blockIdx.x * NW + wave; row < rows; row += gridDim.x * NW) {
        const float mu = mean[row], rs = rstd[row];
        f32x4 xh[NV], g[NV];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int idx = lane + 64 * j;
            if (idx < nvec) {
                const f32x4 xv = ld4b<XBF>(x, (long)row * ldx + idx * 4);
                const f32x4 dv = ld4b<true>(dy, (long)row * lddy + idx * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[j][e] = (xv[e] - mu) * rs;
                    g[j][e] = dv[e] * gm[j][e];
                    c1 += g[j][e];
                    c2 += g[j][e] * xh[j][e];
                    ag[j][e] += dv[e] * xh[j][e];
                    ab[j][e] += dv[e];
                }
            }
        }
        c1 = wave_sum(c1) / cols;
        c2 = wave_sum(c2) / cols;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int idx = lane + 64 * j;
            if (idx < nvec) {
                float* dst = dres + (long)row * lddres + idx * 4;
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rs * (g[j][e] - c1 - xh[j][e] * c2);
                if (accumulate) {
                    const f32x4 old = *(const f32x4*)dst;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += old[e];
                }
                ln_store<2>((f32x4*)dst, o);
                if (dres_lowp) {
                    // the low-precision copy the next branch's GEMMs consume (autocast: the gradient of a bf16 Linear
                    // output is bf16) and its column sums = the bias gradient of that Linear
                    bf16x4 lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { lo[e] = f2bf(o[e]); ac[j][e] += bf2f(lo[e]); }
                    ln_store<3>((bf16x4*)(dres_lowp + (long)row * ldlowp + idx * 4), lo);
                }
            }
        }
    }
    // block reduction of the per-wave dgamma / dbeta partials, then one atomic per column per block
    float* r0 = red;
    float* r1 = red + NW * 512;
    float* r2 = red + 2 * NW * 512;
    // windows of 512 columns: red[k][wave][col - base]
    for (int base = 0; base < cols; base += 512) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int idx = lane + 64 * j;
            const int col = idx * 4;
            if (idx < nvec && col >= base && col < base + 512) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    r0[wave * 512 + (col - base) + e] = ag[j][e];
                    r1[wave * 512 + (col - base) + e] = ab[j][e];
                    r2[wave * 512 + (col - base) + e] = ac[j][e];
                }
            }
        }
        __syncthreads();
        for (int cidx = threadIdx.x; cidx < 512 && base + cidx < cols; cidx += NW * 64) {
            float sg = 0.f, sb = 0.f, sc = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) { sg += r0[w * 512 + cidx]; sb += r1[w * 512 + cidx]; sc += r2[w * 512 + cidx]; }
            atomicAdd(dgamma + base + cidx, sg);
            atomicAdd(dbeta + base + cidx, sb);
            if (dres_colsum) atomicAdd(dres_colsum + base + cidx, sc);
        }
        __syncthreads();
    }
}

// The same backward with the NEXT row's x / dy and the CURRENT row's residual gradient requested before the row reductions
// (two dependent memory round trips per row otherwise: operands -> reductions -> old residual gradient -> stores).  Needs the
// registers of two waves per SIMD (NW = 8, one workgroup per CU); dw_debug_set key 21 bit 1.
template <bool XBF, int NV, int NW>
__global__ __launch_bounds__(NW * 64) void ln_bwd_pf_kernel(const bf16* dy, const void* x, const float* mean,
                                                         const float* rstd, const float* gamma, float* dres,
                                                         int accumulate, float* dgamma, float* dbeta, bf16* dres_lowp,
                                                         float* dres_colsum, int rows, int cols, long lddy, long ldx, long lddres,
                                                         long ldlowp) {
    __shared__ float red[3 * NW * 512];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nvec = cols >> 2;
    f32x4 gm[NV], ag[NV], ab[NV], ac[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int idx = lane + 64 * j;
#pragma unroll
        for (int e = 0; e < 4; ++e) { ag[j][e] = 0.f; ab[j][e] = 0.f; gm[j][e] = 0.f; ac[j][e] = 0.f; }
        if (idx < nvec) gm[j] = *(const f32x4*)(gamma + idx * 4);
    }
    const int stride = gridDim.x * NW;
    int row = blockIdx.x * NW + wave;
    f32x4 xv[NV], dv[NV];
    float mu = 0.f, rs = 0.f;
    if (row < rows) {
        mu = mean[row]; rs = rstd[row];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int idx = lane + 64 * j;
            if (idx < nvec) { xv[j] = ld4b<XBF>(x, (long)row * ldx + idx * 4); dv[j] = ld4b<true>(dy, (long)row * lddy + idx * 4); }
        }
    }
    for (; row < rows; row += stride) {
        const int nrow = row + stride;
        // requests first: the old residual gradient of this row, then the next row's operands
        f32x4 old[NV], nxv[NV], ndv[NV];
        float nmu = 0.f, nrs = 0.f;
        if (accumulate) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int idx = lane + 64 * j;
                if (idx < nvec) old[j] = *(const f32x4*)(dres + (long)row * lddres + idx * 4);
            }
        }
        if (nrow < rows) {
            nmu = mean[nrow]; nrs = rstd[nrow];
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int idx = lane + 64 * j;
                if (idx < nvec) { nxv[j] = ld4b<XBF>(x, (long)nrow * ldx + idx * 4); ndv[j] = ld4b<true>(dy, (long)nrow * lddy + idx * 4); }
            }
        }
        f32x4 xh[NV], g[NV];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int idx = lane + 64 * j;
            if (idx < nvec) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[j][e] = (xv[j][e] - mu) * rs;
                    g[j][e] = dv[j][e] * gm[j][e];
                    c1 += g[j][e];
                    c2 += g[j][e] * xh[j][e];
                    ag[j][e] += dv[j][e] * xh[j][e];
                    ab[j][e] += dv[j][e];
                }
            }
        }
        c1 = wave_sum(c1) / cols;
        c2 = wave_sum(c2) / cols;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int idx = lane + 64 * j;
            if (idx < nvec) {
                float* dst = dres + (long)row * lddres + idx * 4;
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rs * (g[j][e] - c1 - xh[j][e] * c2);
                if (accumulate) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += old[j][e];
                }
                ln_store<2>((f32x4*)dst, o);
                if (dres_lowp) {
                    bf16x4 lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { lo[e] = f2bf(o[e]); ac[j][e] += bf2f(lo[e]); }
                    ln_store<3>((bf16x4*)(dres_lowp + (long)row * ldlowp + idx * 4), lo);
                }
            }
        }
        mu = nmu; rs = nrs;
#pragma unroll
        for (int j = 0; j < NV; ++j) { xv[j] = nxv[j]; dv[j] = ndv[j]; }
    }
    float* r0 = red;
    float* r1 = red + NW * 512;
    float* r2 = red + 2 * NW * 512;
    for (int base = 0; base < cols; base += 512) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int idx = lane + 64 * j;
            const int col = idx * 4;
            if (idx < nvec && col >= base && col < base + 512) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    r0[wave * 512 + (col - base) + e] = ag[j][e];
                    r1[wave * 512 + (col - base) + e] = ab[j][e];
                    r2[wave * 512 + (col - base) + e] = ac[j][e];
                }
            }
        }
        __syncthreads();
        for (int cidx = threadIdx.x; cidx < 512 && base + cidx < cols; cidx += NW * 64) {
            float sg = 0.f, sb = 0.f, sc = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) { sg += r0[w * 512 + cidx]; sb += r1[w * 512 + cidx]; sc += r2[w * 512 + cidx]; }
            atomicAdd(dgamma + base + cidx, sg);
            atomicAdd(dbeta + base + cidx, sb);
            if (dres_colsum) atomicAdd(dres_colsum + base + cidx, sc);
        }
        __syncthreads();
    }
}

// dw_debug_set key 21: bit 0 persistent forward with next-row prefetch (fp32 input, <= 1280 columns: 84.9 -> 72.0 us at
// [48000 x 1280], 4.3 -> 5.1 TB/s; bits 8-11: workgroups per CU, default 5), bit 1 backward with prefetch (1025-1280 columns:
// 225 -> 179 us, 4.4 -> 5.5 TB/s of its 16 B per element).  tools/ln_ab.py.  The same treatment of the bf16-input forward
// (the teacher's stream, cache resident) measured neutral (52.7 vs 51.7 us) and is not built.
int g_ln_variant = 3;

extern "C" int dw_layernorm_fwd_ld(const void* x, int x_dtype, const float* gamma, const float* beta, void* y,
                                   float* mean, float* rstd, int rows, int cols, float eps, int64_t ldx_, int64_t ldy_, void* stream);
extern "C" int dw_layernorm_fwd(const void* x, int x_dtype, const float* gamma, const float* beta, void* y,
                                float* mean, float* rstd, int rows, int cols, float eps, void* stream) {
    return dw_layernorm_fwd_ld(x, x_dtype, gamma, beta, y, mean, rstd, rows, cols, eps, cols, cols, stream);
}
// The same with row pitches (elements) of x and y: activation buffers whose rows are padded against L2-channel camping
// (engine.row_pad; tools/gemm_stride_probe.py).
extern "C" int dw_layernorm_fwd_ld(const void* x, int x_dtype, const float* gamma, const float* beta, void* y,
                                   float* mean, float* rstd, int rows, int cols, float eps, int64_t ldx_, int64_t ldy_, void* stream) {
    DW_CLEAR_ERR();
    if (!x || !gamma || !beta || !y || rows <= 0 || cols <= 0 || (cols & 3) || cols > LN_MAXV * 256) return DW_EINVAL;
    if (ldx_ < cols || ldy_ < cols || (ldx_ & 3) || (ldy_ & 3)) return DW_EINVAL;
    const long ldx = ldx_, ldy = ldy_;
    if ((mean == nullptr) != (rstd == nullptr)) return DW_EINVAL;
    dim3 grid((rows + 3) / 4), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == DW_BF16 && (cols & 7) == 0 && ((ldx | ldy) & 7) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 &&
        ((uintptr_t)gamma & 15) == 0 && ((uintptr_t)beta & 15) == 0) {
        const int nv8 = ((cols >> 3) + 63) / 64;
        if (nv8 <= 1) hipLaunchKernelGGL((ln_fwd_bf16x8_kernel<1>), grid, block, 0, s, (const bf16*)x, gamma, beta, (bf16*)y, mean, rstd, rows, cols, eps, ldx, ldy);
        else if (nv8 <= 2) hipLaunchKernelGGL((ln_fwd_bf16x8_kernel<2>), grid, block, 0, s, (const bf16*)x, gamma, beta, (bf16*)y, mean, rstd, rows, cols, eps, ldx, ldy);
        else if (nv8 <= 3) hipLaunchKernelGGL((ln_fwd_bf16x8_kernel<3>), grid, block, 0, s, (const bf16*)x, gamma, beta, (bf16*)y, mean, rstd, rows, cols, eps, ldx, ldy);
        else hipLaunchKernelGGL((ln_fwd_bf16x8_kernel<4>), grid, block, 0, s, (const bf16*)x, gamma, beta, (bf16*)y, mean, rstd, rows, cols, eps, ldx, ldy);
        DW_CHECK_LAUNCH();
        return DW_OK;
    }
    const int nv = ((cols >> 2) + 63) / 64;
    if ((g_ln_variant & 1) && x_dtype != DW_BF16 && nv <= 5 && rows >= 8192) {
        // resident grid: 5 workgroups of 4 waves per CU (v + next row + parameters: ~100 registers)
        dim3 pg(256 * ((g_ln_variant >> 8) & 15 ? (g_ln_variant >> 8) & 15 : 5));
        hipLaunchKernelGGL((ln_fwd_persist_kernel<5>), pg, block, 0, s, (const float*)x, gamma, beta, (bf16*)y, mean, rstd, rows, cols, eps, ldx, ldy);
        DW_CHECK_LAUNCH();
        return DW_OK;
    }
#define LN_FWD(NVV)                                                                                                   \
    do {                                                                                                              \
        if (x_dtype == DW_BF16)                                                                                       \
            hipLaunchKernelGGL((ln_fwd_kernel<true, NVV>), grid, block, 0, s, x, gamma, beta, (bf16*)y, mean, rstd,   \
                               rows, cols, eps, ldx, ldy);                                                                      \
        else                                                                                                          \
            hipLaunchKernelGGL((ln_fwd_kernel<false, NVV>), grid, block, 0, s, x, gamma, beta, (bf16*)y, mean, rstd,  \
                               rows, cols, eps, ldx, ldy);                                                                      \
    } while (0)
    if (nv <= 2) LN_FWD(2); else if (nv <= 3) LN_FWD(3); else if (nv <= 5) LN_FWD(5); else LN_FWD(8);
#undef LN_FWD
    DW_CHECK_LAUNCH();
    return DW_OK;
}

extern "C" int dw_layernorm_bwd_ld(const void* dy, const void* x, int x_dtype, const float* mean, const float* rstd,
                                   const float* gamma, float* dres, int accumulate, float* dgamma, float* dbeta,
                                   void* dres_lowp, float* dres_colsum, int rows, int cols, int64_t lddy_, int64_t ldx_,
                                   int64_t lddres_, int64_t ldlowp_, void* stream);
extern "C" int dw_layernorm_bwd(const void* dy, const void* x, int x_dtype, const float* mean, const float* rstd,
                                const float* gamma, float* dres, int accumulate, float* dgamma, float* dbeta,
                                void* dres_lowp, float* dres_colsum, int rows, int cols, void* stream) {
    return dw_layernorm_bwd_ld(dy, x, x_dtype, mean, rstd, gamma, dres, accumulate, dgamma, dbeta, dres_lowp, dres_colsum, rows,
                               cols, cols, cols, cols, cols, stream);
}
// ... with row pitches (elements) of dy, x, the fp32 residual gradient and its bf16 copy
extern "C" int dw_layernorm_bwd_ld(const void* dy, const void* x, int x_dtype, const float* mean, const float* rstd,
                                   const float* gamma, float* dres, int accumulate, float* dgamma, float* dbeta,
                                   void* dres_lowp, float* dres_colsum, int rows, int cols, int64_t lddy_, int64_t ldx_,
                                   int64_t lddres_, int64_t ldlowp_, void* stream) {
    DW_CLEAR_ERR();
    if (!dy || !x || !mean || !rstd || !gamma || !dres || !dgamma || !dbeta) return DW_EINVAL;
    if (rows <= 0 || cols <= 0 || (cols & 3) || cols > LN_MAXV * 256) return DW_EINVAL;
    if (lddy_ < cols || ldx_ < cols || lddres_ < cols || ldlowp_ < cols || ((lddy_ | ldx_ | lddres_ | ldlowp_) & 3)) return DW_EINVAL;
    const long lddy = lddy_, ldx = ldx_, lddres = lddres_, ldlowp = ldlowp_;
    hipStream_t s = (hipStream_t)stream;
    const int nv = ((cols >> 2) + 63) / 64;
    // grid = exactly the workgroups that are resident at once: a grid-stride loop over rows on a grid of 1.33 rounds left
    // a third of the chip idle in the tail.  Registers allow 16 / 12 / 4 waves per CU for <=3 / 5 / 8 vectors per lane; at 5
    // (cols = 1280) the twelve waves are ONE workgroup, so that a CU issues a third of the dgamma / dbeta / column-sum
    // atomics of four-wave workgroups: the atomics are rate-limited (2.9 M of them were 21 us of a 248 us launch; measured
    // with them switched off), 249 -> 238 us.  Requesting the residual gradient with the other operands instead of after the
    // reductions costs a wave per SIMD in registers and is slower (279 us), and so are LDS accumulators (ds_add_f32: 948 us)
    // and one lane per column segment with the row statistics exchanged through LDS (295 us).
#define LN_BWD(NVV, NWW)                                                                                              \
    do {                                                                                                              \
        int nb = (rows + NWW - 1) / NWW;                                                                              \
        const int resident = 256 * ((NVV <= 3 ? 16 : NVV <= 5 ? 12 : 4) / NWW);                                       \
        if (nb > resident) nb = resident;                                                                             \
        if (x_dtype == DW_BF16)                                                                                       \
            hipLaunchKernelGGL((ln_bwd_kernel<true, NVV, NWW>), dim3(nb), dim3(NWW * 64), 0, s, (const bf16*)dy, x,   \
                               mean, rstd, gamma, dres, accumulate, dgamma, dbeta, (bf16*)dres_lowp, dres_colsum,     \
                               rows, cols, lddy, ldx, lddres, ldlowp);                                                                           \
        else                                                                                                          \
            hipLaunchKernelGGL((ln_bwd_kernel<false, NVV, NWW>), dim3(nb), dim3(NWW * 64), 0, s, (const bf16*)dy, x,  \
                               mean, rstd, gamma, dres, accumulate, dgamma, dbeta, (bf16*)dres_lowp, dres_colsum,     \
                               rows, cols, lddy, ldx, lddres, ldlowp);                                                                           \
    } while (0)
    if ((g_ln_variant & 2) && nv == 5) {
        int nb = (rows + 7) / 8;
        if (nb > 256) nb = 256;
        if (x_dtype == DW_BF16)
            hipLaunchKernelGGL((ln_bwd_pf_kernel<true, 5, 8>), dim3(nb), dim3(512), 0, s, (const bf16*)dy, x, mean, rstd, gamma, dres,
                               accumulate, dgamma, dbeta, (bf16*)dres_lowp, dres_colsum, rows, cols, lddy, ldx, lddres, ldlowp);
        else
            hipLaunchKernelGGL((ln_bwd_pf_kernel<false, 5, 8>), dim3(nb), dim3(512), 0, s, (const bf16*)dy, x, mean, rstd, gamma, dres,
                               accumulate, dgamma, dbeta, (bf16*)dres_lowp, dres_colsum, rows, cols, lddy, ldx, lddres, ldlowp);
        DW_CHECK_LAUNCH();
        return DW_OK;
    }
    if (nv <= 2) LN_BWD(2, 4); else if (nv <= 3) LN_BWD(3, 4); else if (nv <= 5) LN_BWD(5, 12); else LN_BWD(8, 4);
#undef LN_BWD
    DW_CHECK_LAUNCH();
    return DW_OK;
}
