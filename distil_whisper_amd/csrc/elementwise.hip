// Streaming (HBM-bound) helper kernels of the distillation step for gfx950: embeddings, the conv front end's
// im2col / col2im, conv weight repacking, casts, bias-gradient column sums.  All use 8/16-byte per-lane accesses.
// Reference sites: TF:modeling_whisper.py:566-567, 618-625 (conv1/conv2 + GELU), 675-676, 736-762 (embeddings).
#include "common.h"
#include "../../include/dwamd.h"

// ---------------------------------------------------------------------------------------------------------------
// embeddings
// ---------------------------------------------------------------------------------------------------------------
template <bool TBF, bool OBF>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* ids, const void* tok, const void* pos, void* out,
                                                        int T, int D, long nvec) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nvec) return;
    const int dv = D >> 2;
    const long row = i / dv;
    const int c = (int)(i - row * dv) * 4;
    const int t = (int)(row % T);
    const long id = ids[row];
    f32x4 a, b;
    if (TBF) {
        const bf16x4 x = *(const bf16x4*)((const bf16*)tok + id * D + c);
        const bf16x4 y = *(const bf16x4*)((const bf16*)pos + (long)t * D + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = bf2f(x[e]); b[e] = bf2f(y[e]); }
    } else {
        a = *(const f32x4*)((const float*)tok + id * D + c);
        b = *(const f32x4*)((const float*)pos + (long)t * D + c);
    }
    if (OBF) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(a[e] + b[e]);
        *(bf16x4*)((bf16*)out + row * D + c) = o;
    } else {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = a[e] + b[e];
        *(f32x4*)((float*)out + row * D + c) = o;
    }
}

__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* dx, const int64_t* ids, float* dtok, float* dpos,
                                                        int B, int T, int D) {
    // grid: (T, D/256-ish); each thread owns one (t, c) and walks the batch: dpos without atomics, dtok with atomics
    const int t = blockIdx.x;
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= D) return;
    float ps = 0.f;
    for (int b = 0; b < B; ++b) {
        const long row = (long)b * T + t;
        const float g = dx[row * D + c];
        ps += g;
        atomicAdd(dtok + ids[row] * D + c, g);
    }
    if (dpos) dpos[(long)t * D + c] += ps;
}

// ---------------------------------------------------------------------------------------------------------------
// conv front end: im2col / col2im
// ---------------------------------------------------------------------------------------------------------------
// mel f32 [B][C][T] -> xcol bf16 [B*T][kpad]; xcol[(b,t)][k*C+c] = mel[b][c][t+k-1]  (zero outside, zero pad cols)
__global__ __launch_bounds__(256) void im2col_mel_kernel(const float* mel, bf16* xcol, int C, int T, int kpad) {
    // block: 64 frames x all columns, staged through LDS so that both the mel reads (along t) and the xcol writes
    // (along the column index) are coalesced.
    __shared__ float s[66 * 129];  // [t_local + 1 (halo)][c] with c < 128 per pass
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * 64;
    for (int c0 = 0; c0 < C; c0 += 128) {
        const int nc = min(128, C - c0);
        for (int e = threadIdx.x; e < nc * 66; e += 256) {
            const int c = e / 66, tl = e - c * 66;
            const int t = t0 + tl - 1;
            s[tl * 129 + c] = (t >= 0 && t < T) ? mel[((long)b * C + c0 + c) * T + t] : 0.f;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * 3 * nc; e += 256) {
            const int tl = e / (3 * nc);
            const int r = e - tl * 3 * nc;
            const int k = r / nc, c = r - k * nc;
            if (t0 + tl < T) xcol[((long)b * T + t0 + tl) * kpad + k * C + c0 + c] = f2bf(s[(tl + k) * 129 + c]);
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < 64 * (kpad - 3 * C); e += 256) {
        const int tl = e / (kpad - 3 * C), c = e - tl * (kpad - 3 * C);
        if (t0 + tl < T) xcol[((long)b * T + t0 + tl) * kpad + 3 * C + c] = f2bf(0.f);
    }
}

// a bf16 [B*T][C] -> xcol bf16 [B*T/2][3C]; xcol[(b,t)][k*C+c] = a[b][2t+k-1][c]
__global__ __launch_bounds__(256) void im2col_s2_kernel(const bf16* a, bf16* xcol, int T, int C, long nvec) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nvec) return;
    const int cv = (3 * C) >> 3;
    const long orow = i / cv;
    const int col = (int)(i - orow * cv) * 8;
    const int k = col / C, c = col - k * C;
    const int To = T >> 1;
    const long b = orow / To;
    const int t = (int)(orow - b * To);
    const int ti = 2 * t + k - 1;
    bf16x8 v;
    if (ti >= 0 && ti < T) v = *(const bf16x8*)(a + (b * T + ti) * C + c);
    else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = f2bf(0.f);
    }
    *(bf16x8*)(xcol + orow * 3 * C + col) = v;
}

// dz[b][r][c] = gelu'(z[b][r][c]) * sum_{(t,k): 2t+k-1 = r} dxcol[(b,t)][k*C + c]
__global__ __launch_bounds__(256) void col2im_s2_gelu_bwd_kernel(const bf16* dxcol, const bf16* z, bf16* dz, int T,
                                                                 int C, long nvec) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nvec) return;
    const int cv = C >> 3;
    const long row = i / cv;
    const int c = (int)(i - row * cv) * 8;
    const long b = row / T;
    const int r = (int)(row - b * T);
    const int To = T >> 1;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    // r = 2t + k - 1  ->  k = r + 1 - 2t in {0,1,2}
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int num = r + 1 - k;
        if (num >= 0 && !(num & 1)) {
            const int t = num >> 1;
            if (t < To) {
                const bf16x8 v = *(const bf16x8*)(dxcol + (b * To + t) * 3 * C + k * C + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += bf2f(v[e]);
            }
        }
    }
    const bf16x8 zz = *(const bf16x8*)(z + row * C + c);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(round_bf16(acc[e]) * gelu_grad_f(bf2f(zz[e])));
    *(bf16x8*)(dz + row * C + c) = o;
}

// dz = bf16(dy) * gelu'(z)   (dy f32 or bf16)
template <bool YBF>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const void* dy, const bf16* z, bf16* dz, long n) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    f32x4 g;
    if (YBF) {
        const bf16x4 t = *(const bf16x4*)((const bf16*)dy + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = bf2f(t[e]);
    } else {
        g = *(const f32x4*)((const float*)dy + i);
    }
    const bf16x4 zz = *(const bf16x4*)(z + i);
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f2bf(round_bf16(g[e]) * gelu_grad_f(bf2f(zz[e])));
    *(bf16x4*)(dz + i) = o;
}

// w f32 [D][C][3] -> wp bf16 [D][kpad], wp[d][k*C+c] = w[d][c][k]
__global__ __launch_bounds__(256) void pack_conv_weight_kernel(const float* w, bf16* wp, int D, int C, int kpad) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)D * kpad) return;
    const long d = i / kpad;
    const int col = (int)(i - d * kpad);
    float v = 0.f;
    if (col < 3 * C) {
        const int k = col / C, c = col - k * C;
        v = w[(d * C + c) * 3 + k];
    }
    wp[i] = f2bf(v);
}
__global__ __launch_bounds__(256) void unpack_conv_grad_kernel(const float* gwp, float* gw, int D, int C, int kpad,
                                                               int accumulate) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)D * C * 3) return;
    const long d = i / (3 * C);
    const int r = (int)(i - d * 3 * C);
    const int c = r / 3, k = r - c * 3;
    const float v = gwp[d * kpad + k * C + c];
    gw[i] = accumulate ? gw[i] + v : v;
}

// ---------------------------------------------------------------------------------------------------------------
// casts / add / column sums
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* x, bf16* y, long n) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        const f32x4 v = *(const f32x4*)(x + i);
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
        *(bf16x4*)(y + i) = o;
    } else {
        for (long j = i; j < n; ++j) y[j] = f2bf(x[j]);
    }
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16* x, float* y, long n) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        const bf16x4 v = *(const bf16x4*)(x + i);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = bf2f(v[e]);
        *(f32x4*)(y + i) = o;
    } else {
        for (long j = i; j < n; ++j) y[j] = bf2f(x[j]);
    }
}

__device__ __forceinline__ float ldany(const void* p, int dt, long i) {
    return dt == DW_F32 ? ((const float*)p)[i] : bf2f(((const bf16*)p)[i]);
}
__global__ __launch_bounds__(256) void add_kernel(const void* a, int adt, const void* b, int bdt, void* y, int ydt,
                                                  long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = ldany(a, adt, i) + ldany(b, bdt, i);
        if (ydt == DW_F32) ((float*)y)[i] = v;
        else ((bf16*)y)[i] = f2bf(v);
    }
}

// out[n] (+)= sum_r x[r][n]; block = 16 column groups (8 cols each = 128 cols) x 16 row lanes
__global__ __launch_bounds__(256) void colsum_kernel(const bf16* x, long ld, int rows, int cols, float* out) {
    __shared__ float s[16][129];
    const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c0 = blockIdx.x * 128 + cg * 8;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (c0 < cols) {
        // four independent 16-byte loads in flight per lane (one load per iteration left the kernel latency-bound at
        // 2.3 TB/s: 164 k lanes x 16 bytes per memory round trip)
        const int step = gridDim.y * 16;
        int r = blockIdx.y * 16 + rl;
        for (; r + 3 * step < rows; r += 4 * step) {
            const bf16x8 v0 = *(const bf16x8*)(x + (long)r * ld + c0);
            const bf16x8 v1 = *(const bf16x8*)(x + (long)(r + step) * ld + c0);
            const bf16x8 v2 = *(const bf16x8*)(x + (long)(r + 2 * step) * ld + c0);
            const bf16x8 v3 = *(const bf16x8*)(x + (long)(r + 3 * step) * ld + c0);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (bf2f(v0[e]) + bf2f(v1[e])) + (bf2f(v2[e]) + bf2f(v3[e]));
        }
        for (; r < rows; r += step) {
            const bf16x8 v = *(const bf16x8*)(x + (long)r * ld + c0);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += bf2f(v[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) s[rl][cg * 8 + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 128) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += s[r][threadIdx.x];
        const int c = blockIdx.x * 128 + threadIdx.x;
        if (c < cols) atomicAdd(out + c, t);
    }
}
__global__ void zero_f32_kernel(float* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.f;
}

// ---------------------------------------------------------------------------------------------------------------
// self test: what does ds_read_b64_tr_b16 deliver?  LDS holds element ids 0..1023 (as bf16-exact small ints are not
// enough, ids are written as raw 16-bit patterns); lane l supplies byte address l*8.
// ---------------------------------------------------------------------------------------------------------------
__global__ void selftest_tr16_kernel(int32_t* out) {
    __shared__ __attribute__((aligned(16))) unsigned short s[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) s[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)((char*)s + lane * 8));
    union { bf16x4 b; unsigned short u[4]; } cv;
    cv.b = v;
#pragma unroll
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = cv.u[e];
}

// ---------------------------------------------------------------------------------------------------------------

// Rows of a matrix through an index list, 16 bytes per lane: gather dst[i] = src[idx[i]], scatter dst[idx[i]] = src[i].
// The packed (padding-free) teacher decoder keeps its activations as [live rows][D]; only attention wants the
// (batch, position) layout back (engine._layer_fwd).
__global__ __launch_bounds__(256) void move_rows_kernel(const char* src, long src_pitch, char* dst, long dst_pitch,
                                                        const int* idx, int n, int vpr, int scatter) {
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    const long i = v / vpr;
    if (i >= n) return;
    const int c = (int)(v - i * vpr) * 16;
    const long r = idx[i];
    const long sr = scatter ? i : r, dr = scatter ? r : i;
    *(f32x4*)(dst + dr * dst_pitch + c) = *(const f32x4*)(src + sr * src_pitch + c);
}

extern "C" int dw_version(void) {
    DW_CLEAR_ERR(); return 100; }

extern "C" int dw_embed_fwd(const int64_t* ids, const void* tok, const void* pos, int tab_dtype, void* out,
                            int out_dtype, int B, int T, int D, void* stream) {
    DW_CLEAR_ERR();
    if (!ids || !tok || !pos || !out || B <= 0 || T <= 0 || D <= 0 || (D & 3)) return DW_EINVAL;
    const long nvec = (long)B * T * (D >> 2);
    dim3 grid((nvec + 255) / 256), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (tab_dtype == DW_BF16 && out_dtype == DW_BF16)
        hipLaunchKernelGGL((embed_fwd_kernel<true, true>), grid, block, 0, s, ids, tok, pos, out, T, D, nvec);
    else if (tab_dtype == DW_BF16)
        hipLaunchKernelGGL((embed_fwd_kernel<true, false>), grid, block, 0, s, ids, tok, pos, out, T, D, nvec);
    else if (out_dtype == DW_BF16)
        hipLaunchKernelGGL((embed_fwd_kernel<false, true>), grid, block, 0, s, ids, tok, pos, out, T, D, nvec);
    else
        hipLaunchKernelGGL((embed_fwd_kernel<false, false>), grid, block, 0, s, ids, tok, pos, out, T, D, nvec);
    DW_CHECK_LAUNCH();
    return DW_OK;
}

extern "C" int dw_embed_bwd(const float* dx, const int64_t* ids, float* dtok, float* dpos, int B, int T, int D,
                            void* stream) {
    DW_CLEAR_ERR();
    if (!dx || !ids || !dtok || B <= 0 || T <= 0 || D <= 0) return DW_EINVAL;
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(T, (D + 255) / 256), dim3(256), 0, (hipStream_t)stream, dx, ids, dtok,
                       dpos, B, T, D);
    DW_CHECK_LAUNCH();
    return DW_OK;
}

extern "C" int dw_im2col_mel(const float* mel, void* xcol, int B, int C, int T, int kpad, void* stream) {
    DW_CLEAR_ERR();
    if (!mel || !xcol || B <= 0 || C <= 0 || T <= 0 || kpad < 3 * C || (kpad & 63)) return DW_EINVAL;
    hipLaunchKernelGGL(im2col_mel_kernel, dim3((T + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, mel, (bf16*)xcol,
                       C, T, kpad);
    DW_CHECK_LAUNCH();
    return DW_OK;
}

extern "C" int dw_im2col_s2(const void* a, void* xcol, int B, int T, int C, void* stream) {
    DW_CLEAR_ERR();
    if (!a || !xcol || B <= 0 || T <= 0 || (T & 1) || C <= 0 || (C & 7)) return DW_EINVAL;
    const long nvec = (long)B * (T / 2) * ((3 * C) >> 3);
    hipLaunchKernelGGL(im2col_s2_kernel, dim3((nvec + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const bf16*)a,
                       (bf16*)xcol, T, C, nvec);
    DW_CHECK_LAUNCH();
    return DW_OK;
}

extern "C" int dw_col2im_s2_gelu_bwd(const void* dxcol, const void* z, void* dz, int B, int T, int C, void* stream) {
    DW_CLEAR_ERR();
    if (!dxcol || !z || !dz || B <= 0 || T <= 0 || (T & 1) || C <= 0 || (C & 7)) return DW_EINVAL;
    const long nvec = (long)B * T * (C >> 3);
    hipLaunchKernelGGL(col2im_s2_gelu_bwd_kernel, dim3((nvec + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)dxcol, (const bf16*)z, (bf16*)dz, T, C, nvec);
    DW_CHECK_LAUNCH();
    return DW_OK;
}

extern "C" int dw_gelu_bwd(const void* dy, int dy_dtype, const void* z, void* dz, int64_t n, void* stream) {
    DW_CLEAR_ERR();
    if (!dy || !z || !dz || n <= 0 || (n & 3)) return DW_EINVAL;
    dim3 grid((n / 4 + 255) / 256), block(256);
    if (dy_dtype == DW_BF16)
        hipLaunchKernelGGL(gelu_bwd_kernel<true>, grid, block, 0, (hipStream_t)stream, dy, (const bf16*)z, (bf16*)dz,
                           (long)n);
    else
        hipLaunchKernelGGL(gelu_bwd_kernel<false>, grid, block, 0, (hipStream_t)stream, dy, (const bf16*)z, (bf16*)dz,
                           (long)n);
    DW_CHECK_LAUNCH();
    return DW_OK;
}

extern "C" int dw_pack_conv_weight(const float* w, void* wp, int D, int C, int kpad, void* stream) {
    DW_CLEAR_ERR();
    if (!w || !wp || D <= 0 || C <= 0 || kpad < 3 * C) return DW_EINVAL;
    const long n = (long)D * kpad;
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, (bf16*)wp,
                       D, C, kpad);
    DW_CHECK_LAUNCH();
    return DW_OK;
}

extern "C" int dw_unpack_conv_grad(const float* gwp, float* gw, int D, int C, int kpad, int accumulate, void* stream) {
    DW_CLEAR_ERR();
    if (!gwp || !gw || D <= 0 || C <= 0 || kpad < 3 * C) return DW_EINVAL;
    const long n = (long)D * C * 3;
    hipLaunchKernelGGL(unpack_conv_grad_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, gwp, gw, D, C,
                       kpad, accumulate);
    DW_CHECK_LAUNCH();
    return DW_OK;
}

extern "C" int dw_cast_f32_bf16(const float* x, void* y, int64_t n, void* stream) {
    DW_CLEAR_ERR();
    if (!x || !y || n <= 0 || ((uintptr_t)x & 15) || ((uintptr_t)y & 7)) return DW_EINVAL;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(((n + 3) / 4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, x,
                       (bf16*)y, (long)n);
    DW_CHECK_LAUNCH();
    return DW_OK;
}
extern "C" int dw_cast_bf16_f32(const void* x, float* y, int64_t n, void* stream) {
    DW_CLEAR_ERR();
    if (!x || !y || n <= 0 || ((uintptr_t)x & 7) || ((uintptr_t)y & 15)) return DW_EINVAL;
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(((n + 3) / 4 + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)x, y, (long)n);
    DW_CHECK_LAUNCH();
    return DW_OK;
}

extern "C" int dw_colsum_bf16(const void* x, int64_t ld, int rows, int cols, float* out, int accumulate,
                              void* stream) {
    DW_CLEAR_ERR();
    if (!x || !out || rows <= 0 || cols <= 0 || (cols & 7) || (ld & 7) || ((uintptr_t)x & 15)) return DW_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate) hipLaunchKernelGGL(zero_f32_kernel, dim3((cols + 255) / 256), dim3(256), 0, s, out, cols);
    int ry = (rows + 15) / 16;
    if (ry > 64) ry = 64;
    hipLaunchKernelGGL(colsum_kernel, dim3((cols + 127) / 128, ry), dim3(256), 0, s, (const bf16*)x, (long)ld, rows,
                       cols, out);
    DW_CHECK_LAUNCH();
    return DW_OK;
}

extern "C" int dw_add(const void* a, int a_dtype, const void* b, int b_dtype, void* y, int y_dtype, int64_t n,
                      void* stream) {
    DW_CLEAR_ERR();
    if (!a || !b || !y || n <= 0) return DW_EINVAL;
    long nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(add_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, a, a_dtype, b, b_dtype, y, y_dtype,
                       (long)n);
    DW_CHECK_LAUNCH();
    return DW_OK;
}

extern "C" int dw_move_rows(const void* src, int64_t src_pitch_bytes, void* dst, int64_t dst_pitch_bytes,
                            const int32_t* idx, int n, int row_bytes, int scatter, void* stream) {
    DW_CLEAR_ERR();
    if (!src || !dst || !idx || n <= 0 || row_bytes <= 0 || (row_bytes & 15) || (src_pitch_bytes & 15) ||
        (dst_pitch_bytes & 15) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15))
        return DW_EINVAL;
    const int vpr = row_bytes >> 4;
    const long nv = (long)n * vpr;
    hipLaunchKernelGGL(move_rows_kernel, dim3((nv + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const char*)src,
                       (long)src_pitch_bytes, (char*)dst, (long)dst_pitch_bytes, idx, n, vpr, scatter);
    DW_CHECK_LAUNCH();
    return DW_OK;
}

extern "C" int dw_selftest_tr16(int32_t* out, void* stream) {
    DW_CLEAR_ERR();
    if (!out) return DW_EINVAL;
    hipLaunchKernelGGL(selftest_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
    DW_CHECK_LAUNCH();
    return DW_OK;
}
