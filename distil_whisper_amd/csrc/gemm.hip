// bf16 MFMA GEMM for gfx950 with fused epilogues -- the QKV / out-proj / FFN / LM-head / conv(im2col) contractions
// of the Whisper distillation step, forward and backward.
//
// Replaces nn.Linear/nn.Conv1d matmuls of the reference path (TF:modeling_whisper.py:279-282, 375-376, 444-445,
// 566-567, 970) and their autograd backward (run_distillation.py:1609).
//
// Structure (CDNA4-first, not a warp-shaped port):
//   * block tile BM x BN x 64, waves laid out WM x WN, every wave owns (BM/WM) x (BN/WN) made of 32x32 accumulators
//     fed by v_mfma_f32_32x32x16_bf16 (64-lane wavefront, 16 fp32 accumulators per lane per 32x32 tile);
//   * operands go HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip), double buffered, one barrier per
//     K-step; the LDS image is lane-linear, so the bank-conflict swizzle is applied to the per-lane SOURCE address
//     and undone on the fragment read (same involution on both sides);
//   * operands whose contraction index is NOT contiguous in memory (the backward GEMMs: dX = dY.W, dW = dY^T.X) are
//     staged k-major and their MFMA fragments are fetched with ds_read_b64_tr_b16 (hardware 4x16 transpose read), so
//     no transposed copies of activations or weights are ever materialised in HBM;
//   * blockIdx is remapped so that each XCD (private 4 MiB L2) walks a contiguous range of output tiles.
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
    int tiles_n, nwg;
};

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

template <int BM, int BN, int WM, int WN, bool TA, bool TB>
__global__ __launch_bounds__(64 * WM * WN) void gemm_kernel(const GemmP p) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 32, FN = TN / 32;
    constexpr int CA = (BM / 8) / NW;  // 1 KiB chunks per wave per stage (A)
    constexpr int CB = (BN / 8) / NW;
    constexpr int STAGE = (BM + BN) * 128;
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave / WN) * TM;
    const int wn0 = (wave % WN) * TN;

    // XCD-aware bijective remap: the dispatcher places block b on XCD b%8; give every XCD a contiguous tile range.
    int id;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, local = bid >> 3;
        const int q = p.nwg >> 3, r = p.nwg & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int tm = id / p.tiles_n, tn = id - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- per-lane source pointers of the staging loads (advance by one K-step per iteration) ----
    const bf16* srcA[CA];
    const bf16* srcB[CB];
    long stepA, stepB;
    if (!TA) {
        stepA = 64;
#pragma unroll
        for (int i = 0; i < CA; ++i) {
            const int c = wave + i * NW;
            const int row = c * 8 + (lane >> 3);
            const int ls = (lane & 7) ^ swz7(row);
            int grow = m0 + row;
            grow = grow < p.m ? grow : p.m - 1;
            srcA[i] = p.a + (long)grow * p.lda + ls * 8;
        }
    } else {
        constexpr int RPC = 1024 / (BM * 2), LPR = (BM * 2) / 16;
        stepA = 64 * p.lda;
#pragma unroll
        for (int i = 0; i < CA; ++i) {
            const int c = wave + i * NW;
            const int krow = c * RPC + lane / LPR;
            const int ps = lane % LPR;
            const int ls = ps ^ ((krow & 3) << 2);
            int gcol = m0 + ls * 8;
            gcol = gcol < p.m ? gcol : m0;
            srcA[i] = p.a + (long)krow * p.lda + gcol;
        }
    }
    if (!TB) {
        stepB = 64;
#pragma unroll
        for (int i = 0; i < CB; ++i) {
            const int c = wave + i * NW;
            const int row = c * 8 + (lane >> 3);
            const int ls = (lane & 7) ^ swz7(row);
            int grow = n0 + row;
            grow = grow < p.n ? grow : p.n - 1;
            srcB[i] = p.b + (long)grow * p.ldb + ls * 8;
        }
    } else {
        constexpr int RPC = 1024 / (BN * 2), LPR = (BN * 2) / 16;
        stepB = 64 * p.ldb;
#pragma unroll
        for (int i = 0; i < CB; ++i) {
            const int c = wave + i * NW;
            const int krow = c * RPC + lane / LPR;
            const int ps = lane % LPR;
            const int ls = ps ^ ((krow & 3) << 2);
            int gcol = n0 + ls * 8;
            gcol = gcol < p.n ? gcol : n0;
            srcB[i] = p.b + (long)krow * p.ldb + gcol;
        }
    }

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nt = p.k >> 6;

    auto stage = [&](int buf) {
        char* tA = smem + buf * STAGE;
        char* tB = tA + BM * 128;
#pragma unroll
        for (int i = 0; i < CA; ++i) {
            glds16(srcA[i], tA + (wave + i * NW) * 1024);
            srcA[i] += stepA;
        }
#pragma unroll
        for (int i = 0; i < CB; ++i) {
            glds16(srcB[i], tB + (wave + i * NW) * 1024);
            srcB[i] += stepB;
        }
    };

    stage(0);
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        wait_vm0();        // this wave's pieces of tile t have landed in LDS
        __syncthreads();   // ... everybody's have; and everybody finished reading buffer buf^1 (tile t-1)
        if (t + 1 < nt) stage(buf ^ 1);
        const char* tA = smem + buf * STAGE;
        const char* tB = tA + BM * 128;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 af[FM], bfr[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                if (TA) af[i] = frag_kmajor<BM>(tA, wm0 + i * 32, kk, lane);
                else af[i] = frag_rows(tA, (wm0 >> 5) + i, kk, lane);
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                if (TB) bfr[j] = frag_kmajor<BN>(tB, wn0 + j * 32, kk, lane);
                else bfr[j] = frag_rows(tB, (wn0 >> 5) + j, kk, lane);
            }
            static_for<0, FM>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                static_for<0, FN>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                });
            });
        }
    }

    // ---- epilogue: C/D layout of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
    // (compile-time indices only: a runtime-indexed accumulator array would be demoted to scratch memory)
    const int hi = lane >> 5, ln = lane & 31;
    const bool plain = !p.bias && !p.z_out && p.act == 0 && !p.zgrad && !p.r;
    static_for<0, FM>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        static_for<0, FN>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const int n = n0 + wn0 + j * 32 + ln;
            const bool n_ok = n < p.n;
            const float bv = (p.bias && n_ok) ? p.bias[n] : 0.f;
            static_for<0, 16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (n_ok && m < p.m) {
                    float v = acc[i][j][r];
                    if (!plain) {
                        v += bv;
                        if (p.z_out) p.z_out[(long)m * p.ldz + n] = f2bf(v);
                        if (p.act == 1) v = gelu_f(round_bf16(v));
                        if (p.zgrad) v *= gelu_grad_f(bf2f(p.zgrad[(long)m * p.ldzg + n]));
                        if (p.r) {
                            const int rr = p.r_row_mod > 0 ? (m % p.r_row_mod) : m;
                            const float rv = p.r_dtype == DW_F32 ? ((const float*)p.r)[(long)rr * p.ldr + n]
                                                                 : bf2f(((const bf16*)p.r)[(long)rr * p.ldr + n]);
                            v = (p.round_res ? round_bf16(v) : v) + rv;
                        }
                    }
                    if (p.c_dtype == DW_F32) ((float*)p.c)[(long)m * p.ldc + n] = v;
                    else ((bf16*)p.c)[(long)m * p.ldc + n] = f2bf(v);
                }
            });
        });
    });
}

template <int BM, int BN, int WM, int WN>
static int launch_tile(const GemmP& p0, int ta, int tb, hipStream_t s) {
    GemmP p = p0;
    const int tiles_m = (p.m + BM - 1) / BM;
    p.tiles_n = (p.n + BN - 1) / BN;
    p.nwg = tiles_m * p.tiles_n;
    dim3 grid(p.nwg), block(64 * WM * WN);
    if (!ta && !tb) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, false, false>), grid, block, 0, s, p);
    else if (!ta && tb) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, false, true>), grid, block, 0, s, p);
    else if (ta && !tb) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, true, false>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, true, true>), grid, block, 0, s, p);
    DW_CHECK_LAUNCH();
    return DW_OK;
}

extern "C" int dw_gemm_bf16(const DwGemm* g, void* stream) {
    DW_CLEAR_ERR();
    if (!g || !g->a || !g->b || !g->c) return DW_EINVAL;
    if (g->m <= 0 || g->n <= 0 || g->k <= 0 || (g->k & 63)) return DW_EINVAL;
    if ((g->lda & 7) || (g->ldb & 7)) return DW_EINVAL;
    if (((uintptr_t)g->a & 15) || ((uintptr_t)g->b & 15)) return DW_EINVAL;
    // k-major operands are fetched in 16-byte column slots: a slot whose first column is valid is read whole, so the
    // row must be readable up to the next multiple of 8 columns (true whenever ld covers the padded width)
    if (g->trans_a && g->lda < ((g->m + 7) & ~7)) return DW_EINVAL;
    if (g->trans_b && g->ldb < ((g->n + 7) & ~7)) return DW_EINVAL;
    GemmP p;
    p.a = (const bf16*)g->a; p.b = (const bf16*)g->b; p.c = g->c; p.bias = g->bias;
    p.z_out = (bf16*)g->z_out; p.zgrad = (const bf16*)g->zgrad_in; p.r = g->r;
    p.lda = g->lda; p.ldb = g->ldb; p.ldc = g->ldc; p.ldz = g->ldz; p.ldzg = g->ldzg; p.ldr = g->ldr;
    p.m = g->m; p.n = g->n; p.k = g->k;
    p.act = g->act; p.c_dtype = g->c_dtype; p.r_dtype = g->r_dtype; p.r_row_mod = g->r_row_mod;
    p.round_res = g->round_res; p.tiles_n = 0; p.nwg = 0;
    int tile = g->tile;
    if (tile != 128 && tile != 256) {
        const long t256 = (long)((g->m + 255) / 256) * ((g->n + 255) / 256);
        tile = t256 >= 512 ? 256 : 128;
    }
    hipStream_t s = (hipStream_t)stream;
    if (tile == 256) return launch_tile<256, 256, 2, 4>(p, g->trans_a, g->trans_b, s);
    return launch_tile<128, 128, 2, 2>(p, g->trans_a, g->trans_b, s);
}
