// bf16 MFMA GEMM for gfx950 with fused epilogues -- the QKV / out-proj / FFN / LM-head / conv(im2col) contractions
// of the Whisper distillation step, forward and backward.
//
// Replaces nn.Linear/nn.Conv1d matmuls of the reference path (TF:modeling_whisper.py:279-282, 375-376, 444-445,
// 566-567, 970) and their autograd backward (run_distillation.py:1609).
//
// Structure (CDNA4-first, not a warp-shaped port):
//   * block tile BM x BN x 64, waves laid out WM x WN, every wave owns (BM/WM) x (BN/WN) made of 32x32 accumulators
//     fed by v_mfma_f32_32x32x16_bf16 (64-lane wavefront, 16 fp32 accumulators per lane per 32x32 tile); the default
//     256x256 tile runs 16 waves (4x4, 64x64 per wave, <= 128 VGPRs): four resident waves per SIMD hide each
//     other's barrier / vmcnt / LDS latency better than the 8-wave layout with software-pipelined fragments did;
//   * operands go HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip), double buffered, one barrier per
//     K-step; the LDS image is lane-linear, so the bank-conflict swizzle is applied to the per-lane SOURCE address
//     and undone on the fragment read (same involution on both sides);
//   * operands whose contraction index is NOT contiguous in memory (the backward GEMMs: dX = dY.W, dW = dY^T.X) are
//     staged k-major and their MFMA fragments are fetched with ds_read_b64_tr_b16 (hardware 4x16 transpose read), so
//     no transposed copies of activations or weights are ever materialised in HBM;
//   * blockIdx is remapped so that each XCD (private 4 MiB L2) walks a contiguous range of output tiles.
#include "gemm_common.h"

template <int BM, int BN, int WM, int WN, bool TA, bool TB, int VAR>
__global__ __launch_bounds__(64 * WM * WN, (BM == 128 ? 2 : 1) * WM * WN / 4) void gemm_kernel(const GemmP p) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 32, FN = TN / 32;
    constexpr int CA = (BM / 8) / NW;  // 1 KiB chunks per wave per stage (A)
    constexpr int CB = (BN / 8) / NW;
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int PATCH = NW * 32 * (TN + 4) * 4;   // epilogue transposition patches (alias the operand buffers)
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE > PATCH ? 2 * STAGE : PATCH];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave / WN) * TM;
    const int wn0 = (wave % WN) * TN;

    int job_first, job_count, job_step;
    gemm_job_range(p, job_first, job_count, job_step);
  for (int job = 0; job < job_count; ++job) {
    int tm, tn, ks;
    gemm_job_decode(p, job_first + job * job_step, tm, tn, ks);
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

    int nt = p.k >> 6;
    {
        const int base = nt / p.split_k, rem = nt - base * p.split_k;
        const int first = ks * base + (ks < rem ? ks : rem);
        nt = base + (ks < rem ? 1 : 0);
#pragma unroll
        for (int i = 0; i < CA; ++i) srcA[i] += (long)first * stepA;
#pragma unroll
        for (int i = 0; i < CB; ++i) srcB[i] += (long)first * stepB;
    }

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

    // One K tile (64 deep) of MFMAs for this wave out of LDS buffer `buf`.
    auto compute = [&](int buf) {
        const char* tA = smem + buf * STAGE;
        const char* tB = tA + BM * 128;
        if (VAR == 0) {
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
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
                    });
                });
            }
        } else {
            // explicit register double buffer: the fragments of sub-step kk+1 are requested from LDS before the
            // MFMA cluster of sub-step kk is issued, so the LDS latency hides behind FM*FN matrix instructions
            bf16x8 af[2][FM], bfr[2][FN];
            auto load = [&](auto kc, auto sc) {
                constexpr int kk = decltype(kc)::value;
                constexpr int sl = decltype(sc)::value;
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    if (TA) af[sl][i] = frag_kmajor<BM>(tA, wm0 + i * 32, kk, lane);
                    else af[sl][i] = frag_rows(tA, (wm0 >> 5) + i, kk, lane);
                }
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    if (TB) bfr[sl][j] = frag_kmajor<BN>(tB, wn0 + j * 32, kk, lane);
                    else bfr[sl][j] = frag_rows(tB, (wn0 >> 5) + j, kk, lane);
                }
            };
            load(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            static_for<0, 4>([&](auto kc) {
                constexpr int kk = decltype(kc)::value;
                constexpr int cur = kk & 1;
                if constexpr (kk < 3) load(std::integral_constant<int, kk + 1>{}, std::integral_constant<int, cur ^ 1>{});
                static_for<0, FM>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    static_for<0, FN>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[cur][j], af[cur][i], acc[i][j], 0, 0, 0);
                    });
                });
                if (VAR == 1) {
                    // pin the interleave: LDS reads between consecutive MFMAs (DS_READ mask 0x100, MFMA 0x8)
#pragma unroll
                    for (int q = 0; q < FM * FN; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, (TA || TB) ? 2 : 1, 0);
                    }
                }
            });
        }
    };

    // lock-step schedule: every wave prefetches its pieces of tile t+1, then computes tile t (one barrier per tile).
    // Measured alternatives that did NOT pay on MI355X (kept out of the code): staging pieces
    // interleaved between MFMA clusters, s_setprio around the clusters, a two-group ping-pong schedule, and a 4-stage
    // ring of 32-deep stages with counted vmcnt across raw barriers.
    stage(0);
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        wait_vm0();        // this wave's pieces of tile t have landed in LDS
        __syncthreads();   // ... everybody's have; and everybody finished reading buffer buf^1 (tile t-1)
        if (t + 1 < nt) stage(buf ^ 1);
        compute(buf);
    }
    gemm_epilogue<FM, FN, TN, ((BM == 128 ? 2 : 1) * NW >= 16 ? 2 : 0)>(p, acc, smem, wave, lane, m0, wm0, n0, wn0, ks);
    __syncthreads();  // the LDS patches are reused as operand buffers by the next job
  }  // job loop
}

bool dw_gemm_skinny_ok(const GemmP& p, int trans_a, int trans_b);   // gemm_skinny.hip
int dw_gemm_skinny_launch(const GemmP& p, hipStream_t s);
int dw_gemm_phased_launch(const GemmP& p, int ta, int tb, int mode, hipStream_t s);  // gemm_phased.hip

extern int g_attn_bwd_stage;  // attention.hip
extern int g_attn_decode;
static int g_gemm_persistent = 1;
// 0/1: 8-wave 256 tile (1 = register double-buffered fragments); 2: 16-wave 256 tile; 3: + 8-wave 128 tile;
// 4/5/6: as 3, with the phase-pipelined kernel (gemm_phased.hip; 5 = without s_setprio, 6 = without the wave-row
// stagger) for every 256-tile GEMM; 7 (default): as 3, with the phase-pipelined kernel where it measured faster on
// MI355X -- k-major B operand and a long contraction (dX = dY.W with K >= 3840: +7..10 %; it loses 5-20 % on the
// short-K and row-major shapes, profiles/r2_gemm_variants.md)
static int g_gemm_variant = 7;
static int g_gemm_strip = 0;
extern "C" int dw_debug_set(int key, int value) {
    if (key == 0) { g_gemm_variant = value; return DW_OK; }
    if (key == 1) { g_gemm_strip = value; return DW_OK; }
    if (key == 2) { g_gemm_persistent = value; return DW_OK; }
    if (key == 3) { g_attn_bwd_stage = value; return DW_OK; }
    if (key == 4) { g_attn_decode = value; return DW_OK; }
    return DW_EINVAL;
}

template <int BM, int BN, int WM, int WN, int VAR>
static int launch_tile(const GemmP& p0, int ta, int tb, hipStream_t s) {
    GemmP p = p0;
    const int tiles_m = (p.m + BM - 1) / BM;
    p.tiles_n = (p.n + BN - 1) / BN;
    p.nwg = tiles_m * p.tiles_n;
    {
        // strip width: as many B tiles as fit ~4 MB (6 at K = 1280: measured best in the step, 452 vs 455 ms for 4
        // and 462 for 2); row-major when fewer than 3 fit
        const long tile_bytes = (long)BN * p.k * 2;
        int sw = (int)((4L << 20) / tile_bytes);
        if (sw < 3 || sw >= p.tiles_n) sw = p.tiles_n;
        p.strip = g_gemm_strip > 0 ? g_gemm_strip : sw;
    }
    // persistent launch for the 256-tile (one workgroup per CU, 256 CUs): only when there are more jobs than CUs
    int nblk = p.nwg * p.split_k;
    if (BM == 256 && nblk > 256 && g_gemm_persistent) nblk = 256;
    dim3 grid(nblk), block(64 * WM * WN);
    if (!ta && !tb) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, false, false, VAR>), grid, block, 0, s, p);
    else if (!ta && tb) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, false, true, VAR>), grid, block, 0, s, p);
    else if (ta && !tb) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, true, false, VAR>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, true, true, VAR>), grid, block, 0, s, p);
    DW_CHECK_LAUNCH();
    return DW_OK;
}

// out[i] (+)= sum_s part[s * stride + i]   (combination of split-K partial tiles; deterministic, no atomics)
__global__ __launch_bounds__(256) void reduce_slices_kernel(const float* part, long stride, int slices, float* out,
                                                            long n, int accumulate) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    f32x4 acc = accumulate ? *(const f32x4*)(out + i) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < slices; ++s) {
        const f32x4 v = *(const f32x4*)(part + s * stride + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += v[e];
    }
    *(f32x4*)(out + i) = acc;
}

extern "C" int dw_reduce_slices(const float* part, int64_t stride, int slices, float* out, int64_t n, int accumulate,
                                void* stream) {
    DW_CLEAR_ERR();
    if (!part || !out || slices < 1 || n <= 0 || (n & 3) || (stride & 3) || ((uintptr_t)part & 15) ||
        ((uintptr_t)out & 15))
        return DW_EINVAL;
    hipLaunchKernelGGL(reduce_slices_kernel, dim3((n / 4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, part,
                       (long)stride, slices, out, (long)n, accumulate);
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
    p.split_k = g->split_k > 1 ? g->split_k : 1;
    p.atomic = g->atomic_acc ? 1 : 0;
    if (p.split_k > (g->k >> 6)) p.split_k = g->k >> 6;
    p.slice_stride = 0;
    if (p.split_k > 1 && !p.atomic) {
        // K slices without atomics: every slice stores a plain fp32 partial at c + ks * slice_stride (the caller
        // reduces them, see dw_reduce_slices); only the bare epilogue makes sense here
        if (g->c_dtype != DW_F32 || g->bias || g->z_out || g->zgrad_in || g->r || g->act || g->slice_stride <= 0)
            return DW_EINVAL;
        p.slice_stride = g->slice_stride;
    }
    if (p.atomic && (g->c_dtype != DW_F32 || g->bias || g->z_out || g->zgrad_in || g->r || g->act)) return DW_EINVAL;
    {
        const int es = g->c_dtype == DW_F32 ? 4 : 2;
        bool v = ((uintptr_t)g->c % (4 * es)) == 0 && (g->ldc & 3) == 0;
        if (g->bias) v = v && ((uintptr_t)g->bias & 15) == 0;
        if (g->z_out) v = v && ((uintptr_t)g->z_out & 7) == 0 && (g->ldz & 3) == 0;
        if (g->zgrad_in) v = v && ((uintptr_t)g->zgrad_in & 7) == 0 && (g->ldzg & 3) == 0;
        if (g->r) v = v && ((uintptr_t)g->r % (g->r_dtype == DW_F32 ? 16 : 8)) == 0 && (g->ldr & 3) == 0;
        p.vec = v ? 1 : 0;
    }
    int tile = g->tile;
    hipStream_t s = (hipStream_t)stream;
    // decode regime (M = batch rows): weight-streaming kernel; tile = 16 requests it explicitly
    if (tile == 16 && !dw_gemm_skinny_ok(p, g->trans_a, g->trans_b)) return DW_EINVAL;
    if ((tile == 0 || tile == 16) && dw_gemm_skinny_ok(p, g->trans_a, g->trans_b)) return dw_gemm_skinny_launch(p, s);
    if (tile != 128 && tile != 256) {
        const long t256 = (long)((g->m + 255) / 256) * ((g->n + 255) / 256);
        tile = t256 >= 512 ? 256 : 128;
    }
    if (tile == 256) {
        if (g_gemm_variant >= 4 && g_gemm_variant <= 6) {
            p.strip = g_gemm_strip;
            return dw_gemm_phased_launch(p, g->trans_a, g->trans_b, g_gemm_variant - 4, s);
        }
        if (g_gemm_variant == 7 && !g->trans_a && g->trans_b && g->k >= 3840 && p.split_k == 1) {
            p.strip = g_gemm_strip;
            return dw_gemm_phased_launch(p, 0, 1, 1, s);
        }
        if (g_gemm_variant == 0) return launch_tile<256, 256, 2, 4, 0>(p, g->trans_a, g->trans_b, s);
        if (g_gemm_variant >= 2) return launch_tile<256, 256, 4, 4, 0>(p, g->trans_a, g->trans_b, s);
        return launch_tile<256, 256, 2, 4, 1>(p, g->trans_a, g->trans_b, s);
    }
    if (g_gemm_variant >= 3) return launch_tile<128, 128, 2, 4, 0>(p, g->trans_a, g->trans_b, s);
    return launch_tile<128, 128, 2, 2, 0>(p, g->trans_a, g->trans_b, s);
}
