// Tile kernel of the bf16 MFMA GEMM (see gemm.hip for the design notes) and its launcher template.  Included by
// gemm_tile256.hip and gemm_tile128.hip: one translation unit per block tile so that the four operand-layout
// instantiations of each compile in parallel (the fully unrolled fused epilogue makes every instantiation ~70k lines
// of ISA).
#pragma once
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

    __shared__ int job_slot[2];
    GemmJobs jobs;
    gemm_jobs_begin(p, jobs, job_slot);
  while (jobs.cur < jobs.cnt) {
      gemm_jobs_prefetch(p, jobs, job_slot);
      int tm, tn, ks;
      gemm_job_decode(p, jobs.start + jobs.cur, tm, tn, ks);
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
    gemm_jobs_advance(jobs, job_slot);
  }  // job loop
  gemm_jobs_end(p, jobs);
}

// launch knobs owned by gemm.hip (dw_debug_set)
extern int g_gemm_persistent;
extern int g_gemm_strip;

template <int BM, int BN, int WM, int WN, int VAR>
static int launch_tile(const GemmP& p0, int ta, int tb, hipStream_t s) {
    GemmP p = p0;
    const int tiles_m = (p.m + BM - 1) / BM;
    p.tiles_n = (p.n + BN - 1) / BN;
    p.nwg = tiles_m * p.tiles_n;
    p.strip = gemm_strip_width(p.k, p.tiles_n, g_gemm_strip);
    // persistent launch for the 256-tile (one workgroup per CU, 256 CUs): only when there are more jobs than CUs
    int nblk = p.nwg * p.split_k;
    if (BM == 256 && nblk > g_gemm_cus && g_gemm_persistent) nblk = g_gemm_cus;
    dim3 grid(nblk), block(64 * WM * WN);
    if (!ta && !tb) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, false, false, VAR>), grid, block, 0, s, p);
    else if (!ta && tb) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, false, true, VAR>), grid, block, 0, s, p);
    else if (ta && !tb) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, true, false, VAR>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, true, true, VAR>), grid, block, 0, s, p);
    DW_CHECK_LAUNCH();
    return DW_OK;
}
