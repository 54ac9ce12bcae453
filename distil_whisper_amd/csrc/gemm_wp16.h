// The software-pipelined 8-wave main loop of gemm_wp.h on v_mfma_f32_16x16x32_bf16 instead of v_mfma_f32_32x32x16_bf16.
//
// Why: under the socket power limit the 16x16x32 shape sustains 10-13 % more flops than 32x32x16 on the same operands
// (tools/mfma_power_probe.hip: 2 012 vs 1 780 TFLOP/s MFMA-only, 1 700 vs 1 540 with the K loop's 24 fragment reads per K
// tile) -- half the accumulator register traffic per flop (4 of 16 accumulator registers read + written per 16 K MACs
// instead of 16 per 32 K).  The GEMM stream sits at that limit (profiles/r2_gemm_power_and_tiles.md), so the instruction
// shape is clock.  Same block tile, same operand DMA, same LDS bytes per K tile (24 fragment reads of 1 KiB per wave), same
// epilogue arithmetic; what changes:
//   * fragments are 16 rows x 32 k (lane & 15 = row, lane >> 4 = group of 8 k): the row-major LDS image is swizzled with
//     swz16 (conflict-free for that read pattern), the k-major image with one more row bit (frag_kmajor16);
//   * a K tile is two 32-deep k steps; each is run as two sub-steps over half of the wave's row blocks, so a sub-step is
//     16 (20 for the 320-row tile) MFMAs of 16 cycles -- as long as the four 16-deep sub-steps of gemm_wp.h.  The A
//     fragments of the two halves are the two register sets of a double buffer (the half that is not computing is being
//     refilled); the four B fragments are ONE set, refilled column by column during the second half of a k step right
//     behind their last use (the MFMAs run column-major within a sub-step).  48 / 56 fragment registers, as before;
//   * accumulators: f32x4 acc[row block][column block]; issued as (B fragment, A fragment), so lane & 15 = output row and
//     the four registers are four consecutive output columns (gemm_epilogue LAY = 16).
// The sum over k of an output element: 32 products per instruction instead of 16.  On gfx950 both instructions accumulate
// the products of one k row at a time into the fp32 accumulator, so the results are bit-identical to the 32x32x16 kernels'
// (asserted by tests/test_kernels_gpu.py::test_gemm_16x16x32_main_loop_is_bit_identical; the per-shape kernel choice in
// gemm.hip relies on it only for reproducibility of a step across dispatch rules, not for correctness -- the contract is the
// tolerance against oracle/ref_ops).
#pragma once
#include "gemm_wp.h"

__device__ __forceinline__ int swz16(int row) { return (row >> 1) & 7; }
// Fragment of a row-major [rows][64] tile (swz16) for v_mfma_f32_16x16x32_bf16: lane holds X[blk*16 + (lane&15)][ks*32 + 8*(lane>>4) + 0..7]
__device__ __forceinline__ bf16x8 frag_rows16(const char* tile, int blk, int ks, int lane) {
    const int row = blk * 16 + (lane & 15);
    const int ps = ((ks << 2) | (lane >> 4)) ^ swz16(row);
    return *(const bf16x8*)(tile + row * 128 + ps * 16);
}
// k-major image [64][BX]: 16-byte slot s of k row r is stored at slot s ^ swk16(r)
__device__ __forceinline__ int swk16(int krow) { return ((krow & 3) << 2) | (((krow >> 3) & 1) << 1); }
// Fragment X^T of the k-major tile for one 32-deep k step: lane holds X[k = ks*32 + 8*(lane>>4) + 0..7][x + (lane&15)].
// ds_read_b64_tr_b16 within a 16-lane group: lane p supplies 4 consecutive columns of k row (p >> 2); output lane i receives,
// as element j, element (i & 3) of the 8 bytes supplied by lane 4 j + (i >> 2) -- i.e. column 4 (i >> 2) + (i & 3) = i of k row j.
template <int BX>
__device__ __forceinline__ bf16x8 frag_kmajor16(const char* tile, int x, int ks, int lane) {
    constexpr int RB = BX * 2;
    const int g = lane >> 4, p = lane & 15;
    const int col = x + ((p & 3) << 2);
    const int k0 = ks * 32 + g * 8 + (p >> 2);
    const int sw = ((p >> 2) << 2) | ((g & 1) << 1);          // swk16(k0) = swk16(k0 + 4)
    const char* a0 = tile + k0 * RB + (((col >> 3) ^ sw) << 4) + ((p & 1) << 3);
    const char* a1 = a0 + 4 * RB;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)a0);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)a1);
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

#define PINQ(f) asm volatile("" : "+v"(f))

// TAG: no effect on the code -- a second symbol for the launches the small-M rule of gemm.hip sends here (outputs below two rounds
// of 256-row tiles), so that per-kernel profiles (rocprofv3, tools/pmc_traffic.py) can tell them from the step's wide launches.
template <bool TA, bool TB, int BM = 256, int DBG = 0, int WN = 4, int TAG = 0>
__global__ __launch_bounds__(WN * 128, WN == 4 ? 2 : 1) void gemm_wp16_kernel(const GemmP p) {
    // WN = 4: eight waves (two per SIMD), 128 x 64 (160 x 64) per wave.  WN = 2: FOUR waves, one per SIMD, 128 x 128 per wave --
    // 32 fragment reads per 128 MFMAs instead of 24 per 64: the LDS port (128 B per clock: 192 KiB of fragment reads + 64 KiB of
    // operand DMA per K tile against 2 048 MFMA cycles with eight waves) drops from ~100 % to ~75 % busy.  256 accumulator
    // registers per wave: the wave owns its SIMD's whole register file (256 VGPR + 256 AGPR).
    constexpr int BN = 256, WM = 2, NW = WM * WN, TN = BN / WN;
    static_assert(WN == 4 || (WN == 2 && BM == 256), "wave layouts: 2 x 4, or 2 x 2 on the 256-row tile");
    constexpr int FM = BM / WM / 16, FN = TN / 16;       // 16-row / 16-column blocks of a wave tile: 8 (10) x 4
    constexpr int HM = FM / 2;                           // row blocks of a sub-step
    static_assert(BM == 256 || !TA, "the k-major A image is built for 256-row tiles");
    constexpr int CPA = BM / 8 / NW, CPB = BN / 8 / NW;  // DMA pieces (1 KiB) per wave per operand per K tile
    constexpr bool UNI = BM != 256;
    constexpr int NA0 = (CPA + 1) / 2, NB0 = CPB / 2;
    constexpr int NH0 = NA0 + NB0, NH1 = CPA + CPB - NH0;
    constexpr int STAGE = (BM + BN) * 128;
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];
    __shared__ __attribute__((aligned(1024))) float bias_lds[BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * TN;
    const unsigned smem_w = wp_lds_addr(smem) + (unsigned)(wave * 1024);

    __shared__ int job_slot[2];
    GemmJobs jobs;
    gemm_jobs_begin(p, jobs, job_slot);
    while (jobs.cur < jobs.cnt) {
        // The K loop's per-lane addresses (operand-load offsets, fragment addresses) are functions of the lane index alone; derived
        // from an opaque copy they are recomputed per tile -- a few VALU instructions -- instead of living in 10-16 registers across
        // the epilogue, whose walks then spill their own addresses (a scratch reload is a VMEM load: its vmcnt wait also waits
        // for the previous slab's global stores, ~1 us per slab).
        int lane_k = lane;
        asm volatile("" : "+v"(lane_k));
        int tm, tn, ks;
        gemm_job_decode(p, jobs.start + jobs.cur, tm, tn, ks);
        const int m0 = tm * BM, n0 = tn * BN;

        // ---- DMA sources (as gemm_wp.h; the swizzles are this kernel's) ----
        unsigned offA[8], offB[8];
        const bf16* gA;
        const bf16* gB;
        int stepA, stepB;
        if (!TA) {
            gA = p.a + (long)m0 * p.lda;
            stepA = 128;
#pragma unroll
            for (int i = 0; i < CPA; ++i) {
                const int row = (wave + i * NW) * 8 + (lane_k >> 3);
                const int ls = (lane_k & 7) ^ swz16(row);
                const int grow = m0 + row < p.m ? row : p.m - 1 - m0;
                offA[i] = (unsigned)((grow * p.lda + ls * 8) * 2);
            }
        } else {
            gA = p.a + m0;
            stepA = 128 * (int)p.lda;
#pragma unroll
            for (int i = 0; i < CPA; ++i) {
                const int krow = (wave + i * NW) * 2 + (lane_k >> 5);
                const int ls = (lane_k & 31) ^ swk16(krow);
                const int gcol = m0 + ls * 8 < p.m ? ls * 8 : 0;
                offA[i] = (unsigned)((krow * p.lda + gcol) * 2);
            }
        }
        if (!TB) {
            gB = p.b + (long)n0 * p.ldb;
            stepB = 128;
#pragma unroll
            for (int i = 0; i < CPB; ++i) {
                const int row = (wave + i * NW) * 8 + (lane_k >> 3);
                const int ls = (lane_k & 7) ^ swz16(row);
                const int grow = n0 + row < p.n ? row : p.n - 1 - n0;
                offB[i] = (unsigned)((grow * p.ldb + ls * 8) * 2);
            }
        } else {
            gB = p.b + n0;
            stepB = 128 * (int)p.ldb;
#pragma unroll
            for (int i = 0; i < CPB; ++i) {
                const int krow = (wave + i * NW) * 2 + (lane_k >> 5);
                const int ls = (lane_k & 31) ^ swk16(krow);
                const int gcol = n0 + ls * 8 < p.n ? ls * 8 : 0;
                offB[i] = (unsigned)((krow * p.ldb + gcol) * 2);
            }
        }
        const int pieceA = UNI ? NW * (TA ? 2 : 8) * (int)p.lda * 2 : 0;
        const int pieceB = UNI ? NW * (TB ? 2 : 8) * (int)p.ldb * 2 : 0;
        int nt = p.k >> 6;
        {
            const int base = nt / p.split_k, rem = nt - base * p.split_k;
            const int first = ks * base + (ks < rem ? ks : rem);
            nt = base + (ks < rem ? 1 : 0);
            gA = (const bf16*)((const char*)gA + (long)first * stepA);
            gB = (const bf16*)((const char*)gB + (long)first * stepB);
        }
        const i32x4_t rsA = wp_rsrc(gA), rsB = wp_rsrc(gB);
        int kA = 0, kB = 0;

        f32x4 acc[FM][FN];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // load j of half h of the K tile at (kA, kB) into LDS buffer `buf` (A and B pieces alternate, A first)
        bool in_loop = false;      // (DBG ablations: 1 = no fragment reads, 2 = no operand DMA inside the K loop)
        auto dma1 = [&](auto hc, auto jc, int buf) __attribute__((always_inline)) {
            constexpr int h = decltype(hc)::value, j = decltype(jc)::value;
            if constexpr ((DBG & 2) != 0) { if (in_loop) return; }
            constexpr int na = h ? CPA - NA0 : NA0, nb = h ? CPB - NB0 : NB0, nmin = na < nb ? na : nb;
            constexpr bool isA = j < 2 * nmin ? (j & 1) == 0 : na > nb;
            constexpr int idx = j < 2 * nmin ? j / 2 : j - nmin;
            constexpr int i = (isA ? (h ? NA0 : 0) : (h ? NB0 : 0)) + idx;
            static_assert(j < na + nb, "load index");
            const unsigned lb = smem_w + (unsigned)(buf * STAGE);
            if constexpr (UNI) {
                if constexpr (isA) wp_dma16p<i * (NW * 1024)>(rsA, lb, offA[0], kA, i * pieceA);
                else wp_dma16p<i * (NW * 1024) + BM * 128, false>(rsB, lb, offB[0], kB, i * pieceB);
            } else {
                if constexpr (isA) wp_dma16i<i * (NW * 1024)>(rsA, lb, offA[i], kA);
                else wp_dma16i<i * (NW * 1024) + BM * 128, false>(rsB, lb, offB[i], kB);
            }
        };
        auto dma = [&](auto hc, int buf) __attribute__((always_inline)) {
            static_for<0, (decltype(hc)::value ? NH1 : NH0)>([&](auto jc) __attribute__((always_inline)) { dma1(hc, jc, buf); });
        };

        bf16x8 aX[HM], aY[HM], bq[FN];        // A fragments of row half 0 / half 1; the k step's B fragments
        auto ldA1 = [&](bf16x8 (&dst)[HM], auto ic, int kstep, int h, int buf) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            if constexpr ((DBG & 1) != 0) { if (in_loop) return; }
            const char* tA = smem + buf * STAGE;
            if (TA) dst[i] = frag_kmajor16<BM>(tA, wm0 + (h * HM + i) * 16, kstep, lane_k);
            else dst[i] = frag_rows16(tA, (wm0 >> 4) + h * HM + i, kstep, lane_k);
        };
        auto ldB1 = [&](auto jc, int kstep, int buf) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            if constexpr ((DBG & 1) != 0) { if (in_loop) return; }
            const char* tB = smem + buf * STAGE + BM * 128;
            if (TB) bq[j] = frag_kmajor16<BN>(tB, wn0 + j * 16, kstep, lane_k);
            else bq[j] = frag_rows16(tB, (wn0 >> 4) + j, kstep, lane_k);
        };
        // One sub-step: the HM x FN MFMAs of row half H (fragments `a`), column-major, in EXACTLY this order with its memory
        // instructions in exactly these gaps (a scheduling barrier closes every slot: left to itself the scheduler bunches the
        // fragment reads at the end of the sub-step, right in front of their first use):
        //   * VH = 1 / 2: the operand loads of DMA half VH - 1 into buffer dbuf, one per gap of column 0;
        //   * LA: the A fragments of the NEXT sub-step into the other register set (`an`, from (kn, hn) of LDS buffer bn), one
        //     per gap of the first column without DMA;
        //   * RB: the B fragments of the next k step (kb of buffer bb), each right behind the last MFMA of its column.
        auto substep = [&](auto hc, bf16x8 (&a)[HM], auto lac, bf16x8 (&an)[HM], int kn, int hn, int bn, auto rbc, int kb, int bb,
                           auto vhc, int dbuf) __attribute__((always_inline)) {
            constexpr int H = decltype(hc)::value & 1;
            constexpr bool TAIL = (decltype(hc)::value & 2) != 0;   // last K tile of the output tile (WN = 2: compiler-visible MFMAs)
            constexpr bool LA = decltype(lac)::value != 0, RB = decltype(rbc)::value != 0;
            constexpr int VH = decltype(vhc)::value;
            constexpr int NLD = VH == 1 ? NH0 : (VH == 2 ? NH1 : 0);
            // WN = 4: the operand loads ride on the MFMAs of column 0, the A fragment reads on column 1.  WN = 2: a lone wave per SIMD
            // stalls on its own VMEM issue when the four waves of the CU push 32 KiB of operand loads at once, so its loads are spread
            // over the whole sub-step (every DS-th slot, offset 2) and the A fragment reads sit on column 0 (slots 0..HM-1)
            constexpr int DS = WN == 2 && NLD > 0 ? HM * FN / NLD : 1, DO = WN == 2 ? 2 : 0;
            constexpr int A0 = WN == 2 ? 0 : (NLD > 0 ? HM : 0);
            static_assert(WN == 2 || NLD <= HM, "operand loads of a half ride on the MFMAs of column 0");
            static_assert(DO + (NLD > 0 ? NLD - 1 : 0) * DS < HM * FN, "operand load slots");
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, HM * FN>([&](auto qc) __attribute__((always_inline)) {
                constexpr int q = decltype(qc)::value, i = q % HM, j = q / HM;
                if constexpr (RB && i == 0 && j > 0) ldB1(std::integral_constant<int, j - 1>{}, kb, bb);
                if constexpr (WN == 2 && !TAIL)      // all 256 accumulator registers in AGPRs, by constraint: left to the allocator a fifth of
                    // them live in VGPRs and every MFMA on those is bracketed by v_accvgpr moves behind an s_nop 7
                    // (the asm is opaque to the hazard recognizer: the LAST K tile, whose results the epilogue's register moves read
                    // right behind the MFMA, uses the builtin)
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[H * HM + i][j]) : "v"(bq[j]), "v"(a[i]));
                else
                    acc[H * HM + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[j], a[i], acc[H * HM + i][j], 0, 0, 0);
                if constexpr (NLD > 0 && q >= DO && (q - DO) % DS == 0 && (q - DO) / DS < NLD)
                    dma1(std::integral_constant<int, (VH > 0 ? VH - 1 : 0)>{}, std::integral_constant<int, (q - DO) / DS>{}, dbuf);
                if constexpr (LA && q >= A0 && q < A0 + HM) ldA1(an, std::integral_constant<int, q - A0>{}, kn, hn, bn);
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (RB) { ldB1(std::integral_constant<int, FN - 1>{}, kb, bb); __builtin_amdgcn_sched_barrier(0); }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        auto pinA = [&](bf16x8 (&a)[HM]) __attribute__((always_inline)) {      // (ONE statement: one lgkmcnt wait for the set)
            if constexpr (HM == 4) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
            else asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[HM - 1]));
        };

        // ---- prologue: tile 0 whole, first half of tile 1, fragments of (tile 0, k step 0): A half 0 and B ----
        const bool tile_in = m0 + BM <= p.m && n0 + BN <= p.n;
        // (scalar base + 32-bit lane offset: as a per-lane 64-bit pointer the address was held across the tile loop, spilled, and its
        // reload -- a VMEM load -- waited with vmcnt(0) for the previous tile's stores in front of this tile's first operand load)
        if (p.bias && tile_in && wave == 0) glds16_so(p.bias + n0, (unsigned)lane_k * 16u, lds_addr_of(bias_lds));
        dma(I0{}, 0); dma(I1{}, 0);
        kA += stepA; kB += stepB;
        if (nt > 1) dma(I0{}, 1);
        gemm_jobs_prefetch(p, jobs, job_slot);      // (behind the operand loads: see gemm_wp.h)
        if (nt > 1) {
            if constexpr (NH0 == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (NH0 == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else { static_assert(NH0 == 4 || NH0 == 8 || NH0 == 5, "vmcnt immediate"); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        } else wait_vm0();
        __syncthreads();
        static_for<0, HM>([&](auto ic) __attribute__((always_inline)) { ldA1(aX, ic, 0, 0, 0); });
        static_for<0, FN>([&](auto jc) __attribute__((always_inline)) { ldB1(jc, 0, 0); });
        __builtin_amdgcn_sched_barrier(0);

        // one K tile; MORE1: tile t+1 exists, MORE2: tile t+2 exists.  (kA, kB) point at tile t+1 on entry.
        // Entering: aX = A(t, k step 0, half 0), bq = B(t, k step 0).
        auto body = [&](auto m1c, auto m2c, int t) {
            constexpr bool MORE1 = decltype(m1c)::value, MORE2 = decltype(m2c)::value;
            using H0 = std::integral_constant<int, MORE1 ? 0 : 2>;
            using H1 = std::integral_constant<int, MORE1 ? 1 : 3>;
            const int buf = t & 1;
            // sub-step 0: (k step 0, half 0); refill aY <- (k step 0, half 1); second half of tile t+1's DMA
            pinA(aX);
            if constexpr (MORE1) { substep(H0{}, aX, I1{}, aY, 0, 1, buf, I0{}, 0, 0, I2{}, buf ^ 1); kA += stepA; kB += stepB; }
            else substep(H0{}, aX, I1{}, aY, 0, 1, buf, I0{}, 0, 0, I0{}, 0);
            // sub-step 1: (k step 0, half 1); refill aX <- (k step 1, half 0), bq <- B(k step 1) column by column
            pinA(aY);
            substep(H1{}, aY, I1{}, aX, 1, 0, buf, I1{}, 1, buf, I0{}, 0);
            // sub-step 2: (k step 1, half 0); refill aY <- (k step 1, half 1)
            pinA(aX);
            substep(H0{}, aX, I1{}, aY, 1, 1, buf, I0{}, 0, 0, I0{}, 0);
            // every wave: its DMA pieces of tile t+1 have landed, its last fragments of tile t are in registers
            pinA(aY);
            if constexpr (MORE1) {
                wait_vm0();
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            // sub-step 3: (k step 1, half 1); refill aX, bq <- tile t+1, k step 0; first half of tile t+2's DMA
            if constexpr (MORE2) substep(H1{}, aY, I1{}, aX, 0, 0, buf ^ 1, I1{}, 0, buf ^ 1, I1{}, buf);
            else if constexpr (MORE1) substep(H1{}, aY, I1{}, aX, 0, 0, buf ^ 1, I1{}, 0, buf ^ 1, I0{}, 0);
            else substep(H1{}, aY, I0{}, aX, 0, 0, 0, I0{}, 0, 0, I0{}, 0);
        };
        int t = 0;
        in_loop = true;
        if constexpr ((DBG & 1) != 0) { static_for<0, HM>([&](auto ic) __attribute__((always_inline)) { aY[decltype(ic)::value] = aX[decltype(ic)::value]; }); }
        for (; t + 2 < nt; ++t) body(std::true_type{}, std::true_type{}, t);
        if (nt >= 2) { body(std::true_type{}, std::false_type{}, t); ++t; }
        body(std::false_type{}, std::false_type{}, t);

        if (!(p.stage_next & 16)) {
            if constexpr (WN == 4)
                gemm_epilogue<FM / 2, FN / 2, TN, (BM == 256 ? 8 : DW_EPF), GemmNoHook, true, 16>(p, acc, smem + STAGE, wave, lane, m0, wm0, n0, wn0, ks,
                                                                                            GemmNoHook(), tile_in ? bias_lds : nullptr, nullptr);
            else {
                // 128-column wave tile: the epilogue walks of the 64-column layout, once per column half
                gemm_epilogue<FM / 2, 2, 64, 8, GemmNoHook, true, 16, 0>(p, acc, smem + STAGE, wave, lane, m0, wm0, n0, wn0, ks,
                                                                         GemmNoHook(), tile_in ? bias_lds : nullptr, nullptr);
                gemm_epilogue<FM / 2, 2, 64, 8, GemmNoHook, true, 16, 4>(p, acc, smem + STAGE, wave, lane, m0, wm0, n0, wn0 + 64, ks,
                                                                         GemmNoHook(), tile_in ? bias_lds : nullptr, nullptr);
            }
        }
        else {
            float tsum = 0.f;
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) tsum += acc[i][j][r];
            if (tsum == 123.456f) *(float*)p.c = tsum;
        }
        gemm_lds_barrier();
        gemm_jobs_advance(jobs, job_slot);
    }
    gemm_jobs_end(p, jobs);
}

template <bool TA, bool TB, int BM = 256, int DBG = 0, int WN = 4, int TAG = 0>
static int launch_wp16(const GemmP& p0, hipStream_t s) {
    GemmP p = p0;
    const int tiles_m = (p.m + BM - 1) / BM;
    p.tiles_n = (p.n + 255) / 256;
    p.nwg = tiles_m * p.tiles_n;
    p.strip = gemm_strip_width(p.k, p.tiles_n, p.strip);
    int nblk = p.nwg * p.split_k;
    if (nblk > g_gemm_cus) nblk = g_gemm_cus;
    hipLaunchKernelGGL((gemm_wp16_kernel<TA, TB, BM, DBG, WN, TAG>), dim3(nblk), dim3(WN * 128), 0, s, p);
    DW_CHECK_LAUNCH();
    return DW_OK;
}
