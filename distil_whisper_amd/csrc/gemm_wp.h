// Software-pipelined main loop for the 256x256x64 bf16 MFMA GEMM (built as 8 waves = 2 x 4, 128 x 64 of output per wave).
//
// Same block tile, LDS image, k order and fused epilogue as the 16-wave kernel of gemm_kernel.h (results are
// bit-identical: every output element is the same fp32 chain over k); what changes is the shape of a wave's work.
//   * WM x WN waves, each owns (256/WM) x (256/WN) of output as 32x32 accumulators.  With 2 x 4 waves a wave holds
//     4 x 2 accumulators (128 registers), two waves share a SIMD: one wave's MFMAs run in the issue slots the other
//     spends on LDS reads and operand DMA, a 64-deep K tile costs a wave 24 fragment reads for 32 MFMAs (the 64x64
//     wave tiles of the 16-wave kernel: 16 reads for 16 MFMAs) and the workgroup's barrier has 8 participants, not 16.
//     (2 x 2 waves of 128 x 128 -- ONE wave per SIMD, 16 reads for 32 MFMAs -- also instantiates, but a lone wave
//     cannot cover its own DMA issue slots (~60+ cycles each against a 32-cycle MFMA): 4-10 % slower on every shape.)
//   * The K loop is software pipelined by hand.  Per K tile t a wave runs four sub-steps (one 16-deep k slice of all
//     its accumulators); under the MFMAs of sub-step s it requests the fragments of sub-step s+1 into the other half
//     of a register double buffer (sub-step 3 requests sub-step 0 of tile t+1 from the other LDS buffer) and issues
//     its share of the operand DMA (buffer_load_dwordx4 ... lds, 1 KiB pieces):
//         sub-step 0:  F(t,1)    DMA second half of tile t+1
//         sub-step 1:  F(t,2)
//         sub-step 2:  F(t,3)
//         -- s_waitcnt vmcnt(0) (issued >= 2 sub-steps earlier), fragments of sub-step 3 in registers, s_barrier --
//         sub-step 3:  F(t+1,0)  DMA first half of tile t+2 (into the buffer every wave has just finished reading)
//     ONE barrier per K tile; an LDS read has a whole sub-step to return, a DMA piece 2.5-3.5 sub-steps.
//   * Operand DMA uses buffer addressing: descriptor = uniform tile origin, voffset = the per-lane part (loop
//     invariant), soffset = the K advance: a K tile costs no per-lane address arithmetic.
//   * The instruction interleave inside a sub-step is pinned with sched_group_barrier (one memory instruction group
//     per MFMA gap); the loop is peeled into (steady, last-but-one, last) so the steady body has no branches.
//   * A deeper variant (five 32-deep stages in a 160 KiB ring, counted vmcnt, 4 stages of prefetch) was built and
//     measured: slower for row-major operands (one barrier per 32-deep stage), profiles/r2_gemm_variants.md.
#pragma once
#include "gemm_common.h"

// Make a fragment set opaque at a program point: the sub-step's MFMAs depend on it (they cannot be hoisted into the
// previous sub-step) and the compiler's lgkmcnt wait for the set lands here.
#define PINF(f) asm volatile("" : "+v"((f)[0]), "+v"((f)[1]), "+v"((f)[2]), "+v"((f)[3]))
#define PINF2(f) asm volatile("" : "+v"((f)[0]), "+v"((f)[1]))
#define PIN_SET(set)                                                                                              \
    do {        /* ONE statement per set: one lgkmcnt wait in front of the sub-step instead of one per fragment group */  \
        if constexpr (FM == 4 && FN == 2)                                                                                 \
            asm volatile("" : "+v"(af[set][0]), "+v"(af[set][1]), "+v"(af[set][2]), "+v"(af[set][3]), "+v"(bfr[set][0]), "+v"(bfr[set][1]));   \
        else if constexpr (FM == 5 && FN == 2)                                                                            \
            asm volatile("" : "+v"(af[set][0]), "+v"(af[set][1]), "+v"(af[set][2]), "+v"(af[set][3]), "+v"(af[set][4]), "+v"(bfr[set][0]),     \
                         "+v"(bfr[set][1]));                                                                              \
        else {                                                                                                            \
            if constexpr (FM == 4) PINF(af[set]);                                                                         \
            else if constexpr (FM == 5) { PINF(af[set]); asm volatile("" : "+v"(af[set][4])); }                          \
            else PINF2(af[set]);                                                                                          \
            if constexpr (FN == 4) PINF(bfr[set]); else PINF2(bfr[set]);                                                  \
        }                                                                                                                 \
    } while (0)

// One 16-byte-per-lane operand DMA (buffer_load_dwordx4 ... lds: LDS destination = M0 + lane * 16), written as inline
// assembly ON PURPOSE.  Issued through the builtin, the compiler models it as an LDS store it cannot disambiguate from
// the fragment reads of the OTHER K-tile buffer and puts `s_waitcnt vmcnt(0)` in front of the first ds_read after every
// DMA issue: the wave then sits out the whole operand latency once per K tile (the "parked on vmcnt" 40 % of the PMC
// profile), and the two-sub-step lead the pipeline gives each piece is never used.  Hidden from its memory model, the
// only vmcnt waits left are the explicit ones in front of the barriers.  (Giving the two buffers separate LDS
// variables + a K loop unrolled by two, so that alias scopes exist, also removes the wait but costs 8 address
// registers the kernel does not have: 1 700 spilled dwords.)
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
__device__ __forceinline__ i32x4_t wp_rsrc(const void* base) {
    const unsigned long long a = (unsigned long long)base;
    i32x4_t r;
    r[0] = (int)(unsigned)a; r[1] = (int)((unsigned)(a >> 32) & 0xffffu); r[2] = 0x7fffffff; r[3] = 0x00020000;
    return r;
}
__device__ __forceinline__ void wp_dma16(const i32x4_t& rsrc, const char* lds_dst, unsigned voffset, int soffset) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(const lds_void_t*)lds_dst);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(m0v), "v"(voffset), "s"(rsrc), "s"(soffset));
}
// The same with the LDS destination as a 32-bit LDS ADDRESS (integer arithmetic on the address of the operand buffers, taken
// once per kernel): the generic -> LDS pointer conversion of the form above costs a null check per call (s_cmp_lg_u64 /
// s_cselect_b32 / 64-bit add: four scalar instructions in front of every operand load of the K loop).
__device__ __forceinline__ unsigned wp_lds_addr(const void* lds_ptr) {
    return __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(const lds_void_t*)lds_ptr);
}
__device__ __forceinline__ void wp_dma16u(const i32x4_t& rsrc, unsigned lds_addr, unsigned voffset, int soffset) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voffset), "s"(rsrc), "s"(soffset));
}
// ... with the piece's compile-time offset added on the way into M0 (one scalar instruction instead of two)
// ... and, for the kernels whose pieces differ by a scalar stride (UNI), the piece's scalar offset computed in the wait state the
// M0 write needs anyway (instead of an s_nop and a separate s_add)
// (-DDW_DMA_NT=1 / 2 / 3: the loads of the A / B / both operands carry the non-temporal hint -- experiments, see gemm_store_out)
#ifndef DW_DMA_NT
#define DW_DMA_NT 0
#endif
template <int IMM, bool ISA = true>
__device__ __forceinline__ void wp_dma16p(const i32x4_t& rsrc, unsigned lds_base, unsigned voffset, int kbase, int piece_off) {
    int so;
    if constexpr ((DW_DMA_NT & (ISA ? 1 : 2)) != 0)
        asm volatile("s_add_u32 m0, %1, %5\n\ts_add_u32 %0, %4, %6\n\tbuffer_load_dwordx4 %2, %3, %0 offen nt lds"
                     : "=&s"(so) : "s"(lds_base), "v"(voffset), "s"(rsrc), "s"(kbase), "i"(IMM), "s"(piece_off) : "scc");
    else
        asm volatile("s_add_u32 m0, %1, %5\n\ts_add_u32 %0, %4, %6\n\tbuffer_load_dwordx4 %2, %3, %0 offen lds"
                     : "=&s"(so) : "s"(lds_base), "v"(voffset), "s"(rsrc), "s"(kbase), "i"(IMM), "s"(piece_off) : "scc");
}
template <int IMM, bool ISA = true>
__device__ __forceinline__ void wp_dma16i(const i32x4_t& rsrc, unsigned lds_base, unsigned voffset, int soffset) {
    if constexpr ((DW_DMA_NT & (ISA ? 1 : 2)) != 0)
        asm volatile("s_add_u32 m0, %0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds"
                     :: "s"(lds_base), "v"(voffset), "s"(rsrc), "s"(soffset), "i"(IMM) : "scc");
    else
        asm volatile("s_add_u32 m0, %0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     :: "s"(lds_base), "v"(voffset), "s"(rsrc), "s"(soffset), "i"(IMM) : "scc");
}

// ASMDMA = false builds the same kernel with the operand DMA issued through the compiler builtin (A/B reference only).
// BM = 320 (row-major A only): a 320 x 256 block tile, 160 x 64 per wave (5 x 2 accumulators).  M = 48000 is 150 row
// tiles exactly and N = 1280 / 2560 / 3840 give 750 / 1500 / 2250 tiles = 2.93 / 5.86 / 8.79 rounds of the 256 CUs
// where 256-row tiles give 940 / 1880 / 2820 = 3.67 / 7.34 / 11.02 (the last round a third full); per MFMA the wave
// reads 7 % fewer fragment bytes and the workgroup stages 10 % fewer operand bytes -- on a chip whose GEMM rate is set
// by the socket power limit (tools/gemm_power_probe.py) bytes moved per flop are what the clock is paid with.
// NST = 3 (128-row tile): a THREE-stage operand ring.  A sub-step of the 64 x 64 wave tile is four MFMAs, so the 2.5-3.5 sub-steps
// a DMA piece gets in the two-stage schedule are ~0.4 us -- less than its latency, and the wave sits out the rest at the K-tile
// boundary (measured: 1.1 us per K tile, the same as the lock-step 128 x 128 kernel).  With three stages the pieces of K tile
// t + 2 are requested during tile t and the boundary waits with a COUNTED vmcnt (the newest NH0 + NH1 loads may stay in flight:
// the counter retires in order), one barrier per K tile as before.
template <bool TA, bool TB, int WM, int WN, bool ASMDMA = true, int DBG = 0, int BM = 256, int NST = 2>
__global__ __launch_bounds__(64 * WM * WN, WM * WN / 4) void gemm_wp_kernel(const GemmP p) {
    constexpr int BN = 256, NW = WM * WN, FM = BM / WM / 32, FN = BN / WN / 32, TN = BN / WN;
    static_assert(NST == 2 || (NST == 3 && DBG == 0 && ASMDMA && BM < 256), "three stages: the 128-row tile");
    static_assert(NW * 32 * (TN + 4) * 4 <= 2 * (BM + BN) * 128, "epilogue patches must fit the operand buffers");
    static_assert(BM == 256 || !TA, "the k-major A image is built for 256-row tiles");
    constexpr int CPA = BM / 8 / NW, CPB = BN / 8 / NW;   // DMA pieces (1 KiB) per wave per operand per K tile
    constexpr int CP = CPB;
    constexpr bool UNI = BM == 320;                     // (BM = 128 keeps one clamped offset per piece like BM = 256: any M)
#ifndef DW_EPF
#define DW_EPF 4
#endif
    constexpr int EPF = DW_EPF;                       // epilogue side-input prefetch distance of the 320-row tile (register budget)
    static_assert(CPA * 8 * NW == BM && CPB * 8 * NW == BN && CPA <= 8 && CPB <= 8, "piece split");
    // a K tile is requested in two halves; half h carries A pieces [HA0(h), HA0(h+1)) and B pieces [HB0(h), HB0(h+1))
    constexpr int NA0 = (CPA + 1) / 2, NB0 = CPB / 2;          // pieces of half 0
    constexpr int NH0 = NA0 + NB0, NH1 = CPA + CPB - NH0;      // loads a wave issues per half
    constexpr int STAGE = (BM + BN) * 128;            // one K tile: A [256][64] | B [256][64] (or their k-major images)
    __shared__ __attribute__((aligned(1024))) char smem[NST * STAGE];
    __shared__ __attribute__((aligned(1024))) float bias_lds[BN];     // this tile's bias slice (see gemm_epilogue lds_bias)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * TN;
    const unsigned smem_w = wp_lds_addr(smem) + (unsigned)(wave * 1024);     // this wave's first DMA piece of buffer 0

    __shared__ int job_slot[2];
    GemmJobs jobs;
    gemm_jobs_begin(p, jobs, job_slot);
    if (p.stagger > 0) {
        // Experiment (dw_debug_set key 12 = S | unit << 8): the workgroups of a launch run their tiles in lockstep, so every
        // tile round ends with 256 CUs storing at once (33 MB in ~6 us = the HBM write rate) while the matrix pipes idle.
        // Start offsets (local index mod S) x unit x ~4 us spread the bursts; the dynamic job hand-out shares the tiles.
        const int k = ((blockIdx.x >> 3) % (p.stagger & 255)) * (p.stagger >> 8);
        for (int i = 0; i < k; ++i) __builtin_amdgcn_s_sleep(127);
    }
    // phase timestamps (100 MHz constant clock) of the first 8 tiles of every workgroup: 0 tile start, 1 first operand tile
    // landed, 2 K loop done, 3 epilogue returned (wave 0: its stores are issued), 4 behind the tile's last barrier
#define DW_TRACE(slot)                                                                                             \
    do {                                                                                                           \
        if (p.trace && tid == 0 && jobs.iter < 8)                                                                  \
            p.trace[((long)blockIdx.x * 8 + jobs.iter) * 8 + (slot)] = (long long)__builtin_amdgcn_s_memrealtime(); \
    } while (0)
    bool prefetched = false;     // this tile's first operand tile was requested before the previous tile's epilogue
    while (jobs.cur < jobs.cnt) {
        // The K loop's per-lane addresses (operand-load offsets, fragment addresses) are functions of the lane index alone; derived
        // from an opaque copy they are recomputed per tile -- a few VALU instructions -- instead of living in 10-16 registers across
        // the epilogue, whose walks then spill their own addresses (a scratch reload is a VMEM load: its vmcnt wait also waits
        // for the previous slab's global stores, ~1 us per slab).
        int lane_k = lane;
        asm volatile("" : "+v"(lane_k));
        DW_TRACE(0);
        gemm_jobs_prefetch(p, jobs, job_slot);
        int tm, tn, ks;
        gemm_job_decode(p, jobs.start + jobs.cur, tm, tn, ks);
        const int m0 = tm * BM, n0 = tn * BN;

        // ---- DMA sources: uniform base (advanced per K tile) + per-lane byte offset (loop invariant) ----
        // piece c = wave + 4 i (i = 0..7) of each operand: rows 8c..8c+7 of the row-major image (8 lanes of 16 B per
        // row, swizzled slot) or k rows 2c, 2c+1 of the k-major image [64][256] (32 lanes of 16 B per k row)
        // (buffer addressing: descriptor = uniform tile origin, voffset = the per-lane part, soffset = the K advance,
        // so a K tile costs no per-lane address arithmetic; all offsets stay below 2^31, checked by the launcher)
        unsigned offA[8], offB[8];              // (CP used; a dependent-size array captured by the lambdas below makes hipcc drop the host stub)
        const bf16* gA;
        const bf16* gB;
        int stepA, stepB;
        if (!TA) {
            gA = p.a + (long)m0 * p.lda;
            stepA = 128;
#pragma unroll
            for (int i = 0; i < CPA; ++i) {
                const int row = (wave + i * NW) * 8 + (lane_k >> 3);
                const int ls = (lane_k & 7) ^ swz7(row);
                const int grow = m0 + row < p.m ? row : p.m - 1 - m0;
                offA[i] = (unsigned)((grow * p.lda + ls * 8) * 2);
            }
        } else {
            gA = p.a + m0;
            stepA = 128 * (int)p.lda;
#pragma unroll
            for (int i = 0; i < CPA; ++i) {
                const int krow = (wave + i * NW) * 2 + (lane_k >> 5);
                const int ls = (lane_k & 31) ^ ((krow & 3) << 2);
                const int gcol = m0 + ls * 8 < p.m ? ls * 8 : 0;
                offA[i] = (unsigned)((krow * p.lda + gcol) * 2);
            }
        }
        if (!TB) {
            gB = p.b + (long)n0 * p.ldb;
            stepB = 128;
#pragma unroll
            for (int i = 0; i < CP; ++i) {
                const int row = (wave + i * NW) * 8 + (lane_k >> 3);
                const int ls = (lane_k & 7) ^ swz7(row);
                const int grow = n0 + row < p.n ? row : p.n - 1 - n0;
                offB[i] = (unsigned)((grow * p.ldb + ls * 8) * 2);
            }
        } else {
            gB = p.b + n0;
            stepB = 128 * (int)p.ldb;
#pragma unroll
            for (int i = 0; i < CP; ++i) {
                const int krow = (wave + i * NW) * 2 + (lane_k >> 5);
                const int ls = (lane_k & 31) ^ ((krow & 3) << 2);
                const int gcol = n0 + ls * 8 < p.n ? ls * 8 : 0;
                offB[i] = (unsigned)((krow * p.ldb + gcol) * 2);
            }
        }
        // UNI (320-row tiles; the launcher guarantees m % 320 == 0 and n % 256 == 0, so no row is ever clamped): the
        // pieces of an operand differ by a uniform byte stride, which goes into the scalar offset of the load -- ONE
        // address register per operand instead of CPA + CPB (the 160 accumulator + 56 fragment registers of this tile
        // leave no room for nine; a value spilled inside the K loop is reloaded with a VMEM instruction whose vmcnt
        // wait also waits for every operand load in flight)
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
        int kA = 0, kB = 0;                               // byte offset of the K tile the next DMA fetches

        f32x16 acc[FM][FN];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        // load j of half h (NH0 / NH1 loads per wave) of the K tile at (kA, kB) into LDS buffer `buf`: A and B pieces
        // alternate (A first) until one operand's share of the half is used up
        // (the descriptors travel as arguments, not as captures: with rsA / rsB captured by reference the 320-row kernel's
        // register allocation degrades -- 600 spilled SGPRs, 80 spilled VGPRs against 290 / 25)
        auto dma1x = [&](const i32x4_t& ra, const i32x4_t& rb, auto hc, auto jc, int buf) __attribute__((always_inline)) {
            constexpr int h = decltype(hc)::value, j = decltype(jc)::value;
            constexpr int na = h ? CPA - NA0 : NA0, nb = h ? CPB - NB0 : NB0, nmin = na < nb ? na : nb;
            constexpr bool isA = j < 2 * nmin ? (j & 1) == 0 : na > nb;
            constexpr int idx = j < 2 * nmin ? j / 2 : j - nmin;             // index within the operand's share of the half
            constexpr int i = (isA ? (h ? NA0 : 0) : (h ? NB0 : 0)) + idx;
            static_assert(j < na + nb, "load index");
            const char* tA = smem + ((DBG & 2) ? (buf & 1) : buf) * STAGE + wave * 1024 + i * (NW * 1024);
            if constexpr ((DBG & 2) != 0) { if (buf < 8) return; }                  // ablation: no operand DMA in the loop
            if constexpr (ASMDMA) {
                const unsigned lb = smem_w + (unsigned)(((DBG & 2) ? (buf & 1) : buf) * STAGE);
                if constexpr (UNI) {
                    if constexpr (isA) wp_dma16p<i * (NW * 1024)>(ra, lb, offA[0], kA, i * pieceA);
                    else wp_dma16p<i * (NW * 1024) + BM * 128, false>(rb, lb, offB[0], kB, i * pieceB);
                } else {
                    if constexpr (isA) wp_dma16i<i * (NW * 1024)>(ra, lb, offA[i], kA);
                    else wp_dma16i<i * (NW * 1024) + BM * 128, false>(rb, lb, offB[i], kB);
                }
            } else {
                const auto bA = __builtin_amdgcn_make_buffer_rsrc((void*)gA, 0, 0x7fffffff, 0x00020000);
                const auto bB = __builtin_amdgcn_make_buffer_rsrc((void*)gB, 0, 0x7fffffff, 0x00020000);
                if constexpr (isA)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(bA, (lds_void_t*)tA, 16, offA[i], kA, 0, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(bB, (lds_void_t*)(tA + BM * 128), 16, offB[i], kB, 0, 0);
            }
        };
        auto dma1 = [&](auto hc, auto jc, int buf) __attribute__((always_inline)) { dma1x(rsA, rsB, hc, jc, buf); };
        auto dma = [&](auto hc, int buf) __attribute__((always_inline)) {
            static_for<0, (decltype(hc)::value ? NH1 : NH0)>([&](auto jc) __attribute__((always_inline)) { dma1(hc, jc, buf); });
        };
        bf16x8 af[2][FM], bfr[2][FN];
        auto frags = [&](auto sc, auto kc, int buf) {
            constexpr int set = decltype(sc)::value, kk = decltype(kc)::value;
            if constexpr ((DBG & 1) != 0) { if (buf < 8) return; }                  // ablation: no fragment reads in the loop
            const char* tA = smem + ((DBG & 1) ? (buf & 1) : buf) * STAGE;
            const char* tB = tA + BM * 128;
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                if (TA) af[set][i] = frag_kmajor<BM>(tA, wm0 + i * 32, kk, lane_k);
                else af[set][i] = frag_rows(tA, (wm0 >> 5) + i, kk, lane_k);
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                if (TB) bfr[set][j] = frag_kmajor<BN>(tB, wn0 + j * 32, kk, lane_k);
                else bfr[set][j] = frag_rows(tB, (wn0 >> 5) + j, kk, lane_k);
            }
        };
        // The MFMAs of a sub-step with the memory instructions pinned between them: fragment reads in the first gaps
        // (sched_group_barrier: one MFMA, then a group of DS reads), then -- VH = 1 / 2 -- the CP operand loads of half
        // VH - 1, one per gap (inline assembly has no scheduling class: each is fenced into its gap).  The DMA writes a
        // buffer no wave reads any more, so its place behind the reads is a choice, not a dependence.
        // Masks: 0x8 MFMA, 0x100 DS read.
        auto mfma1 = [&](auto sc, auto qc) __attribute__((always_inline)) {
            constexpr int set = decltype(sc)::value, i = decltype(qc)::value / FN, j = decltype(qc)::value % FN;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[set][j], af[set][i], acc[i][j], 0, 0, 0);
        };
        auto substep = [&](auto sc, auto rc, auto vhc, int dbuf) __attribute__((always_inline)) {
            constexpr bool R = decltype(rc)::value != 0;
            constexpr int VH = decltype(vhc)::value;
            constexpr int NMF = FM * FN;
            constexpr int DSI = FM * (TA ? 2 : 1) + FN * (TB ? 2 : 1);   // DS instructions of the sub-step's fragments
            constexpr int RG = VH ? NMF / 4 : NMF / 2;                   // gaps that carry fragment reads
            constexpr int PER = (DSI + RG - 1) / RG;
            constexpr int NLD = VH == 1 ? NH0 : NH1;                       // operand loads of the half this sub-step issues
            static_assert(VH == 0 || RG + NLD <= NMF, "not enough MFMA gaps for the operand loads");
            static_for<0, RG>([&](auto qc) __attribute__((always_inline)) { mfma1(sc, qc); });
#pragma unroll
            for (int q = 0; q < RG; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                if (R) __builtin_amdgcn_sched_group_barrier(0x100, PER, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (VH != 0) {
                static_for<0, NLD>([&](auto jc) __attribute__((always_inline)) {
                    mfma1(sc, std::integral_constant<int, RG + decltype(jc)::value>{});
                    dma1(std::integral_constant<int, VH - 1>{}, jc, dbuf);
                    __builtin_amdgcn_sched_barrier(0);
                });
                static_for<RG + NLD, NMF>([&](auto qc) __attribute__((always_inline)) { mfma1(sc, qc); });
            } else {
                static_for<RG, NMF>([&](auto qc) __attribute__((always_inline)) { mfma1(sc, qc); });
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>;
        using I8 = std::integral_constant<int, 8>;
        (void)sizeof(I8);

        // ---- prologue: tile 0 whole, first half of tile 1, fragments of (0, 0) ----
        const bool tile_in = m0 + BM <= p.m && n0 + BN <= p.n;
        // (issued first: lands with tile 0; scalar base + 32-bit lane offset, see gemm_wp16.h)
        if (p.bias && tile_in && wave == 0) glds16_so(p.bias + n0, (unsigned)lane_k * 16u, lds_addr_of(bias_lds));
        if constexpr (NST == 3) {
            // ---- three-stage ring: tiles 0 and 1 whole, first half of tile 2; tile 0 has landed when only those are outstanding ----
            dma(I0{}, 0); dma(I1{}, 0);
            kA += stepA; kB += stepB;
            if (nt > 1) { dma(I0{}, 1); dma(I1{}, 1); kA += stepA; kB += stepB; }
            if (nt > 2) dma(I0{}, 2);
            if (nt > 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NH0 + NH1) : "memory");
            else if (nt > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NH0 + NH1) : "memory");
            else wait_vm0();
            __syncthreads();
            DW_TRACE(1);
            frags(I0{}, I0{}, 0);
            __builtin_amdgcn_sched_barrier(0);
            // one K tile out of buffer b0; b1 / b2: the buffers of tiles t+1 / t+2.  M1..M3: tile t+1 / t+2 / t+3 exists.
            // (kA, kB) point at tile t+2 on entry (its first half is in flight, its second half is issued in sub-step 0).
            auto body3 = [&](auto m1c, auto m2c, auto m3c, int b0, int b1, int b2) {
                constexpr bool M1 = decltype(m1c)::value, M2 = decltype(m2c)::value, M3 = decltype(m3c)::value;
                PIN_SET(0);
                frags(I1{}, I1{}, b0);
                if constexpr (M2) { substep(I0{}, I1{}, I2{}, b2); kA += stepA; kB += stepB; }
                else substep(I0{}, I1{}, I0{}, 0);
                PIN_SET(1);
                frags(I0{}, I2{}, b0);
                substep(I1{}, I1{}, I0{}, 0);
                PIN_SET(0);
                frags(I1{}, I3{}, b0);
                substep(I0{}, I1{}, I0{}, 0);
                PIN_SET(1);
                if constexpr (M1) {
                    // tile t+1 has landed once only tile t+2's pieces (requested after it) are outstanding
                    if constexpr (M2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NH0 + NH1) : "memory");
                    else wait_vm0();
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
                // sub-step 3: fragments of (t+1, 0); first half of tile t+3 into b0 (every wave is past its last read of it)
                if constexpr (M1) frags(I0{}, I0{}, b1);
                if constexpr (M3) substep(I1{}, I1{}, I1{}, b0);
                else if constexpr (M1) substep(I1{}, I1{}, I0{}, 0);
                else substep(I1{}, I0{}, I0{}, 0);
            };
            int t = 0, b0 = 0, b1 = 1, b2 = 2;
            auto rot = [&]() { const int x = b0; b0 = b1; b1 = b2; b2 = x; ++t; };
            for (; t + 3 < nt; rot()) body3(std::true_type{}, std::true_type{}, std::true_type{}, b0, b1, b2);
            if (nt >= 3) { body3(std::true_type{}, std::true_type{}, std::false_type{}, b0, b1, b2); rot(); }
            if (nt >= 2) { body3(std::true_type{}, std::false_type{}, std::false_type{}, b0, b1, b2); rot(); }
            body3(std::false_type{}, std::false_type{}, std::false_type{}, b0, b1, b2);
        } else {
        if (!prefetched) { dma(I0{}, (DBG & 2) ? 8 : 0); dma(I1{}, (DBG & 2) ? 8 : 0); }
        kA += stepA; kB += stepB;
        if (nt > 1) dma(I0{}, (DBG & 2) ? 9 : 1);
        if (nt > 1) {      // tile 0 has landed when only the NH0 loads of tile 1's first half are outstanding
            if constexpr (NH0 == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (NH0 == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else if constexpr (NH0 == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else { static_assert(NH0 == 4 || NH0 == 8 || NH0 == 5 || NH0 == 3, "vmcnt immediate"); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        } else wait_vm0();
        __syncthreads();
        DW_TRACE(1);
        frags(I0{}, I0{}, (DBG & 1) ? 8 : 0);
        if constexpr ((DBG & 1) != 0) frags(I1{}, I1{}, 8);
        __builtin_amdgcn_sched_barrier(0);

        // one K tile; MORE1: tile t+1 exists, MORE2: tile t+2 exists.  (kA, kB) point at tile t+1 on entry.
        auto body = [&](auto m1c, auto m2c, int t) {
            constexpr bool MORE1 = decltype(m1c)::value, MORE2 = decltype(m2c)::value;
            const int buf = t & 1;
            // sub-step 0: second half of tile t+1
            PIN_SET(0);
            frags(I1{}, I1{}, buf);
            if constexpr (MORE1) { substep(I0{}, I1{}, I2{}, buf ^ 1); if constexpr ((DBG & 4) == 0) { kA += stepA; kB += stepB; } }   // (DBG & 4: every DMA of the loop re-fetches the first K tiles -- always L2-warm; tools/gemm_dma_diag.py)
            else substep(I0{}, I1{}, I0{}, 0);
            // sub-step 1
            PIN_SET(1);
            frags(I0{}, I2{}, buf);
            substep(I1{}, I1{}, I0{}, 0);
            // sub-step 2
            PIN_SET(0);
            frags(I1{}, I3{}, buf);
            substep(I0{}, I1{}, I0{}, 0);
            // every wave: its DMA pieces of tile t+1 have landed, its last fragments of tile t are in registers
            PIN_SET(1);
            if constexpr (MORE1) {
                wait_vm0();
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            // sub-step 3: fragments of (t+1, 0), first half of tile t+2
            if constexpr (MORE1) frags(I0{}, I0{}, buf ^ 1);
            if constexpr (MORE2) substep(I1{}, I1{}, I1{}, buf);
            else if constexpr (MORE1) substep(I1{}, I1{}, I0{}, 0);
            else substep(I1{}, I0{}, I0{}, 0);
        };
        int t = 0;
        for (; t + 2 < nt; ++t) body(std::true_type{}, std::true_type{}, t);
        if (nt >= 2) { body(std::true_type{}, std::false_type{}, t); ++t; }
        body(std::false_type{}, std::false_type{}, t);
        }   // NST == 2

        DW_TRACE(2);
        // ---- optional (dw_debug_set key 11 bit 128, OFF by default): the NEXT tile's first operand tile, requested before
        // this tile's epilogue.  The epilogue's LDS patches live in operand buffer 1 only (swizzled, unpadded: 8 waves x
        // 8 KiB), so the next job's tile 0 can stream into buffer 0 while this tile's results are transposed and stored;
        // safe without a barrier when the K loop had an even number of tiles (its last tile sat in buffer 1; every wave
        // left buffer 0 behind the barrier inside the last-but-one tile) and both tiles are interior (same per-lane DMA
        // offsets, only the descriptors change).  Measured (profiles/r3_gemm_epilogue.md): the prologue wait shrinks from
        // 2.7 to 1.2 us and the epilogue grows by as much -- vmcnt retires in order and counts stores, so whichever wait
        // comes first after the epilogue's 32 stores sits out their acknowledgement; operand latency was never what the
        // prologue waited for.  Hence off.
        prefetched = false;
        if constexpr (ASMDMA && DBG == 0 && TN == 64 && BM >= 256) {
            const int nxt = jobs.dynamic ? job_slot[(jobs.iter + 1) & 1] : jobs.cur + jobs.step;
            if ((p.stage_next & 128) && (nt & 1) == 0 && nt >= 2 && nxt < jobs.cnt && tile_in && !p.zgrad && !p.r) {
                int tm2, tn2, ks2;
                gemm_job_decode(p, jobs.start + nxt, tm2, tn2, ks2);
                if ((tm2 + 1) * BM <= p.m && (tn2 + 1) * BN <= p.n) {
                    const int ntall = p.k >> 6, base2 = ntall / p.split_k, rem2 = ntall - base2 * p.split_k;
                    const long first2 = (long)ks2 * base2 + (ks2 < rem2 ? ks2 : rem2);
                    const char* a2 = (const char*)(TA ? p.a + tm2 * BM : p.a + (long)tm2 * BM * p.lda) + first2 * stepA;
                    const char* b2 = (const char*)(TB ? p.b + tn2 * BN : p.b + (long)tn2 * BN * p.ldb) + first2 * stepB;
                    const i32x4_t ra2 = wp_rsrc(a2), rb2 = wp_rsrc(b2);
                    kA = 0; kB = 0;
                    static_for<0, NH0>([&](auto jc) __attribute__((always_inline)) { dma1x(ra2, rb2, I0{}, jc, 0); });
                    static_for<0, NH1>([&](auto jc) __attribute__((always_inline)) { dma1x(ra2, rb2, I1{}, jc, 0); });
                    prefetched = true;
                }
            }
        }
        // (the eight swizzled 8 KiB patches fill operand buffer 1 of the 256- / 320-row tiles; the 128-row tile's 48 KiB stages are
        // smaller than that: its patches start at buffer 0 and run into buffer 1)
        if (!(p.stage_next & 16)) gemm_epilogue<FM, FN, TN, (BM == 320 ? EPF : 8), GemmNoHook, TN == 64>(p, acc, smem + (TN == 64 && STAGE >= NW * 8192 ? STAGE : 0), wave, lane, m0, wm0, n0, wn0, ks, GemmNoHook(), tile_in ? bias_lds : nullptr,
                                                           (p.trace && tid == 0 && jobs.iter < 8) ? p.trace + ((long)blockIdx.x * 8 + jobs.iter) * 8 : nullptr);
        else { float t = 0.f;
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) t += acc[i][j][r];
            if (t == 123.456f) *(float*)p.c = t; }
        DW_TRACE(3);
        gemm_lds_barrier();   // the LDS patches are reused as operand buffers by the next job (no vmcnt wait: the stores and
                              // the next tile's operand DMA stay in flight)
        DW_TRACE(4);
        gemm_jobs_advance(jobs, job_slot);
    }
    gemm_jobs_end(p, jobs);
}

template <bool TA, bool TB, int WM, int WN, bool ASMDMA = true, int DBG = 0, int BM = 256, int NST = 2>
static int launch_wp(const GemmP& p0, hipStream_t s) {
    GemmP p = p0;
    const int tiles_m = (p.m + BM - 1) / BM;
    p.tiles_n = (p.n + 255) / 256;
    p.nwg = tiles_m * p.tiles_n;
    p.strip = gemm_strip_width(p.k, p.tiles_n, p.strip);
    int nblk = p.nwg * p.split_k;
    if (nblk > g_gemm_cus) nblk = g_gemm_cus;
    hipLaunchKernelGGL((gemm_wp_kernel<TA, TB, WM, WN, ASMDMA, DBG, BM, NST>), dim3(nblk), dim3(64 * WM * WN), 0, s, p);
    DW_CHECK_LAUNCH();
    return DW_OK;
}
