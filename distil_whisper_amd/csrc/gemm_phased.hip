// Phase-pipelined main loop for the 256x256x64 bf16 MFMA GEMM (written for all four operand layouts; built for dX).
//
// Same tile, fragment layout, k order and fused epilogue as gemm.hip (bit-identical results); what differs is HOW a
// K tile moves through the CU:
//   * 8 waves (2 x 4), 128 x 64 of output per wave = 4 x 2 accumulators of 32x32 (128 accumulator registers);
//   * a K tile is consumed in 4 phases, one 64 x 32 output quadrant each (8 MFMAs: 2 row blocks x 4 k sub-steps):
//         phase 1 reads B0 + A0 -> (A0,B0)   phase 2 reads B1 -> (A0,B1)   phase 3 reads A1 -> (A1,B1)
//         phase 4 reads nothing -> (A1,B0)   (B0 stays in registers): 24 fragment reads per K tile instead of 32;
//   * every phase is [fragment reads | 2 operand DMAs | counted wait]  barrier  [8 MFMAs]  barrier, and the two wave
//     rows run ONE SLOT APART (the second row takes one extra barrier up front): on every SIMD one wave is in its MFMA
//     slot while the other one reads LDS and issues DMA, so the matrix pipe does not wait for ds_read round trips;
//   * the LDS image of a K tile is REGION-major: a region (16 KB) is the set of operand rows that all waves read in
//     the same phase -- A-X (phase 1), B-X (phase 1), B-Y (phase 2), A-Y (phase 3) -- so it becomes free for the K
//     tile after next as soon as that phase is over.  Operand DMA (global_load_lds_dwordx4) is issued one region per
//     phase and lands 9-11 slots (more than a whole K tile) before the region is read again.  Waits are COUNTED
//     (s_waitcnt vmcnt(8) in steady state): 8-10 DMAs per wave stay in flight across every barrier, the queue is
//     never drained inside the loop (the plain kernel drains it once per K tile);
//   * k-major operands (TA / TB: the backward GEMMs) use a [64 k][128] region image gathered by the per-lane DMA source
//     addresses and are read with ds_read_b64_tr_b16, as in gemm.hip.
// Slot arithmetic (t = K tile, slot 8t + 2(p-1) = read slot of phase p for wave row 0, one later for wave row 1):
//   a region last read in slot s may be overwritten from slot s + 2 (every reader has passed its lgkmcnt wait and a
//   barrier); A-X/B-X(t) are last read in slot 8t+1, B-Y(t) in 8t+3, A-Y(t) in 8t+5.  Issue schedule
//   (t,p1): B-Y(t+1)  (t,p2): A-Y(t+1)  (t,p3): A-X(t+2)  (t,p4): B-X(t+2)   -> earliest issue slots 8t, 8t+2, 8t+4,
//   8t+6 against frees at 8t-3, 8t-1, 8t+3, 8t+3.  A wave waits for ITS pieces of the region of phase p+1 at the end
//   of its read slot of phase p; every reader then passes at least one barrier before touching the region.
#include "gemm_common.h"

__device__ __forceinline__ void wait_vm(int n) {
    // counted wait on this wave's outstanding vector-memory operations (the DMA pieces are the only ones in the loop)
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    }
}

__device__ __forceinline__ void slot_barrier() {
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);   // nothing is scheduled across a slot boundary
}
// The MFMAs are pure register instructions: without data dependencies on something that is ordered against the
// barriers the compiler sinks them out of their slot (it did: whole MFMA clusters moved below the slot-closing barrier
// and were interleaved with the next phase's LDS reads).  PIN makes values opaque at a program point (an empty
// volatile asm keeps its order relative to the barrier builtins): fragments are pinned right AFTER the slot-opening
// barrier (the compiler's lgkmcnt wait for them lands there too), accumulators right BEFORE the slot-closing one.
#define PIN4(a) asm volatile("" : "+v"((a)[0]), "+v"((a)[1]), "+v"((a)[2]), "+v"((a)[3]))
#define PIN2(x, y) asm volatile("" : "+v"(x), "+v"(y))

constexpr int REGION = 16384;                       // one operand region of a K tile
constexpr int STAGE8 = 4 * REGION;                  // [A-X | A-Y | B-X | B-Y]

template <bool TA, bool TB, int PRIO, int STAGGER>
__global__ __launch_bounds__(512, 2) void gemm_phased_kernel(const GemmP p) {
    constexpr int BM = 256, BN = 256;
    constexpr int FM = 4, FN = 2, TN = 64;
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE8];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int wm0 = wr * 128, wn0 = wc * 64;

    // fragment read offsets of a row-major region image [128 rows][64 k] (swizzled 16-byte slots): row (lane & 31)
    // of a 32-row block, k sub-step kk
    int foff[4];
    {
        const int row = lane & 31;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) foff[kk] = row * 128 + ((((kk << 1) | (lane >> 5)) ^ swz7(row)) << 4);
    }

    __shared__ int job_slot[2];
    GemmJobs jobs;
    gemm_jobs_begin(p, jobs, job_slot);
    while (jobs.cur < jobs.cnt) {
        gemm_jobs_prefetch(p, jobs, job_slot);
        int tm, tn, ks;
        gemm_job_decode(p, jobs.start + jobs.cur, tm, tn, ks);
        const int m0 = tm * BM, n0 = tn * BN;

        // ---- per-lane DMA sources: 2 pieces (1 KB) per region per wave, piece j = 2 * wave + i of the region ----
        // row-major operand: piece j = region rows 8j .. 8j+7; region row r of A-X/A-Y is tile row (r/64)*128 + h*64 +
        // r%64, of B-X/B-Y tile row (r/32)*64 + h*32 + r%32 (h = 0 for X, 1 for Y).
        // k-major operand: piece j = k rows 4j .. 4j+3 of the [64][128] region image; region column c is tile column
        // (c/64)*128 + h*64 + c%64 (A) or (c/32)*64 + h*32 + c%32 (B).
        const bf16* srcA[2][2];
        const bf16* srcB[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int j = 2 * wave + i;
                if (!TA) {
                    const int r = j * 8 + (lane >> 3);
                    const int ls = (lane & 7) ^ swz7(r);
                    int grow = m0 + (r >> 6) * 128 + h * 64 + (r & 63);
                    grow = grow < p.m ? grow : p.m - 1;
                    srcA[h][i] = p.a + (long)grow * p.lda + ls * 8;
                } else {
                    const int krow = j * 4 + (lane >> 4);
                    const int c = (((lane & 15) ^ ((krow & 3) << 2))) << 3;
                    int gcol = m0 + (c >> 6) * 128 + h * 64 + (c & 63);
                    gcol = gcol < p.m ? gcol : m0;
                    srcA[h][i] = p.a + (long)krow * p.lda + gcol;
                }
                if (!TB) {
                    const int r = j * 8 + (lane >> 3);
                    const int ls = (lane & 7) ^ swz7(r);
                    int grow = n0 + (r >> 5) * 64 + h * 32 + (r & 31);
                    grow = grow < p.n ? grow : p.n - 1;
                    srcB[h][i] = p.b + (long)grow * p.ldb + ls * 8;
                } else {
                    const int krow = j * 4 + (lane >> 4);
                    const int c = (((lane & 15) ^ ((krow & 3) << 2))) << 3;
                    int gcol = n0 + (c >> 5) * 64 + h * 32 + (c & 31);
                    gcol = gcol < p.n ? gcol : n0;
                    srcB[h][i] = p.b + (long)krow * p.ldb + gcol;
                }
            }
        const long stepA = TA ? 64 * p.lda : 64, stepB = TB ? 64 * p.ldb : 64;
        int nt = p.k >> 6;
        {
            const int base = nt / p.split_k, rem = nt - base * p.split_k;
            const int first = ks * base + (ks < rem ? ks : rem);
            nt = base + (ks < rem ? 1 : 0);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 2; ++i) { srcA[h][i] += (long)first * stepA; srcB[h][i] += (long)first * stepB; }
        }

        f32x16 acc[FM][FN];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        // issue this wave's two pieces of region (A or B, h = X/Y) of K tile `kt` into buffer kt & 1
        auto issue_a = [&](int h, int kt) {
            char* t = smem + (kt & 1) * STAGE8 + h * REGION + 2 * wave * 1024;
#pragma unroll
            for (int i = 0; i < 2; ++i) glds16(srcA[h][i] + (long)kt * stepA, t + i * 1024);
        };
        auto issue_b = [&](int h, int kt) {
            char* t = smem + (kt & 1) * STAGE8 + (2 + h) * REGION + 2 * wave * 1024;
#pragma unroll
            for (int i = 0; i < 2; ++i) glds16(srcB[h][i] + (long)kt * stepB, t + i * 1024);
        };
        // fragments of one phase: A row blocks {2*wr, 2*wr+1} of region h, B block wc of region h
        auto read_a = [&](bf16x8 (&f)[2][4], const char* reg) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (TA) f[i][kk] = frag_kmajor<128>(reg, wr * 64 + i * 32, kk, lane);
                    else f[i][kk] = *(const bf16x8*)(reg + (wr * 2 + i) * 4096 + foff[kk]);
                }
        };
        auto read_b = [&](bf16x8 (&f)[4], const char* reg) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (TB) f[kk] = frag_kmajor<128>(reg, wc * 32, kk, lane);
                else f[kk] = *(const bf16x8*)(reg + wc * 4096 + foff[kk]);
            }
        };

        // ---- prologue: regions in issue order A-X(0) B-X(0) B-Y(0) A-Y(0) A-X(1) B-X(1) ----
        issue_a(0, 0); issue_b(0, 0); issue_b(1, 0); issue_a(1, 0);
        if (nt > 1) { issue_a(0, 1); issue_b(0, 1); }
        wait_vm(nt > 1 ? 8 : 4);                      // A-X(0), B-X(0) have landed (this wave's pieces)
        slot_barrier();
        if (STAGGER && wr == 1) slot_barrier();       // wave row 1 runs one slot behind wave row 0

        bf16x8 af[2][4], b0[4], b1[4];
        for (int t = 0; t < nt; ++t) {
            const char* buf = smem + (t & 1) * STAGE8;
            const bool more1 = t + 1 < nt, more2 = t + 2 < nt;
            // ---------------- phase 1: B0, A0 -> (A0, B0) ----------------
            read_b(b0, buf + 2 * REGION);
            read_a(af, buf);
            if (more1) issue_b(1, t + 1);
            wait_vm(more1 ? 8 : 2);                   // B-Y(t) landed: 4 later regions may still be in flight
            slot_barrier();
            PIN4(b0); PIN4(af[0]); PIN4(af[1]);
            if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0[kk], af[0][kk], acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0[kk], af[1][kk], acc[1][0], 0, 0, 0);
            }
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            PIN2(acc[0][0], acc[1][0]);
            slot_barrier();
            // ---------------- phase 2: B1 -> (A0, B1) ----------------
            read_b(b1, buf + 3 * REGION);
            if (more1) issue_a(1, t + 1);
            wait_vm(more1 ? 8 : 0);                   // A-Y(t) landed
            slot_barrier();
            PIN4(b1);
            if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1[kk], af[0][kk], acc[0][1], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1[kk], af[1][kk], acc[1][1], 0, 0, 0);
            }
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            PIN2(acc[0][1], acc[1][1]);
            slot_barrier();
            // ---------------- phase 3: A1 -> (A1, B1) ----------------
            read_a(af, buf + REGION);
            if (more2) issue_a(0, t + 2);
            slot_barrier();                           // (phase 4 reads nothing: no wait here)
            PIN4(af[0]); PIN4(af[1]);
            if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                acc[2][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1[kk], af[0][kk], acc[2][1], 0, 0, 0);
                acc[3][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1[kk], af[1][kk], acc[3][1], 0, 0, 0);
            }
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            PIN2(acc[2][1], acc[3][1]);
            slot_barrier();
            // ---------------- phase 4: (A1, B0) ----------------
            if (more2) issue_b(0, t + 2);
            wait_vm(more2 ? 8 : (more1 ? 4 : 0));     // A-X(t+1), B-X(t+1) landed
            slot_barrier();
            PIN4(b0);
            if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0[kk], af[0][kk], acc[2][0], 0, 0, 0);
                acc[3][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0[kk], af[1][kk], acc[3][0], 0, 0, 0);
            }
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            PIN2(acc[2][0], acc[3][0]);
            slot_barrier();
        }
        if (STAGGER && wr == 0) slot_barrier();       // both wave rows have executed the same number of barriers
        // (every DMA has landed: the last phases waited with vmcnt(0); gemm_epilogue starts with a full barrier)
        gemm_epilogue<FM, FN, TN, 2>(p, acc, smem, wave, lane, m0, wm0, n0, wn0, ks);
        __syncthreads();   // the LDS patches are reused as operand buffers by the next job
        gemm_jobs_advance(jobs, job_slot);
    }
    gemm_jobs_end(p, jobs);
}

// One build is kept: row-major A, k-major B (the dX GEMMs), wave rows staggered, no s_setprio -- the measured best of the
// variants tried on MI355X: with s_setprio around the MFMA slots -2 %, with both wave rows in lock step +3..7 % at
// K = 1280 but -1 % at K >= 3840; the other three operand layouts ran 5-20 % behind the plain kernel and are not
// instantiated (each instantiation costs ~40 s of compile time for its unrolled epilogue), profiles/r2_gemm_variants.md.
// TA / TB / PRIO / STAGGER stay template parameters of the kernel for such experiments.
int dw_gemm_phased_launch(const GemmP& p0, int ta, int tb, hipStream_t s) {
    if (ta || !tb) return DW_EINVAL;
    GemmP p = p0;
    const int tiles_m = (p.m + 255) / 256;
    p.tiles_n = (p.n + 255) / 256;
    p.nwg = tiles_m * p.tiles_n;
    p.strip = gemm_strip_width(p.k, p.tiles_n, p.strip);
    int nblk = p.nwg * p.split_k;
    if (nblk > g_gemm_cus) nblk = g_gemm_cus;
    hipLaunchKernelGGL((gemm_phased_kernel<false, true, 0, 1>), dim3(nblk), dim3(512), 0, s, p);
    DW_CHECK_LAUNCH();
    return DW_OK;
}
