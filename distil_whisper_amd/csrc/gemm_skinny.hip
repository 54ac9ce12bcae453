// Skinny-M bf16 GEMM for the KV-cache decode step: C[M,N] = A[M,K] . W[N,K]^T with M = decode batch (<= 64 rows).
//
// Replaces the nn.Linear calls of one `generate` token step (TF:modeling_whisper.py:279-282, 444-445, 970 on the
// cache branch 312-335; run_eval.py:739): q/k/v, out_proj, fc1, fc2 and the tied LM head with M = batch.
//
// This regime is pure weight streaming (every W byte is used M <= 64 times), so the kernel is laid out for
// memory-level parallelism instead of tile reuse:
//   * one workgroup per 16 output columns (W rows), its waves split K; a wave requests its WHOLE K slice of the
//     16 weight rows up front as 16-byte loads straight in v_mfma_f32_16x16x32_bf16 operand layout (lane = row,
//     8 consecutive k per lane; a 16-row x 64-byte footprint per load instruction, the workgroup's weights are one
//     contiguous 16*K*2-byte range of HBM), so all of W is in flight at once across the grid;
//   * the activations (M x K, L2 resident) are fetched in the same layout and fed as the other MFMA operand; the
//     accumulators come out transposed (lane = activation row m, 4 consecutive n per lane) so the epilogue stores
//     are 4-wide along N;
//   * the per-wave partial sums meet in LDS; the fused epilogue (bias, exact GELU, residual, fp32/bf16 store) is the
//     same arithmetic as the tile kernel's (gemm_common.h).
// Two fusions cut launches out of the token step (a kernel boundary costs 6-8 us there, more than most of these GEMVs):
//   * LN != 0: the A operand is bf16(LayerNorm(x)) built on load.  A lane keeps its slice of the row (<= 10 k steps x 8
//     values) in REGISTERS, so x is read once; mean and centred variance (two passes, like csrc/norm.hip) are combined
//     across the waves through LDS while the weight loads issued at kernel entry are still in flight.  (The first
//     attempt re-read x from L2 in three passes behind three barriers and was slower than the LayerNorm launch it saved.)
//   * kv_out: output columns >= kv_split are stored straight into the K/V cache row of their sequence position (the
//     append of TF:modeling_whisper.py:312-335) instead of a staging buffer + copy kernel.
#include "gemm_common.h"

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int SK_MAXS = 10;   // 32-deep k steps a wave keeps in registers (10 x 16 B of W + as much of A per lane)
constexpr int SK_MAXW = 16;   // waves per workgroup
int g_skinny_wide = 5;        // dw_debug_set key 8: bit 0 wide (64 columns per workgroup) variant for the LM head; bit 1: LayerNorm variants keep ONE column block per workgroup; bits 2-3: columns per workgroup of the projections back to d_model (4: 8 columns, the default; 0: 4; 12: 16)

// 8 consecutive elements of the LayerNorm input (f32 or bf16) as floats
template <bool XBF>
__device__ __forceinline__ void ld_x8(const void* x, long off, float (&v)[8]) {
    if (XBF) {
        const bf16x8 t = *(const bf16x8*)((const bf16*)x + off);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = bf2f(t[e]);
    } else {
        const f32x4_t a = *(const f32x4_t*)((const float*)x + off);
        const f32x4_t b = *(const f32x4_t*)((const float*)x + off + 4);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    }
}

// LN: 0 = A is a ready bf16 operand; 1 / 2 = A = bf16(LayerNorm(ln_x)), ln_x in f32 / bf16 (k <= 1280: <= 4 waves, so
// the kernel may use up to 512 registers per lane and keep the row slices resident)
// NB (LayerNorm variants only): 16-column weight blocks per workgroup.  A LayerNorm-on-load workgroup keeps ~300 registers per
// lane (row slices, gamma, beta, weight fragments): ONE workgroup per CU, so fc1's 320 blocks of 16 columns were two rounds of the
// 256 CUs (18.2 us against 10.5 for the 240 blocks of the QKV projection, profiles/r6_decode_step_sequence.md) and every
// workgroup repeated the row statistics of all 16 rows for 40 KB of weights.  With two blocks per workgroup the launch is 160 /
// 120 workgroups -- one round -- and the LayerNorm work per weight byte halves.
// VR (plain variant only): weight rows (output columns) a workgroup owns, 16 / 8 / 4.  A projection back to d_model has N / 16 = 80
// workgroups at D = 1280 -- 80 of the 256 CUs pull the weights, and a CU sustains ~25 GB/s of misses (fc2's 13 MB: 12 us against a
// 5.5 us launch floor).  With 8 or 4 valid rows per workgroup (the other lanes of the 16-row MFMA operand stay zero: the matrix
// pipe is idle here anyway) the same bytes are pulled by 160 / 320 workgroups.
template <int MB, int LN, int NB = 1, int VR = 16>
__global__ __launch_bounds__(LN ? 256 : 64 * SK_MAXW) void gemm_skinny_kernel(const GemmP p, int nw, int steps_total) {
    extern __shared__ float red[];  // [nw][NB][MB][64][4]
    static_assert(NB == 1 || LN != 0, "several column blocks per workgroup: the LayerNorm variants");
    static_assert(VR == 16 || (LN == 0 && NB == 1 && (VR == 8 || VR == 4)), "partial column blocks: the plain variant");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = blockIdx.x * (VR == 16 ? 16 * NB : VR);
    const int r16 = lane & 15, g = lane >> 4;
    // this wave's k steps: [s0, s0 + ns)
    const int base = steps_total / nw, rem = steps_total - base * nw;
    const int s0 = wave * base + (wave < rem ? wave : rem);
    const int ns = base + (wave < rem ? 1 : 0);

    bf16x8 wf[NB][SK_MAXS];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        int wrow = n0 + nb * 16 + r16;
        wrow = wrow < p.n ? wrow : p.n - 1;
        const bf16* wp = p.b + (long)wrow * p.ldb + (long)s0 * 32 + g * 8;
#pragma unroll
        for (int s = 0; s < SK_MAXS; ++s) {
            if constexpr (VR < 16) wf[nb][s] = bf16x8{};
            if (s < ns && (VR == 16 || r16 < VR)) wf[nb][s] = ld_stream<1>((const bf16x8*)(wp + s * 32));
        }
    }

    f32x4_t acc[NB][MB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[nb][mb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if constexpr (LN != 0) {
        constexpr bool XBF = LN == 2;
        // this lane's slices of rows mb*16 + r16: k = (s0 + s)*32 + g*8 + 0..7
        float xs[MB][SK_MAXS][8];
        float part[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            int arow = mb * 16 + r16;
            arow = arow < p.m ? arow : p.m - 1;
            float a1 = 0.f;
#pragma unroll
            for (int s = 0; s < SK_MAXS; ++s)
                if (s < ns) {
                    ld_x8<XBF>(p.ln_x, (long)arow * p.ld_lnx + (long)(s0 + s) * 32 + g * 8, xs[mb][s]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) a1 += xs[mb][s][e];
                }
            a1 += __shfl_xor(a1, 16);
            a1 += __shfl_xor(a1, 32);
            part[mb] = a1;
        }
        // LayerNorm parameters of this lane's columns (independent of the statistics: requested now)
        float gm[SK_MAXS][8], bt[SK_MAXS][8];
#pragma unroll
        for (int s = 0; s < SK_MAXS; ++s)
            if (s < ns) {
                ld_x8<false>(p.ln_g, (long)(s0 + s) * 32 + g * 8, gm[s]);
                ld_x8<false>(p.ln_b, (long)(s0 + s) * 32 + g * 8, bt[s]);
            }
        if (g == 0) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) red[(wave * MB + mb) * 16 + r16] = part[mb];
        }
        __syncthreads();
        float mu[MB], rs[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            float t = 0.f;
            for (int w = 0; w < nw; ++w) t += red[(w * MB + mb) * 16 + r16];
            mu[mb] = t / (float)p.k;
        }
        __syncthreads();
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            float a2 = 0.f;
#pragma unroll
            for (int s = 0; s < SK_MAXS; ++s)
                if (s < ns) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float d = xs[mb][s][e] - mu[mb]; a2 += d * d; }
                }
            a2 += __shfl_xor(a2, 16);
            a2 += __shfl_xor(a2, 32);
            part[mb] = a2;
        }
        if (g == 0) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) red[(wave * MB + mb) * 16 + r16] = part[mb];
        }
        __syncthreads();
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            float t = 0.f;
            for (int w = 0; w < nw; ++w) t += red[(w * MB + mb) * 16 + r16];
            rs[mb] = rsqrtf(t / (float)p.k + p.ln_eps);
        }
        __syncthreads();                               // `red` is reused for the accumulator exchange below
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
            for (int s = 0; s < SK_MAXS; ++s)
                if (s < ns) {
                    bf16x8 af;
#pragma unroll
                    for (int e = 0; e < 8; ++e) af[e] = f2bf((xs[mb][s][e] - mu[mb]) * rs[mb] * gm[s][e] + bt[s][e]);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nb][s], af, acc[nb][mb], 0, 0, 0);
                }
        }
    } else {
    static_for<0, MB>([&](auto mc) {
        constexpr int mb = decltype(mc)::value;
        int arow = mb * 16 + r16;
        arow = arow < p.m ? arow : p.m - 1;
        const bf16* ap = p.a + (long)arow * p.lda + (long)s0 * 32 + g * 8;
        bf16x8 af[SK_MAXS];
#pragma unroll
        for (int s = 0; s < SK_MAXS; ++s)
            if (s < ns) af[s] = *(const bf16x8*)(ap + s * 32);
#pragma unroll
        for (int s = 0; s < SK_MAXS; ++s)
            if (s < ns) acc[0][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][s], af[s], acc[0][mb], 0, 0, 0);
    });
    }
    // acc[mb][r] = partial C[m = mb*16 + (lane & 15)][n = n0 + 4*(lane >> 4) + r]
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) *(f32x4_t*)(red + (((wave * NB + nb) * MB + mb) * 64 + lane) * 4) = acc[nb][mb];
    __syncthreads();
    for (int idx = threadIdx.x; idx < NB * MB * 64; idx += blockDim.x) {
        const int nb = idx / (MB * 64), mb = (idx >> 6) % MB, l = idx & 63;
        f32x4_t v4 = {0.f, 0.f, 0.f, 0.f};
        for (int w = 0; w < nw; ++w) {
            const f32x4_t t = *(const f32x4_t*)(red + (((w * NB + nb) * MB + mb) * 64 + l) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v4[e] += t[e];
        }
        const int m = mb * 16 + (l & 15);
        const int n = n0 + nb * 16 + 4 * (l >> 4);
        if (VR < 16 && 4 * (l >> 4) >= VR) continue;
        if (m >= p.m || n >= p.n) continue;      // (n is a multiple of 4 and p.n of 16: all four columns are valid)
        float v[4] = {v4[0], v4[1], v4[2], v4[3]};
        if (p.bias) {
            const f32x4_t b4 = *(const f32x4_t*)(p.bias + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += b4[e];
        }
        if (p.act == 1) {
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                f32x2 x2; x2[0] = round_bf16(v[e]); x2[1] = round_bf16(v[e + 1]);
                const f32x2 g2 = gelu_fast2(x2);
                v[e] = g2[0]; v[e + 1] = g2[1];
            }
        }
        if (p.r) {
            float rv[4];
            if (p.r_dtype == DW_F32) {
                const f32x4_t r4 = *(const f32x4_t*)((const float*)p.r + (long)m * p.ldr + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) rv[e] = r4[e];
            } else {
                const bf16x4 r4 = *(const bf16x4*)((const bf16*)p.r + (long)m * p.ldr + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) rv[e] = bf2f(r4[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (p.round_res ? round_bf16(v[e]) : v[e]) + rv[e];
        }
        if (p.kv_out && n >= p.kv_split) {           // K/V columns: straight into the cache row of this position
            const int bq = m / p.kv_rpb;
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
            *(bf16x4*)(p.kv_out + ((long)bq * p.kv_pitch + p.kv_row0 + (m - bq * p.kv_rpb)) * p.kv_ld + (n - p.kv_split)) = o;
        } else if (p.c_dtype == DW_F32) {
            f32x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = v[e];
            *(f32x4_t*)((float*)p.c + (long)m * p.ldc + n) = o;
        } else {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
            *(bf16x4*)((bf16*)p.c + (long)m * p.ldc + n) = o;
        }
    }
}

// Wide variant for the tied LM head (N = 51 904 columns, plain epilogue): a workgroup owns 16 * NB output columns, so the
// activation fragments (fetched once per wave) serve NB weight blocks -- with one block per workgroup the L2 traffic of
// the activations (40 KB per workgroup) equals the HBM traffic of the weights it streams.  k <= 1280 (<= 4 waves,
// registers for NB x 10 weight fragments per lane).
template <int MB, int NB>
__global__ __launch_bounds__(256) void gemm_skinny_wide_kernel(const GemmP p, int nw, int steps_total) {
    extern __shared__ float red[];  // [nw][NB][MB][64][4]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = blockIdx.x * 16 * NB;
    const int r16 = lane & 15, g = lane >> 4;
    const int base = steps_total / nw, rem = steps_total - base * nw;
    const int s0 = wave * base + (wave < rem ? wave : rem);
    const int ns = base + (wave < rem ? 1 : 0);
    bf16x8 wf[NB][SK_MAXS];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        int wrow = n0 + nb * 16 + r16;
        wrow = wrow < p.n ? wrow : p.n - 1;
        const bf16* wp = p.b + (long)wrow * p.ldb + (long)s0 * 32 + g * 8;
#pragma unroll
        for (int s = 0; s < SK_MAXS; ++s)
            if (s < ns) wf[nb][s] = ld_stream<1>((const bf16x8*)(wp + s * 32));
    }
    f32x4_t acc[NB][MB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[nb][mb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    static_for<0, MB>([&](auto mc) {
        constexpr int mb = decltype(mc)::value;
        int arow = mb * 16 + r16;
        arow = arow < p.m ? arow : p.m - 1;
        const bf16* ap = p.a + (long)arow * p.lda + (long)s0 * 32 + g * 8;
        bf16x8 af[SK_MAXS];
#pragma unroll
        for (int s = 0; s < SK_MAXS; ++s)
            if (s < ns) af[s] = *(const bf16x8*)(ap + s * 32);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int s = 0; s < SK_MAXS; ++s)
                if (s < ns) acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nb][s], af[s], acc[nb][mb], 0, 0, 0);
    });
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) *(f32x4_t*)(red + (((wave * NB + nb) * MB + mb) * 64 + lane) * 4) = acc[nb][mb];
    __syncthreads();
    for (int idx = threadIdx.x; idx < NB * MB * 64; idx += blockDim.x) {
        const int nb = idx / (MB * 64), mb = (idx / 64) % MB, l = idx & 63;
        f32x4_t v4 = {0.f, 0.f, 0.f, 0.f};
        for (int w = 0; w < nw; ++w) {
            const f32x4_t t = *(const f32x4_t*)(red + (((w * NB + nb) * MB + mb) * 64 + l) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v4[e] += t[e];
        }
        const int m = mb * 16 + (l & 15);
        const int n = n0 + nb * 16 + 4 * (l >> 4);
        if (m >= p.m || n >= p.n) continue;
        if (p.c_dtype == DW_F32) *(f32x4_t*)((float*)p.c + (long)m * p.ldc + n) = v4;
        else {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f2bf(v4[e]);
            *(bf16x4*)((bf16*)p.c + (long)m * p.ldc + n) = o;
        }
    }
}

// true when the problem fits this kernel (the caller falls back to the tile kernel otherwise)
bool dw_gemm_skinny_ok(const GemmP& p, int trans_a, int trans_b) {
    if (trans_a || trans_b || p.m > 64 || p.split_k > 1 || p.atomic || p.z_out || p.zgrad || p.r_row_mod > 0) return false;
    if (!p.vec || (p.n & 15) || (p.k & 31) || p.k > SK_MAXS * SK_MAXW * 32) return false;
    if (p.ln_x && (p.m > 32 || p.k > SK_MAXS * 4 * 32)) return false;
    return true;
}

int dw_gemm_skinny_launch(const GemmP& p, hipStream_t s) {
    const int steps = p.k >> 5;
    int nw = (steps + 9) / 10;                    // ~10 steps (640 B of each weight row) per wave
    if (nw > SK_MAXW) nw = SK_MAXW;
    if ((steps + nw - 1) / nw > SK_MAXS) return DW_EINVAL;
    const int mb = (p.m + 15) / 16;
    if ((g_skinny_wide & 1) && !p.bias && !p.act && !p.r && !p.ln_x && !p.kv_out && nw <= 4 && p.n >= 16384 && mb <= 2) {
        constexpr int NB = 4;
        dim3 grid((p.n + 16 * NB - 1) / (16 * NB)), block(64 * nw);
        const size_t lds = (size_t)nw * NB * mb * 64 * 4 * sizeof(float);
        if (mb == 1) hipLaunchKernelGGL((gemm_skinny_wide_kernel<1, NB>), grid, block, lds, s, p, nw, steps);
        else hipLaunchKernelGGL((gemm_skinny_wide_kernel<2, NB>), grid, block, lds, s, p, nw, steps);
        DW_CHECK_LAUNCH();
        return DW_OK;
    }
    dim3 grid(p.n / 16), block(64 * nw);
    const size_t lds = (size_t)nw * (mb == 3 ? 4 : mb) * 64 * 4 * sizeof(float);
    if (p.ln_x) {
        const bool xbf = p.ln_x_dtype == DW_BF16;
        // two column blocks per workgroup when that is what brings the launch into one round of the CUs (or halves a round that
        // fits already, one row block only: the LayerNorm prologue is the same work per workgroup); dw_debug_set key 8 bit 1 = off
        // (measured at batch 16, D = 1280: fc1's 320 blocks 18.2 -> 12.5 us; the QKV projection's 240 blocks fit one round already
        // and LOSE 1.5 us as 120 workgroups -- fewer CUs pulling the weights)
        if (mb == 1 && (g_skinny_wide & 2) == 0 && p.n % 32 == 0 && p.n / 16 > 256) {
            dim3 grid2(p.n / 32);
            if (!xbf) hipLaunchKernelGGL((gemm_skinny_kernel<1, 1, 2>), grid2, block, 2 * lds, s, p, nw, steps);
            else hipLaunchKernelGGL((gemm_skinny_kernel<1, 2, 2>), grid2, block, 2 * lds, s, p, nw, steps);
            DW_CHECK_LAUNCH();
            return DW_OK;
        }
        if (mb == 1 && !xbf) hipLaunchKernelGGL((gemm_skinny_kernel<1, 1>), grid, block, lds, s, p, nw, steps);
        else if (mb == 1) hipLaunchKernelGGL((gemm_skinny_kernel<1, 2>), grid, block, lds, s, p, nw, steps);
        else if (!xbf) hipLaunchKernelGGL((gemm_skinny_kernel<2, 1>), grid, block, lds, s, p, nw, steps);
        else hipLaunchKernelGGL((gemm_skinny_kernel<2, 2>), grid, block, lds, s, p, nw, steps);
    } else if (mb == 1 && p.n / 16 <= 128 && (g_skinny_wide & 12) != 12) {
        // few column blocks (a projection back to d_model): 8 columns per workgroup so that 160 instead of 80 CUs pull the weights.
        // Measured at batch 16, D = 1280 (eager token step, tools/decode_profile.py): 16 columns 0.2195 ms, 8 columns 0.2161, 4 columns
        // 0.2319 -- every workgroup fetches the activation slice of all 16 rows, so more workgroups multiply the L2 traffic
        // (fc2: 160 KB per workgroup) and four columns lose what the spread wins.
        // (dw_debug_set key 8 bits 2-3: 0 -> 4 columns, 4 -> 8 columns, 12 -> 16 columns as before)
        if (g_skinny_wide & 4) hipLaunchKernelGGL((gemm_skinny_kernel<1, 0, 1, 8>), dim3(p.n / 8), block, lds, s, p, nw, steps);
        else hipLaunchKernelGGL((gemm_skinny_kernel<1, 0, 1, 4>), dim3(p.n / 4), block, lds, s, p, nw, steps);
    } else if (mb == 1) hipLaunchKernelGGL((gemm_skinny_kernel<1, 0>), grid, block, lds, s, p, nw, steps);
    else if (mb == 2) hipLaunchKernelGGL((gemm_skinny_kernel<2, 0>), grid, block, lds, s, p, nw, steps);
    else hipLaunchKernelGGL((gemm_skinny_kernel<4, 0>), grid, block, lds, s, p, nw, steps);
    DW_CHECK_LAUNCH();
    return DW_OK;
}
