// Flash-style scaled-dot-product attention for gfx950, head_dim 64, forward + backward.
//
// Replaces the attention core the reference reaches through `sdpa_attention_forward` / `eager_attention_forward`
// (TF:modeling_whisper.py:215-238, 337-351) for its three shape classes: encoder self (1500x1500, no mask), decoder
// self (447x447, causal) and cross (447x1500, no mask); no padding mask ever reaches attention in training.
//
// CDNA4 design: everything is computed "transposed" so that the softmax row statistics are lane-local:
//   S^T[key][q] = K . Q^T     (A = K fragment from LDS, B = Q fragment held in registers)
//   O^T[d][q]   = V^T . P^T   (A = V^T fragment via ds_read_b64_tr_b16, B = P^T straight from the S^T registers)
// In the 32x32 accumulator layout the column (= query) is lane&31, so max / sum / rescale never cross lanes except
// for one lane^32 exchange, and P never goes through LDS: the MFMA k-slot order of the P^T operand is simply
// *defined* as the order in which the S^T accumulator registers hold the keys (key = 16s + 8(e>>2) + 4hi + (e&3)),
// and the V^T operand is fetched with the same key permutation by choosing the tr-read row addresses.
// The two backward kernels reuse the same skeleton with the roles of the stationary (register) and streamed (LDS)
// operands swapped; dK/dV and dQ are produced by separate passes so no atomics are needed (deterministic).
#include "common.h"
#include "../../include/dwamd.h"

#define NEG_BIG (-1.0e30f)
#ifndef DW_ATTN_DEFER
#define DW_ATTN_DEFER 8
#endif
#ifndef DW_ATTN_IDLE_SKIP
#define DW_ATTN_IDLE_SKIP 1   // a wave whose 32 stationary rows lie wholly behind the end of the sequence skips its tiles' arithmetic
#endif
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct AttnP {
    const bf16* q; const bf16* k; const bf16* v; bf16* o; float* lse;
    const bf16* d_o; float* delta; bf16* dq; bf16* dk; bf16* dv;
    long ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
    int B, H, Lq, Lk;
    long q_rows, kv_rows;  // rows between consecutive batches in memory (= Lq / Lk unless reading a padded KV cache)
    float* dq_colsum;      // f32 [B][H*64] += per-batch column sums of the stored dq (q_proj.bias gradient), or null
    float* dv_colsum;      // f32 [B][H*64] += per-batch column sums of the stored dv (v_proj.bias gradient), or null
    const int* q_start; const int* q_len;     // ragged batches (attn_fwd_kernel VL): packed-row offset and length of every sequence
    const int* kv_start; const int* kv_len;   // ... of its keys / values (null: rectangular K / V, batch min(b, kv_B - 1))
    int kv_B;
    int plain_order;       // 1 = workgroups take (tile, head, batch) in launch order (A/B switch of attn_workgroup)
    float defer;           // deferred-maximum threshold of the forward softmax in log2 units (g_attn_defer; 0 = the exact rule)
    int coff;              // causal mask: key <= query + coff (0 = top-left aligned, Lk - Lq = bottom-right aligned)
    float scale;
};

// registers r = 8*s2 .. 8*s2+7 of a 32x32 accumulator -> bf16x8 MFMA operand
__device__ __forceinline__ bf16x8 pack8(const f32x16& x, int s2) {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = f2bf(x[s2 * 8 + e]);
    return r;
}

// 16-byte global load of 8 bf16 of the stationary operand: row `row`, elements col..col+7
__device__ __forceinline__ bf16x8 ldg8(const bf16* base, long row, long ld, int col) {
    return *(const bf16x8*)(base + row * ld + col);
}

// store a transposed 32x32 accumulator block: lane (row = lane&31, hi), regs r -> column cb*32 + (r&3)+8*(r>>2)+4*hi
__device__ __forceinline__ void store_t(bf16* base, long ld, long row, bool ok, int cb, int hi, const f32x16& a,
                                        float mul) {
    if (!ok) return;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        bf16x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = f2bf(a[g * 4 + e] * mul);
#ifdef DW_NT_ATTN
        __builtin_nontemporal_store(w, (bf16x4*)(base + row * ld + cb * 32 + g * 8 + hi * 4));
#else
        *(bf16x4*)(base + row * ld + cb * 32 + g * 8 + hi * 4) = w;
#endif
    }
}

// The same store through a wave-private LDS patch (32 rows x 144 B): the lanes hold a ROW each (8-byte pieces of it per
// register group), so a direct store instruction touches 64 different 128-byte lines for 8 bytes each; turned row-major in
// LDS, a store instruction writes 8 complete 128-byte rows (16 bytes per lane).  Without any output store the kernels run
// 5 % (encoder shape) to 13 % (cross-attention backward) faster -- the row-per-lane stores were most of that (section 14).
// `patch`: this wave's 4608 bytes; the caller has made sure that no wave still reads the operand tiles it overlays.
#ifndef DW_ATTN_ROWSTORE
#define DW_ATTN_ROWSTORE 1
#endif
__device__ __forceinline__ void store_rows(bf16* base, long ld, int row0, int nrows, char* patch, int lane,
                                           const f32x16& a0, const f32x16& a1, float mul) {
    const int hi = lane >> 5, ln = lane & 31;
    char* const wr = patch + ln * 144 + hi * 8;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const f32x16& a = cb == 0 ? a0 : a1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x4 w;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = f2bf(a[g * 4 + e] * mul);
            *(bf16x4*)(wr + cb * 64 + g * 16) = w;
        }
    }
    // (wave-private: the compiler's lgkmcnt wait orders the writes before the reads)
    const int rr = lane >> 3, sc = lane & 7;
    u32x4 o[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) o[it] = *(const u32x4*)(patch + (it * 8 + rr) * 144 + sc * 16);
    const unsigned loff = (unsigned)(rr * (int)ld + sc * 8) * 2u;
    char* const rb = (char*)(base + (long)row0 * ld);
#pragma unroll
    for (int it = 0; it < 4; ++it)
        if (row0 + it * 8 + rr < nrows) *(u32x4*)(rb + (long)it * 8 * ld * 2 + loff) = o[it];
}

// Column sums of a transposed 32x32 accumulator block as it is stored (bf16-rounded, rows that are not `ok` excluded),
// added to colsum[cb*32 + ...]: the bias gradient of the projection whose output gradient the block is -- one pass of
// lane exchanges at the end of the kernel instead of a separate kernel reading the whole dq / dv matrix again.
__device__ __forceinline__ void colsum_t(float* colsum, bool ok, int cb, int hi, int ln, const f32x16& a, float mul) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = ok ? bf2f(f2bf(a[r] * mul)) : 0.f;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) v += __shfl_xor(v, o);
        if (ln == 0) atomicAdd(colsum + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, v);
    }
}

// Workgroup -> (tile of stationary rows, head, batch).  The dispatcher places workgroup L of a 1-D grid on XCD L % 8 and
// every XCD has a private 4 MiB L2: each XCD gets a contiguous range of (batch, head) pairs with ALL their tiles, in
// dispatch order, so the rows a pair streams (K/V, or Q/dO in the dK/dV pass) are fetched from the fabric into one L2
// instead of into up to eight (a 3-D grid, tile index fastest, sprayed the 12 tiles of a pair over all XCDs: 3.6x the
// algorithmic fetch bytes, profiles/r3_pmc_traffic.json).
__device__ __forceinline__ void attn_workgroup(int ntile, int H, bool plain, int& tile, int& h, int& b) {
    const int n = gridDim.x, L = blockIdx.x;
    const int xcd = L & 7, slot = L >> 3, q = n >> 3, r = n & 7;
    const int logical = plain ? L : xcd * q + min(xcd, r) + slot;   // (plain: dw_debug_set key 18, the A/B switch)
    tile = logical % ntile;
    const int bh = logical / ntile;
    h = bh % H;
    b = bh / H;
}

// ---------------------------------------------------------------------------------------------------------------
// forward: block = NW waves x 32 queries; key/value tiles of 64 rows double-buffered in LDS
// ---------------------------------------------------------------------------------------------------------------
// ABL (builds with -DDW_ABLATE only, tools/attn_ablate.py): timing experiments that leave out one resource user each and
// compute garbage -- 1 no exp, 2 / 4 K / V fragments from registers instead of LDS, 8 no operand staging inside the loop,
// 16 no barrier, 32 / 64 without the QK / PV MFMAs.
template <bool CAUSAL, int NW = 4, int ABL = 0, bool VL = false>
__global__ __launch_bounds__(64 * NW, CAUSAL ? 2 : 4) void attn_fwd_kernel(const AttnP p) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * 16384];  // [buf][K tile 8K | V tile 8K]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5, ln = lane & 31;
    int tile, h, b;
    attn_workgroup((p.Lq + 32 * NW - 1) / (32 * NW), p.H, p.plain_order, tile, h, b);
    const int qb0 = tile * (32 * NW);
    // VL (ragged batches, forward-only passes over packed rows): sequence b owns query rows [q_start[b], q_start[b] + q_len[b]) of
    // the packed buffers, and -- self-attention -- the same rows as keys / values (kv_start == q_start); cross-attention keeps the
    // rectangular K / V of batch min(b, kv_B - 1) (entries behind kv_B: filler rows of a quantised row count, any finite result).
    // p.Lq is the longest sequence (the grid's tile count); workgroups behind their sequence's last query leave at once.
    long q_row0 = (long)b * p.q_rows, kv_row0 = (long)b * p.kv_rows;
    int Lq = p.Lq, Lk = p.Lk;
    if constexpr (VL) {
        q_row0 = p.q_start[b];
        Lq = p.q_len[b];
        if (p.kv_start) { kv_row0 = p.kv_start[b]; Lk = p.kv_len[b]; }
        else kv_row0 = (long)min(b, p.kv_B - 1) * p.kv_rows;
        if (qb0 >= Lq) return;
    }
    const int q = qb0 + wave * 32 + ln;          // this lane's query (column of S^T)
    const bool q_ok = q < Lq;
    const int qc = q_ok ? q : Lq - 1;
    const bf16* Q = p.q + q_row0 * p.ldq + h * 64;
    const bf16* K = p.k + kv_row0 * p.ldk + h * 64;
    const bf16* V = p.v + kv_row0 * p.ldv + h * 64;

    bf16x8 qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = ldg8(Q, qc, p.ldq, kk * 16 + hi * 8);

    int nkt = (Lk + 63) >> 6;
    if (CAUSAL) {
        const int lim = ((min(qb0 + 32 * NW - 1, Lq - 1) + p.coff) >> 6) + 1;
        nkt = min(nkt, lim);
    }
    const float c = p.scale * 1.4426950408889634f;
    float m_run = NEG_BIG, l_run = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }

    stage_tile64<NW>(K, p.ldk, 0, Lk, smem, wave, lane);
    stage_tile64<NW>(V, p.ldv, 0, Lk, smem + 8192, wave, lane);
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (!(ABL & 16) || kt == 0) {
            wait_vm0_seen();    // (seen by the compiler: its counted waits for the Q fragment loads otherwise sit inside this loop)
            __syncthreads();
        }
        if (kt + 1 < nkt && !(ABL & 8)) {
            stage_tile64<NW>(K, p.ldk, (kt + 1) * 64, Lk, smem + (buf ^ 1) * 16384, wave, lane);
            stage_tile64<NW>(V, p.ldv, (kt + 1) * 64, Lk, smem + (buf ^ 1) * 16384 + 8192, wave, lane);
        }
        const char* tK = smem + buf * 16384;
        const char* tV = tK + 8192;
        if (DW_ATTN_IDLE_SKIP && qb0 + wave * 32 >= Lq) continue;   // none of this wave's 32 queries exists (the last wave of the
                                                                      // 1500-query tail workgroup): it stages and meets the barriers only
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = (ABL & 32) ? o[kb][r] : 0.f;
            if (!(ABL & 32)) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((ABL & 2) ? qf[kk ^ kb] : frag_rows(tK, kb, kk, lane), qf[kk], s[kb], 0, 0, 0);
            }
        }
        // mask (key tail / causal diagonal) only on the tiles that need it -- a wave-uniform test
        const int wq0 = qb0 + wave * 32 + (CAUSAL ? p.coff : 0);   // last key the wave's first query may see
        const bool need_mask = ((kt + 1) * 64 > Lk) || (CAUSAL && kt * 64 + 63 > wq0);
        if (CAUSAL && kt * 64 > wq0 + 31) continue;  // every key of this tile is in the future of all 32 queries
        float mx = NEG_BIG;
        if (need_mask) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const bool ok = key < Lk && (!CAUSAL || key <= q + p.coff);
                    s[kb][r] = ok ? s[kb][r] : NEG_BIG;
                }
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
        mx = xhalf_max(mx);
        // Deferred maximum: the reference point of the running softmax moves only when some row's tile maximum exceeds it by more
        // than 2^DW_ATTN_DEFER (probabilities then stay below 2^8 -- nothing for fp32 sums or the bf16 P operand, whose relative
        // precision does not depend on the scale); with the exact rule some row of the wave's 32 sees a new maximum in three tiles
        // out of four at 1500 keys, and every such tile pays the 32-register rescale of O.  -DDW_ATTN_DEFER=0 is the exact rule.
        const bool move = __any((mx - m_run) * c > p.defer);
        const float m_new = move ? fmaxf(m_run, mx) : m_run;
        const float mc = m_new * c;
        // two scores per instruction where the ISA has a packed form (v_pk_fma_f32, v_pk_add_f32): the softmax's vector
        // instructions, not the matrix pipe, bound this loop at head_dim 64
        f32x2 rs2[2] = {{0.f, 0.f}, {0.f, 0.f}};
        const f32x2 c2 = {c, c}, mc2 = {mc, mc};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2 t = {s[kb][r], s[kb][r + 1]};
                t = t * c2 - mc2;
                f32x2 pv = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                if (ABL & 1) pv = t;
                s[kb][r] = pv[0];
                s[kb][r + 1] = pv[1];
                rs2[(r >> 1) & 1] += pv;
            }
        rs2[0] += rs2[1];
        float rs = rs2[0][0] + rs2[0][1];
        rs = xhalf_sum(rs);
        if (move) {  // rescale only when the reference moved (exact: alpha == 1 for the rows whose maximum stayed)
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
        }
        l_run += rs;
        m_run = m_new;
        // O^T += V^T . P^T
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pf = pack8(s[kb], s2);
                const int rb = kb * 32 + s2 * 16 + hi * 4;
                if (ABL & 64) {
                    o[kb][s2] += bf2f(pf[0]) + bf2f(pf[2]) + bf2f(pf[4]) + bf2f(pf[6]);
                    continue;
                }
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((ABL & 4) ? qf[db * 2 + s2] : frag_tr(tV, db, rb, rb + 8, lane), pf, o[db], 0, 0, 0);
            }
    }
    const float inv = 1.0f / l_run;
    bf16* O = p.o + q_row0 * p.ldo + h * 64;
    if (DW_ATTN_ROWSTORE && NW <= 7 && (p.ldo & 7) == 0 && ((uintptr_t)p.o & 15) == 0) {
        __syncthreads();                              // every wave is done with the last K/V tile: the patches overlay the tiles
        store_rows(O, p.ldo, qb0 + wave * 32, Lq, smem + wave * 4608, lane, o[0], o[1], inv);
    } else {
        store_t(O, p.ldo, q, q_ok, 0, hi, o[0], inv);
        store_t(O, p.ldo, q, q_ok, 1, hi, o[1], inv);
    }
    if (!VL && p.lse && q_ok && hi == 0) p.lse[((long)b * p.H + h) * Lq + q] = m_run * p.scale + __logf(l_run);
}

// ---------------------------------------------------------------------------------------------------------------
// forward, software-pipelined across key tiles (non-causal shapes; dw_debug_set key 26).  In attn_fwd_kernel a wave runs
// QK(t) -> softmax(t) -> PV(t) strictly in sequence: the softmax's ~100 vector instructions wait for the QK MFMAs to drain and the
// PV MFMAs for the softmax, and only the other waves of the SIMD fill the holes.  Here the scores of tile t+1 are computed one
// iteration ahead (two score register sets, ping-pong by name): inside an iteration the QK MFMAs of tile t+1 and the softmax of
// tile t are independent instruction streams of the SAME wave -- the matrix pipe works on one while the vector unit works on the
// other.  Costs: 32 more registers (three waves per SIMD instead of four) and a THIRD K / V stage in LDS (tile t+1 must be
// resident while tile t is still read by the PV product): 48 KiB per workgroup, three workgroups per CU.
// Same arithmetic per element and the same order of every sum as attn_fwd_kernel: bit-identical output.
// ---------------------------------------------------------------------------------------------------------------
template <int NW = 4, int OCC = 3>
__global__ __launch_bounds__(64 * NW, OCC) void attn_fwd_pipe_kernel(const AttnP p) {
    __shared__ __attribute__((aligned(1024))) char smem[3 * 16384];  // [stage][K tile 8K | V tile 8K]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5, ln = lane & 31;
    int tile, h, b;
    attn_workgroup((p.Lq + 32 * NW - 1) / (32 * NW), p.H, p.plain_order, tile, h, b);
    const int qb0 = tile * (32 * NW);
    const int q = qb0 + wave * 32 + ln;
    const bool q_ok = q < p.Lq;
    const int qc = q_ok ? q : p.Lq - 1;
    const bf16* Q = p.q + (long)b * p.q_rows * p.ldq + h * 64;
    const bf16* K = p.k + (long)b * p.kv_rows * p.ldk + h * 64;
    const bf16* V = p.v + (long)b * p.kv_rows * p.ldv + h * 64;
    bf16x8 qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = ldg8(Q, qc, p.ldq, kk * 16 + hi * 8);
    const int nkt = (p.Lk + 63) >> 6;
    const float c = p.scale * 1.4426950408889634f;
    float m_run = NEG_BIG, l_run = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }

    auto stage = [&](int kt, int st) __attribute__((always_inline)) {
        stage_tile64<NW>(K, p.ldk, kt * 64, p.Lk, smem + st * 16384, wave, lane);
        stage_tile64<NW>(V, p.ldv, kt * 64, p.Lk, smem + st * 16384 + 8192, wave, lane);
    };
    // S^T of key tile kt (stage st), masked on the tail tile
    auto scores = [&](int kt, int st, f32x16 (&s)[2]) __attribute__((always_inline)) {
        const char* tK = smem + st * 16384;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(tK, kb, kk, lane), qf[kk], s[kb], 0, 0, 0);
        }
        if ((kt + 1) * 64 > p.Lk) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    s[kb][r] = key < p.Lk ? s[kb][r] : NEG_BIG;
                }
        }
    };
    // softmax of one tile's scores (in place: s becomes P) + the PV product out of stage st
    auto softmax_pv = [&](int st, f32x16 (&s)[2]) __attribute__((always_inline)) {
        const char* tV = smem + st * 16384 + 8192;
        float mx = NEG_BIG;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
        mx = xhalf_max(mx);
        const bool move = __any((mx - m_run) * c > p.defer);
        const float m_new = move ? fmaxf(m_run, mx) : m_run;
        const float mc = m_new * c;
        f32x2 rs2[2] = {{0.f, 0.f}, {0.f, 0.f}};
        const f32x2 c2 = {c, c}, mc2 = {mc, mc};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2 t = {s[kb][r], s[kb][r + 1]};
                t = t * c2 - mc2;
                const f32x2 pv = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                s[kb][r] = pv[0];
                s[kb][r + 1] = pv[1];
                rs2[(r >> 1) & 1] += pv;
            }
        rs2[0] += rs2[1];
        float rs = rs2[0][0] + rs2[0][1];
        rs = xhalf_sum(rs);
        if (move) {
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
        }
        l_run += rs;
        m_run = m_new;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pf = pack8(s[kb], s2);
                const int rb = kb * 32 + s2 * 16 + hi * 4;
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(tV, db, rb, rb + 8, lane), pf, o[db], 0, 0, 0);
            }
    };

    stage(0, 0);
    if (nkt > 1) stage(1, 1);
    wait_vm0_seen();
    __syncthreads();
    f32x16 sa[2], sb[2];
    scores(0, 0, sa);
    // iteration kt: tile kt's scores are in `cur`; tile kt in stage st0, tile kt+1 (requested an iteration ago) in st1
    auto iter = [&](int kt, int st0, int st1, int st2, f32x16 (&cur)[2], f32x16 (&nxt)[2]) __attribute__((always_inline)) {
        if (kt > 0) {                       // (iteration 0: tiles 0 and 1 were waited for in front of the loop)
            wait_vm0_seen();                // tile kt+1 has landed ...
            __syncthreads();                // ... for every wave, and every wave is done with tile kt-1 (stage st2)
        }
        if (kt + 2 < nkt) stage(kt + 2, st2);
        if (kt + 1 < nkt) scores(kt + 1, st1, nxt);
        softmax_pv(st0, cur);
    };
    int st0 = 0, st1 = 1, st2 = 2;
    for (int kt = 0; kt < nkt; kt += 2) {
        iter(kt, st0, st1, st2, sa, sb);
        if (kt + 1 < nkt) iter(kt + 1, st1, st2, st0, sb, sa);
        const int t0 = st0; st0 = st2; st2 = st1; st1 = t0;      // two tiles on: (st0, st1, st2) -> (st2, st0, st1)
    }
    const float inv = 1.0f / l_run;
    bf16* O = p.o + (long)b * p.q_rows * p.ldo + h * 64;
    if (DW_ATTN_ROWSTORE && (p.ldo & 7) == 0 && ((uintptr_t)p.o & 15) == 0) {
        __syncthreads();
        store_rows(O, p.ldo, qb0 + wave * 32, p.Lq, smem + wave * 4608, lane, o[0], o[1], inv);
    } else {
        store_t(O, p.ldo, q, q_ok, 0, hi, o[0], inv);
        store_t(O, p.ldo, q, q_ok, 1, hi, o[1], inv);
    }
    if (p.lse && q_ok && hi == 0) p.lse[((long)b * p.H + h) * p.Lq + q] = m_run * p.scale + __logf(l_run);
}

// ---------------------------------------------------------------------------------------------------------------
// decode: ONE query per (batch, head) against Lk cached keys/values (cross-attention over the 1500 encoder positions
// and self-attention over the tokens so far, TF:modeling_whisper.py:312-335).  HBM-bound: every K and V byte is read
// exactly once (15.4 MB of cross K/V per sequence and decoder layer pair), nothing is staged or re-read.
// One workgroup per (batch, head), NW waves; 8 lanes share a key (16 bytes = 8 of the 64 head dimensions each), so a
// wave instruction reads 8 complete 128-byte rows.  Pass 1: scores -> LDS + block max.  Pass 2: p = exp2(s - m),
// l += p, o += bf16(p) * v (P is rounded to bf16 before the PV product exactly like the tile kernel does), lane and
// wave partials meet through shuffles and LDS.  All (batch, head) workgroups are resident at once (320 for the
// long-form batch of 16), unlike the 128-query tile kernel which would run 1.25 rounds of mostly idle tiles.
// ---------------------------------------------------------------------------------------------------------------
// PRE: every lane requests ALL its K and V chunks at kernel entry (Lk <= NW * 8 * 12 keys: 12 + 12 16-byte loads per
// lane, 96 registers), so the whole K/V stream of the (batch, head) pair is in flight at once and the two passes run out
// of registers -- the same idea as the weight-streaming GEMV; without it the second pass (V) cannot start its loads
// before the maximum of the first pass is known.
template <int NW, bool PRE>
__global__ __launch_bounds__(64 * NW) void attn_decode_kernel(const AttnP p) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];   // [Lk scores | NW x 64 partial outputs | NW | NW]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = blockIdx.y, b = blockIdx.z;
    const int sub = lane >> 3, ds = (lane & 7) * 8;              // key sub-index within a group of 8, first head dim
    float* sc = dsm;
    float* part = dsm + ((p.Lk + 3) & ~3);
    float* redm = part + NW * 64;
    float* redl = redm + NW;
    const bf16* Q = p.q + (long)b * p.q_rows * p.ldq + h * 64 + ds;
    const bf16* K = p.k + (long)b * p.kv_rows * p.ldk + h * 64 + ds;
    const bf16* V = p.v + (long)b * p.kv_rows * p.ldv + h * 64 + ds;
    const float c = p.scale * 1.4426950408889634f;
    float qv[8];
    {
        const bf16x8 q8 = *(const bf16x8*)Q;
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[e] = bf2f(q8[e]) * c;
    }
    constexpr int NIT = 12;
    bf16x8 kr[PRE ? NIT : 1], vr[PRE ? NIT : 1];
    if constexpr (PRE) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int k = wave * 8 + it * (NW * 8) + sub;
            k = k < p.Lk ? k : p.Lk - 1;
            kr[it] = ld_stream<2>((const bf16x8*)(K + (long)k * p.ldk));
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int k = wave * 8 + it * (NW * 8) + sub;
            k = k < p.Lk ? k : p.Lk - 1;
            vr[it] = ld_stream<2>((const bf16x8*)(V + (long)k * p.ldv));
        }
    }
    // ---- pass 1: scores (in log2 units) and their maximum ----
    float mx = NEG_BIG;
    float sreg[PRE ? NIT : 1];
    if constexpr (PRE) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int k = wave * 8 + it * (NW * 8) + sub;
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) s = fmaf(qv[e], bf2f(kr[it][e]), s);
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 4);
            sreg[it] = s;
            if (k < p.Lk) mx = fmaxf(mx, s);
        }
    } else {
    for (int k0 = wave * 8; k0 < p.Lk; k0 += NW * 8) {
        const int k = k0 + sub;
        float s = 0.f;
        if (k < p.Lk) {
            const bf16x8 k8 = *(const bf16x8*)(K + (long)k * p.ldk);
#pragma unroll
            for (int e = 0; e < 8; ++e) s = fmaf(qv[e], bf2f(k8[e]), s);
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        if (k < p.Lk) {
            if ((lane & 7) == 0) sc[k] = s;
            mx = fmaxf(mx, s);
        }
    }
    }
    mx = wave_max(mx);
    if (lane == 0) redm[wave] = mx;
    __syncthreads();
    mx = redm[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) mx = fmaxf(mx, redm[i]);
    // ---- pass 2: probabilities, normaliser and the weighted sum of V ----
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float l = 0.f;
    if constexpr (PRE) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int k = wave * 8 + it * (NW * 8) + sub;
            if (k < p.Lk) {
                const float pv = __builtin_amdgcn_exp2f(sreg[it] - mx);
                l += pv;
                const float pb = round_bf16(pv);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = fmaf(pb, bf2f(vr[it][e]), o[e]);
            }
        }
    } else {
    for (int k0 = wave * 8; k0 < p.Lk; k0 += NW * 8) {
        const int k = k0 + sub;
        if (k < p.Lk) {
            const float pv = __builtin_amdgcn_exp2f(sc[k] - mx);
            l += pv;
            const float pb = round_bf16(pv);
            const bf16x8 v8 = *(const bf16x8*)(V + (long)k * p.ldv);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fmaf(pb, bf2f(v8[e]), o[e]);
        }
    }
    }
    // the 8 lanes of a key group hold the same l contribution: count it once (lane & 7 == 0), then reduce
    l = (lane & 7) == 0 ? l : 0.f;
    l = wave_sum(l);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        o[e] += __shfl_xor(o[e], 8);
        o[e] += __shfl_xor(o[e], 16);
        o[e] += __shfl_xor(o[e], 32);
    }
    if (sub == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) part[wave * 64 + ds + e] = o[e];
    }
    if (lane == 0) redl[wave] = l;
    __syncthreads();
    if (wave == 0) {
        float acc = 0.f, lt = 0.f;
#pragma unroll
        for (int i = 0; i < NW; ++i) { acc += part[i * 64 + lane]; lt += redl[i]; }
        bf16* O = p.o + (long)b * p.q_rows * p.ldo + h * 64;
        O[lane] = f2bf(acc / lt);
        // natural-log logsumexp of scale * q.k, as the tile kernel reports it
        if (p.lse && lane == 0) p.lse[(long)b * p.H + h] = (mx + __log2f(lt)) * 0.6931471805599453f;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward dQ: stationary = 32 queries per wave (Q, dO fragments + lse, delta in registers); stream K, V tiles
//   S^T = K.Q^T, dP^T = V.dO^T, dS^T = P^T*(dP^T - delta), dQ^T[d][q] += K^T . dS^T
// ---------------------------------------------------------------------------------------------------------------
template <bool CAUSAL, bool FS, int NW = 4>
__global__ __launch_bounds__(64 * NW, 3) void attn_bwd_dq_kernel(const AttnP p) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * 16384];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5, ln = lane & 31;
    int tile, h, b;
    attn_workgroup((p.Lq + 32 * NW - 1) / (32 * NW), p.H, p.plain_order, tile, h, b);
    const int qb0 = tile * (32 * NW);
    const int q = qb0 + wave * 32 + ln;
    const bool q_ok = q < p.Lq;
    const int qc = q_ok ? q : p.Lq - 1;
    const bf16* Q = p.q + (long)b * p.Lq * p.ldq + h * 64;
    const bf16* DO = p.d_o + (long)b * p.Lq * p.lddo + h * 64;
    const bf16* K = p.k + (long)b * p.Lk * p.ldk + h * 64;
    const bf16* V = p.v + (long)b * p.Lk * p.ldv + h * 64;

    bf16x8 qf[4], gf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        qf[kk] = ldg8(Q, qc, p.ldq, kk * 16 + hi * 8);
        gf[kk] = ldg8(DO, qc, p.lddo, kk * 16 + hi * 8);
    }
    const long sidx = ((long)b * p.H + h) * p.Lq + qc;
    const float c = p.scale * 1.4426950408889634f;
    // delta[q] = sum_d dO[q][d] * O[q][d] is computed HERE, from the dO fragments the lane already holds and the matching
    // pieces of its O row (round 5: the separate pre-pass read O and dO once more -- 246 MB and a launch per attention
    // backward, 1.8 ms per step).  The S and dP accumulators start from -lse / scale and -delta of the lane's query, so
    // P = exp2(c * S) and dS = P * dP are packed multiplies with nothing to subtract; the two tables are also written out
    // for the dK/dV kernel, which streams them with its Q / dO tiles and is launched behind this kernel.
    float dsum = 0.f;
    {
        const bf16* Orow = p.o + (long)b * p.Lq * p.ldo + h * 64;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const bf16x8 o8 = ldg8(Orow, qc, p.ldo, kk * 16 + hi * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) dsum += bf2f(o8[e]) * bf2f(gf[kk][e]);
        }
        dsum = xhalf_sum(dsum);
    }
    const float s0 = -p.lse[sidx] / p.scale;
    const float dp0 = -dsum;
    if (q_ok && hi == 0) {
        p.delta[sidx] = dp0;
        p.delta[(long)p.B * p.H * p.Lq + sidx] = s0;
    }

    int nkt = (p.Lk + 63) >> 6;
    if (CAUSAL) nkt = min(nkt, (min(qb0 + 32 * NW - 1, p.Lq - 1) >> 6) + 1);
    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }

    stage_tile64<NW, FS>(K, p.ldk, 0, p.Lk, smem, wave, lane);
    stage_tile64<NW, FS>(V, p.ldv, 0, p.Lk, smem + 8192, wave, lane);
    // The tile loop exists per mask mode (0 none, 1 every pair, 2 decided per tile: the causal kernels) and the non-causal launch
    // runs the full key tiles through mode 0 and the last, partial one through mode 1: tested inside ONE unrolled body the
    // wave-uniform flag became sixteen taken branches per tile (round 5, from the generated code).
    auto tiles = [&](auto modec, int kt_begin, int kt_end) __attribute__((always_inline)) {
    constexpr int MODE = decltype(modec)::value;
    // lane-derived values of the staging and of the mask are re-derived per loop copy from an opaque copy of the lane index: held
    // across both copies they are spilled, and a scratch reload in front of an operand load waits (vmcnt) for the load before it
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int hi_o = lane_o >> 5, q_o = qb0 + wave * 32 + (lane_o & 31);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int buf = kt & 1;
        wait_vm0();
        __syncthreads();
        if (kt + 1 < nkt) {
            int lane_s = lane_o;           // (per iteration: hoisted out of the loop, the staging's lane offsets are what gets spilled)
            asm volatile("" : "+v"(lane_s));
            stage_tile64<NW, FS>(K, p.ldk, (kt + 1) * 64, p.Lk, smem + (buf ^ 1) * 16384, wave, lane_s);
            stage_tile64<NW, FS>(V, p.ldv, (kt + 1) * 64, p.Lk, smem + (buf ^ 1) * 16384 + 8192, wave, lane_s);
        }
        const char* tK = smem + buf * 16384;
        const char* tV = tK + 8192;
        const int wq0 = qb0 + wave * 32;
        const bool need_mask = MODE == 1 || (MODE == 2 && (((kt + 1) * 64 > p.Lk) || (CAUSAL && kt * 64 + 63 > wq0)));
        if (CAUSAL && kt * 64 > wq0 + 31) continue;
        if (DW_ATTN_IDLE_SKIP && wq0 >= p.Lq) continue;               // (see attn_fwd_kernel)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = s0; dp[r] = dp0; }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(tK, kb, kk, lane), qf[kk], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(tV, kb, kk, lane), gf[kk], dp, 0, 0, 0);
            }
            const f32x2 c2 = {c, c};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2 t = {s[r], s[r + 1]};
                t = t * c2;
                f32x2 pv = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                if (need_mask) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int key = kt * 64 + kb * 32 + ((r + e) & 3) + 8 * ((r + e) >> 2) + 4 * hi_o;
                        pv[e] = (key < p.Lk && (!CAUSAL || key <= q_o)) ? pv[e] : 0.f;
                    }
                }
                const f32x2 d2 = {dp[r], dp[r + 1]};
                pv = pv * d2;  // dS^T = P^T * (dP^T - delta)
                s[r] = pv[0];
                s[r + 1] = pv[1];
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 df = pack8(s, s2);
                const int rb = kb * 32 + s2 * 16 + hi * 4;
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(tK, db, rb, rb + 8, lane), df, acc[db], 0, 0, 0);
            }
        }
    }
    };
    if constexpr (CAUSAL) tiles(std::integral_constant<int, 2>{}, 0, nkt);
    else {
        const int nfull = p.Lk >> 6;
        tiles(std::integral_constant<int, 0>{}, 0, nfull);
        tiles(std::integral_constant<int, 1>{}, nfull, nkt);
    }
    bf16* DQ = p.dq + (long)b * p.Lq * p.lddq + h * 64;
    if (DW_ATTN_ROWSTORE && NW <= 7 && (p.lddq & 7) == 0 && ((uintptr_t)p.dq & 15) == 0) {
        __syncthreads();
        store_rows(DQ, p.lddq, qb0 + wave * 32, p.Lq, smem + wave * 4608, lane, acc[0], acc[1], p.scale);
    } else {
        store_t(DQ, p.lddq, q, q_ok, 0, hi, acc[0], p.scale);
        store_t(DQ, p.lddq, q, q_ok, 1, hi, acc[1], p.scale);
    }
    if (p.dq_colsum) {
        colsum_t(p.dq_colsum + ((long)b * p.H + h) * 64, q_ok, 0, hi, ln, acc[0], p.scale);
        colsum_t(p.dq_colsum + ((long)b * p.H + h) * 64, q_ok, 1, hi, ln, acc[1], p.scale);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward dK/dV: stationary = 32 keys per wave (K, V fragments in registers); stream Q, dO tiles (+ lse, delta)
//   S = Q.K^T, dP = dO.V^T  (rows = queries in registers, column = key = lane&31)
//   dV^T[d][key] += dO^T . P,   dK^T[d][key] += Q^T . dS
// ---------------------------------------------------------------------------------------------------------------
template <bool CAUSAL, bool FS, int OCC, int NW = 4>
__global__ __launch_bounds__(64 * NW, OCC) void attn_bwd_dkv_kernel(const AttnP p) {
    // [buf][Q tile 8K | dO tile 8K | lse 64 f32 | delta 64 f32]
    constexpr int STG = 16384 + 512;
    __shared__ __attribute__((aligned(1024))) char smem[2 * 17408];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5, ln = lane & 31;
    int tile, h, b;
    attn_workgroup((p.Lk + 32 * NW - 1) / (32 * NW), p.H, p.plain_order, tile, h, b);
    const int kb0 = tile * (32 * NW);
    const int key = kb0 + wave * 32 + ln;
    const bool k_ok = key < p.Lk;
    const int kc = k_ok ? key : p.Lk - 1;
    const bf16* Q = p.q + (long)b * p.Lq * p.ldq + h * 64;
    const bf16* DO = p.d_o + (long)b * p.Lq * p.lddo + h * 64;
    const bf16* K = p.k + (long)b * p.Lk * p.ldk + h * 64;
    const bf16* V = p.v + (long)b * p.Lk * p.ldv + h * 64;
    const float* DEL = p.delta + ((long)b * p.H + h) * p.Lq;               // -delta
    const float* LSE = DEL + (long)p.B * p.H * p.Lq;                        // -lse / scale

    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        kf[kk] = ldg8(K, kc, p.ldk, kk * 16 + hi * 8);
        vf[kk] = ldg8(V, kc, p.ldv, kk * 16 + hi * 8);
    }
    const float c = p.scale * 1.4426950408889634f;
    const int nqt = (p.Lq + 63) >> 6;
    int qt0 = 0;
    if (CAUSAL) qt0 = kb0 >> 6;  // queries before the first key of this block see none of its keys
    f32x16 av[2], ak[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { av[0][r] = 0.f; av[1][r] = 0.f; ak[0][r] = 0.f; ak[1][r] = 0.f; }

    auto stage = [&](int qt, int buf, int lane) __attribute__((always_inline)) {
        char* base = smem + buf * 17408;
        stage_tile64<NW, FS>(Q, p.ldq, qt * 64, p.Lq, base, wave, lane);
        stage_tile64<NW, FS>(DO, p.lddo, qt * 64, p.Lq, base + 8192, wave, lane);
        if (wave < 2) {  // waves 0,1: 64 lse + 64 delta values through the same async path (4 B per lane; scalar base, 32-bit lane offset)
            int qi = qt * 64 + lane;
            qi = qi < p.Lq ? qi : p.Lq - 1;
            glds4_so(wave == 0 ? LSE : DEL, (unsigned)qi * 4u, lds_addr_of(base + 16384) + (wave == 0 ? 0 : 256));
        }
    };

    if (qt0 < nqt) stage(qt0, 0, lane);
    const int wk0 = kb0 + wave * 32;
    // (one copy of the tile loop per mask mode: see the dQ kernel)
    auto tiles = [&](auto modec, int qt_begin, int qt_end) __attribute__((always_inline)) {
    constexpr int MODE = decltype(modec)::value;
    int lane_o = lane;                     // (see the dQ kernel)
    asm volatile("" : "+v"(lane_o));
    const int hi_o = lane_o >> 5, key_o = kb0 + wave * 32 + (lane_o & 31);
    const bool k_ok_o = key_o < p.Lk;
    for (int qt = qt_begin; qt < qt_end; ++qt) {
        const int buf = (qt - qt0) & 1;
        wait_vm0_seen();    // (seen by the compiler: its counted waits for the K / V fragment loads otherwise sit inside this loop)
        __syncthreads();
        if (qt + 1 < nqt) {
            int lane_s = lane_o;           // (per iteration: see the dQ kernel)
            asm volatile("" : "+v"(lane_s));
            stage(qt + 1, buf ^ 1, lane_s);
        }
        const char* tQ = smem + buf * 17408;
        const char* tG = tQ + 8192;
        const float* tL = (const float*)(tQ + 16384);
        const float* tD = tL + 64;
        // mask needed on the query tail, on a partially valid key block, or on the causal diagonal
        const bool need_mask = MODE == 1 || (MODE == 2 && (((qt + 1) * 64 > p.Lq) || (wk0 + 31 >= p.Lk) || (CAUSAL && wk0 + 31 > qt * 64)));
        if (CAUSAL && wk0 > qt * 64 + 63) continue;  // all 64 queries of this tile precede every key of this wave
        if (DW_ATTN_IDLE_SKIP && wk0 >= p.Lk) continue;               // none of this wave's 32 keys exists
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            // accumulators start from -lse/scale and -delta of their query rows (register r <-> query
            // (r&3) + 8(r>>2) + 4hi of the 32-query block: four consecutive queries per 16-byte LDS read)
            f32x16 s, dp;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ql = qb * 32 + g * 8 + hi * 4;
                const f32x4 l4 = *(const f32x4*)(tL + ql);
                const f32x4 d4 = *(const f32x4*)(tD + ql);
#pragma unroll
                for (int e = 0; e < 4; ++e) { s[g * 4 + e] = l4[e]; dp[g * 4 + e] = d4[e]; }
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(tQ, qb, kk, lane), kf[kk], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(tG, qb, kk, lane), vf[kk], dp, 0, 0, 0);
            }
            f32x16 pr;
            const f32x2 c2 = {c, c};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2 t = {s[r], s[r + 1]};
                t = t * c2;
                f32x2 pv = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};   // P = exp(scale * S - lse)
                if (need_mask) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int qi = qt * 64 + qb * 32 + ((r + e) & 3) + 8 * ((r + e) >> 2) + 4 * hi_o;
                        pv[e] = (qi < p.Lq && k_ok_o && (!CAUSAL || key_o <= qi)) ? pv[e] : 0.f;
                    }
                }
                pr[r] = pv[0];
                pr[r + 1] = pv[1];
                const f32x2 d2 = {dp[r], dp[r + 1]};
                pv = pv * d2;                                          // dS = P * (dP - delta)
                s[r] = pv[0];
                s[r + 1] = pv[1];
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pf = pack8(pr, s2);
                const bf16x8 df = pack8(s, s2);
                const int rb = qb * 32 + s2 * 16 + hi * 4;
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    av[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(tG, db, rb, rb + 8, lane), pf, av[db], 0, 0, 0);
                    ak[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(tQ, db, rb, rb + 8, lane), df, ak[db], 0, 0, 0);
                }
            }
        }
    }
    };
    if constexpr (CAUSAL) tiles(std::integral_constant<int, 2>{}, qt0, nqt);
    else {
        // a workgroup whose key block is not all valid masks every tile; the others only the last, partial query tile.
        // The split is WORKGROUP-uniform (derived from the block's last key, not the wave's): the tile loop holds
        // __syncthreads(), and every wave of a workgroup must reach the same barrier sites -- with a per-wave split the
        // tail wave of the Lk = 1500 block ran the MODE 1 copy while its siblings ran MODE 0 (round-5 advisor finding).
        const int nfull = kb0 + 32 * NW - 1 >= p.Lk ? 0 : (p.Lq >> 6);
        tiles(std::integral_constant<int, 0>{}, 0, nfull);
        tiles(std::integral_constant<int, 1>{}, nfull, nqt);
    }
    bf16* DK = p.dk + (long)b * p.Lk * p.lddk + h * 64;
    bf16* DV = p.dv + (long)b * p.Lk * p.lddv + h * 64;
    if (DW_ATTN_ROWSTORE && NW <= 7 && ((p.lddk | p.lddv) & 7) == 0 && (((uintptr_t)p.dk | (uintptr_t)p.dv) & 15) == 0) {
        __syncthreads();
        store_rows(DK, p.lddk, kb0 + wave * 32, p.Lk, smem + wave * 4608, lane, ak[0], ak[1], p.scale);
        store_rows(DV, p.lddv, kb0 + wave * 32, p.Lk, smem + wave * 4608, lane, av[0], av[1], 1.0f);
    } else {
        store_t(DK, p.lddk, key, k_ok, 0, hi, ak[0], p.scale);
        store_t(DK, p.lddk, key, k_ok, 1, hi, ak[1], p.scale);
        store_t(DV, p.lddv, key, k_ok, 0, hi, av[0], 1.0f);
        store_t(DV, p.lddv, key, k_ok, 1, hi, av[1], 1.0f);
    }
    if (p.dv_colsum) {
        colsum_t(p.dv_colsum + ((long)b * p.H + h) * 64, k_ok, 0, hi, ln, av[0], 1.0f);
        colsum_t(p.dv_colsum + ((long)b * p.H + h) * 64, k_ok, 1, hi, ln, av[1], 1.0f);
    }
}

// bit 0: dq kernel, bit 1: dkv kernel use the 32-bit-offset tile staging; bit 2: dkv kernel compiled for 3 waves per
// SIMD (168 registers; since the accumulators start from the -lse/-delta tables it spills 1-2 registers instead of 14)
int g_attn_bwd_stage = 5;  // (dw_debug_set key 3; 5 measured best: 1.65 vs 1.75 ms per encoder-layer backward)
int g_attn_bwd_waves = 4;  // dw_debug_set key 17: 4 / 12 waves per workgroup of the non-causal backward kernels (12: 384 stationary
                           // rows per workgroup, a third of the tile staging -- measured neutral at the encoder shape, slower at 448 x 1500)
int g_attn_fwd_waves = 4;  // dw_debug_set key 16: 4 / 8 waves (x 32 queries) per workgroup of the non-causal forward kernel (8: half
int g_attn_fwd_pipe = 0;      // dw_debug_set key 26: non-causal forward on attn_fwd_pipe_kernel (scores of tile t+1 under the softmax of tile t)
                           // the K/V staging per query -- measured neutral: the kernel is not bound by the staging traffic)
int g_attn_plain_order = 0; // dw_debug_set key 18
int g_attn_defer = DW_ATTN_DEFER;  // dw_debug_set key 23: the forward's deferred-maximum threshold (0 = exact running maximum; the
                                   // A/B of tests/test_sharp_parity_gpu.py prices this deviation on peaked attention rows)
int g_attn_ablate = 0;     // dw_debug_set key 15 (DW_ABLATE builds)
int g_attn_decode = 1;     // dw_debug_set key 4: 1 = single-query attention runs the streaming decode kernel
static int check_ld(int64_t ld) { return (ld & 7) ? DW_EINVAL : DW_OK; }

extern "C" int dw_attn_fwd_ex(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Lq,
                              int Lk, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t q_batch_rows,
                              int64_t kv_batch_rows, int causal, float scale, void* stream);

extern "C" int dw_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Lq,
                           int Lk, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int causal, float scale,
                           void* stream) {
    if (!lse) return DW_EINVAL;
    return dw_attn_fwd_ex(q, k, v, o, lse, B, H, Lq, Lk, ldq, ldk, ldv, ldo, Lq, Lk, causal, scale, stream);
}

extern "C" int dw_attn_fwd_ex(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Lq,
                              int Lk, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t q_batch_rows,
                              int64_t kv_batch_rows, int causal, float scale, void* stream) {
    DW_CLEAR_ERR();
    if (!q || !k || !v || !o || B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return DW_EINVAL;
    if (q_batch_rows < Lq || kv_batch_rows < Lk) return DW_EINVAL;
    if ((int64_t)Lk * ldk >= (1LL << 31) || (int64_t)Lk * ldv >= (1LL << 31)) return DW_EINVAL;  // 32-bit tile offsets
    if (check_ld(ldq) || check_ld(ldk) || check_ld(ldv) || (ldo & 3)) return DW_EINVAL;
    if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15) || ((uintptr_t)o & 7)) return DW_EINVAL;
    AttnP p = {};
    p.q = (const bf16*)q; p.k = (const bf16*)k; p.v = (const bf16*)v; p.o = (bf16*)o; p.lse = lse;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
    p.B = B; p.H = H; p.Lq = Lq; p.Lk = Lk; p.scale = scale;
    p.q_rows = q_batch_rows; p.kv_rows = kv_batch_rows;
    // causal: 1 = query i sees keys <= i (training); 2 = bottom-right aligned, query i sees keys <= i + (Lk - Lq):
    // several new queries against a longer KV cache (multi-token verify step of speculative decoding)
    if (causal != 0 && causal != 1 && causal != 2) return DW_EINVAL;
    if (causal == 2 && Lk < Lq) return DW_EINVAL;
    p.coff = causal == 2 ? Lk - Lq : 0;
    hipStream_t s = (hipStream_t)stream;
    if (Lq == 1 && g_attn_decode && Lk <= 8192) {
        // one query per (batch, head): the streaming decode kernel (the causal mask is void for a single last query)
        const size_t smem = (((size_t)Lk + 3) & ~(size_t)3) * 4 + 16 * 64 * 4 + 2 * 16 * 4;
        if (Lk >= 512 && Lk <= 16 * 8 * 12 && (g_attn_decode & 2))   // (measured slower: 111 registers -> one workgroup per CU)
            hipLaunchKernelGGL((attn_decode_kernel<16, true>), dim3(1, H, B), dim3(1024), smem, s, p);
        else if (Lk >= 512) hipLaunchKernelGGL((attn_decode_kernel<16, false>), dim3(1, H, B), dim3(1024), smem, s, p);
        else hipLaunchKernelGGL((attn_decode_kernel<4, false>), dim3(1, H, B), dim3(256), smem, s, p);
        DW_CHECK_LAUNCH();
        return DW_OK;
    }
    const int g4 = (Lq + 127) / 128;
    dim3 grid(g4 * H * B), block(256);
    p.plain_order = g_attn_plain_order;
    p.defer = (float)g_attn_defer;
#ifdef DW_ABLATE
    if (!causal && g_attn_ablate) {
        switch (g_attn_ablate) {
#define ABLC(v) case v: hipLaunchKernelGGL((attn_fwd_kernel<false, 4, v>), grid, block, 0, s, p); break;
            ABLC(1) ABLC(2) ABLC(4) ABLC(6) ABLC(8) ABLC(24) ABLC(32) ABLC(64) ABLC(7) ABLC(31) ABLC(30)
#undef ABLC
            default: return DW_EINVAL;
        }
        return DW_OK;
    }
#endif
    // 256 queries per workgroup (8 waves) where that pads no more than 128 would and still fills the chip: the K/V tiles
    // are fetched and written to LDS once per 256 queries (1500 -> 1536 and 448 -> 512 either way)
    const int g8 = (Lq + 255) / 256;
    if (causal) hipLaunchKernelGGL(attn_fwd_kernel<true>, grid, block, 0, s, p);
    else if (g_attn_fwd_pipe == 2 && Lk > 64) hipLaunchKernelGGL((attn_fwd_pipe_kernel<4, 2>), grid, block, 0, s, p);
    else if (g_attn_fwd_pipe && Lk > 64) hipLaunchKernelGGL((attn_fwd_pipe_kernel<4, 3>), grid, block, 0, s, p);
    else if (g_attn_fwd_waves == 8 && g8 * 2 == g4 && (long)g8 * H * B >= 512)
        hipLaunchKernelGGL((attn_fwd_kernel<false, 8>), dim3(g8 * H * B), dim3(512), 0, s, p);
    else hipLaunchKernelGGL(attn_fwd_kernel<false>, grid, block, 0, s, p);
    DW_CHECK_LAUNCH();
    return DW_OK;
}

// Forward over RAGGED batches (packed rows; forward-only passes: no lse).  n_seq sequences; sequence i has q_len[i] <= max_q queries
// at packed rows q_start[i]...; self-attention (kv_start = q_start, kv_len = q_len: keys / values are rows of the same packed
// buffers, causal or not) or cross-attention (kv_start = kv_len = null: K / V are [kv_batches][kv_batch_rows] rectangles of Lk
// valid rows, sequence i reads batch min(i, kv_batches - 1)).  Replaces the scatter -> rectangular attention -> gather the packed
// teacher decoder went through (SURVEY 8 a4 / a5 on the live rows of a batch: TF:modeling_whisper.py:284-356).
extern "C" int dw_attn_fwd_varlen(const void* q, const void* k, const void* v, void* o, int n_seq, int H, int max_q, int Lk,
                                  int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, const int32_t* q_start, const int32_t* q_len,
                                  const int32_t* kv_start, const int32_t* kv_len, int kv_batches, int64_t kv_batch_rows, int causal,
                                  float scale, void* stream) {
    DW_CLEAR_ERR();
    if (!q || !k || !v || !o || !q_start || !q_len || n_seq <= 0 || H <= 0 || max_q <= 0) return DW_EINVAL;
    if ((kv_start == nullptr) != (kv_len == nullptr)) return DW_EINVAL;
    if (!kv_start && (Lk <= 0 || kv_batches <= 0 || kv_batch_rows < Lk || causal)) return DW_EINVAL;
    if (causal != 0 && causal != 1) return DW_EINVAL;
    if (check_ld(ldq) || check_ld(ldk) || check_ld(ldv) || (ldo & 7)) return DW_EINVAL;
    if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15) || ((uintptr_t)o & 15)) return DW_EINVAL;
    const int64_t lk_max = kv_start ? max_q : Lk;
    if (lk_max * ldk >= (1LL << 31) || lk_max * ldv >= (1LL << 31)) return DW_EINVAL;  // 32-bit tile offsets
    AttnP p = {};
    p.q = (const bf16*)q; p.k = (const bf16*)k; p.v = (const bf16*)v; p.o = (bf16*)o; p.lse = nullptr;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
    p.B = n_seq; p.H = H; p.Lq = max_q; p.Lk = kv_start ? max_q : Lk; p.scale = scale;
    p.q_rows = 0; p.kv_rows = kv_batch_rows; p.coff = 0;
    p.q_start = q_start; p.q_len = q_len; p.kv_start = kv_start; p.kv_len = kv_len; p.kv_B = kv_batches;
    p.plain_order = g_attn_plain_order;
    p.defer = (float)g_attn_defer;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(((max_q + 127) / 128) * H * n_seq), block(256);
    if (causal) hipLaunchKernelGGL((attn_fwd_kernel<true, 4, 0, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((attn_fwd_kernel<false, 4, 0, true>), grid, block, 0, s, p);
    DW_CHECK_LAUNCH();
    return DW_OK;
}

extern "C" int dw_attn_bwd_ex(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                              const float* lse, float* delta, void* dq, void* dk, void* dv, int B, int H, int Lq, int Lk,
                              int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo, int64_t lddq,
                              int64_t lddk, int64_t lddv, int causal, float scale, float* dq_colsum, float* dv_colsum,
                              void* stream);
extern "C" int dw_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                           const float* lse, float* delta, void* dq, void* dk, void* dv, int B, int H, int Lq, int Lk,
                           int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo, int64_t lddq,
                           int64_t lddk, int64_t lddv, int causal, float scale, void* stream) {
    return dw_attn_bwd_ex(q, k, v, o, d_o, lse, delta, dq, dk, dv, B, H, Lq, Lk, ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv,
                          causal, scale, nullptr, nullptr, stream);
}

extern "C" int dw_attn_bwd_ex(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                              const float* lse, float* delta, void* dq, void* dk, void* dv, int B, int H, int Lq, int Lk,
                              int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo, int64_t lddq,
                              int64_t lddk, int64_t lddv, int causal, float scale, float* dq_colsum, float* dv_colsum,
                              void* stream) {
    DW_CLEAR_ERR();
    if (!q || !k || !v || !o || !d_o || !lse || !delta || !dq || !dk || !dv) return DW_EINVAL;
    if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return DW_EINVAL;
    if (causal != 0 && causal != 1) return DW_EINVAL;   // (the bottom-right aligned mask is a decoding-only mode)
    if (check_ld(ldq) || check_ld(ldk) || check_ld(ldv) || check_ld(ldo) || check_ld(lddo)) return DW_EINVAL;
    if ((lddq & 3) || (lddk & 3) || (lddv & 3)) return DW_EINVAL;
    if ((int64_t)Lk * ldk >= (1LL << 31) || (int64_t)Lk * ldv >= (1LL << 31) || (int64_t)Lq * ldq >= (1LL << 31) ||
        (int64_t)Lq * lddo >= (1LL << 31))
        return DW_EINVAL;  // 32-bit tile offsets
    if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15) || ((uintptr_t)o & 15) ||
        ((uintptr_t)d_o & 15) || ((uintptr_t)dq & 7) || ((uintptr_t)dk & 7) || ((uintptr_t)dv & 7))
        return DW_EINVAL;
    AttnP p = {};
    p.q = (const bf16*)q; p.k = (const bf16*)k; p.v = (const bf16*)v; p.o = (bf16*)o; p.lse = (float*)lse;
    p.d_o = (const bf16*)d_o; p.delta = delta; p.dq = (bf16*)dq; p.dk = (bf16*)dk; p.dv = (bf16*)dv;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.lddo = lddo; p.lddq = lddq; p.lddk = lddk; p.lddv = lddv;
    p.B = B; p.H = H; p.Lq = Lq; p.Lk = Lk; p.scale = scale;
    p.dq_colsum = dq_colsum; p.dv_colsum = dv_colsum;
    hipStream_t s = (hipStream_t)stream;
    const int q4 = (Lq + 127) / 128, k4 = (Lk + 127) / 128;
    dim3 gq(q4 * H * B), gk(k4 * H * B), block(256);
    p.plain_order = g_attn_plain_order;
    p.defer = (float)g_attn_defer;
    const int fs = g_attn_bwd_stage;
    if (causal) {
        hipLaunchKernelGGL((attn_bwd_dq_kernel<true, false>), gq, block, 0, s, p);
        // (two waves per SIMD: since the tile staging went to scalar bases the causal kernel needs 172 registers -- at 168 it spills 9)
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<true, false, 2>), gk, block, 0, s, p);
    } else {
        // 384 stationary rows (12 waves = the three waves per SIMD the registers allow, as ONE workgroup) where that pads
        // no more than 128 would: the streamed tiles are fetched and written to LDS once per 384 rows (1500 -> 1536)
        const int q12 = (Lq + 383) / 384, k12 = (Lk + 383) / 384;
        const bool dq12 = g_attn_bwd_waves == 12 && q12 * 3 == q4 && (long)q12 * H * B >= 256;
        const bool dkv12 = g_attn_bwd_waves == 12 && k12 * 3 == k4 && (long)k12 * H * B >= 256;
        if (dq12) hipLaunchKernelGGL((attn_bwd_dq_kernel<false, true, 12>), dim3(q12 * H * B), dim3(768), 0, s, p);
        else if (fs & 1) hipLaunchKernelGGL((attn_bwd_dq_kernel<false, true>), gq, block, 0, s, p);
        else hipLaunchKernelGGL((attn_bwd_dq_kernel<false, false>), gq, block, 0, s, p);
        if (dkv12) hipLaunchKernelGGL((attn_bwd_dkv_kernel<false, false, 3, 12>), dim3(k12 * H * B), dim3(768), 0, s, p);
        else if ((fs & 6) == 6) hipLaunchKernelGGL((attn_bwd_dkv_kernel<false, true, 3>), gk, block, 0, s, p);
        else if (fs & 4) hipLaunchKernelGGL((attn_bwd_dkv_kernel<false, false, 3>), gk, block, 0, s, p);
        else if (fs & 2) hipLaunchKernelGGL((attn_bwd_dkv_kernel<false, true, 2>), gk, block, 0, s, p);
        else hipLaunchKernelGGL((attn_bwd_dkv_kernel<false, false, 2>), gk, block, 0, s, p);
    }
    DW_CHECK_LAUNCH();
    return DW_OK;
}
