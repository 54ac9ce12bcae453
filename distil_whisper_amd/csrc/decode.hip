// Token selection of batched greedy decoding: the reference's logits processors + argmax + EOS bookkeeping for one
// decoding step of a whole batch in ONE launch (what GenerationMixin._sample does in a dozen small torch kernels).
//
// Reference behaviour (third-party `transformers`, TF: = transformers/generation/):
//   MinNewTokensLengthLogitsProcessor, SuppressTokensAtBeginLogitsProcessor, SuppressTokensLogitsProcessor and
//   WhisperTimeStampLogitsProcessor (TF:logits_process.py; installed in that order by
//   TF:models/whisper/generation_whisper.py:1774-1812), then `argmax` and
//   `next = next * unfinished + pad * (1 - unfinished)` (TF:utils.py `_sample`), as reached from
//   run_distillation.py:1524-1528, run_eval.py:690-739 and run_pseudo_labelling.py:861-996.
// Every rule is a predicate on (column, row history), so nothing is materialised: one pass over the row of bf16 logits
// keeps the best allowed text token and the best allowed timestamp token, a second pass (timestamp mode only) sums the
// timestamp probability mass for the "timestamps together beat the best text token" rule.
// HBM-bound: B x V x 2 bytes read once (twice in timestamp mode; the row is L2 resident).  One 1024-thread workgroup
// per row; being a plain kernel on the launch stream it is captured into the per-position HIP graphs like the rest of
// the step (the torch implementation of the timestamp rules was not graph-safe).
#include "common.h"
#include "../../include/dwamd.h"

// dw_debug_set key 7 (A/B): bit 0 LayerNorm-on-load off, bit 1 K/V append fusion off, bit 2 / bit 3: the self- / cross-attention of
// the token step as separate projection + attention launches.  Default 4: measured at batch 16 (profiles/r5_decode_fusions.md) the
// cross-attention with its q projection inside saves 7.5 us per token step (2 launches fewer; 38.3 us against 13.6 + 30.8), the
// self-attention with its q / k / v projection inside LOSES 8.6 us -- its 320 workgroups each pull the head's 491 KB of weights
// through L2 (157 MB against the GEMV's 9.8 MB from HBM) for a 5 us attention.
int g_decode_fuse_off = 4;

#define SEL_NT 1024
#define NEG_BIG_D (-1.0e30f)

struct Best { float v; int i; };
__device__ __forceinline__ Best better(Best a, Best b) {      // larger value wins, ties go to the smaller index
    return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ __forceinline__ Best block_best(Best x, Best* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Best y; y.v = __shfl_xor(x.v, o); y.i = __shfl_xor(x.i, o);
        x = better(x, y);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = x;
    __syncthreads();
    Best t = red[0];
    for (int i = 1; i < SEL_NT / 64; ++i) t = better(t, red[i]);
    return t;
}

__global__ __launch_bounds__(SEL_NT) void greedy_select_kernel(
    const bf16* logits, int V, long ld, const uint8_t* suppress, const uint8_t* begin_suppress, int first, int no_eos,
    int forced, int tb, int max_initial, int64_t* tokens, long tok_ld, int n, int begin_index, int eos, int fill,
    uint8_t* done, int64_t* cur) {
    __shared__ Best red[SEL_NT / 64];
    __shared__ float redf[SEL_NT / 64];
    __shared__ int redi[SEL_NT / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    int64_t* row_tok = tokens + (long)b * tok_ld;
    if (forced) {                                  // position n still belongs to the forced prefix (teacher forcing)
        if (tid == 0) cur[b] = row_tok[n];
        return;
    }
    const bf16* row = logits + (long)b * ld;
    const bool ts_mode = tb >= 0;
    const int tsb = ts_mode ? tb : V + 1;          // first timestamp id (beyond the vocabulary when the rules are off)
    // ---- row state of the timestamp rules (WhisperTimeStampLogitsProcessor) ----
    bool last_ts = false, pen_ts = true, any_ts = false;
    int ts_last = 0;
    const int L = n - begin_index;
    if (ts_mode && L >= 1) {
        last_ts = row_tok[n - 1] >= tsb;
        pen_ts = L >= 2 ? row_tok[n - 2] >= tsb : true;
        int pos = 0;                               // 1-based position (within the generated part) of the last timestamp
        for (int i = tid; i < L; i += SEL_NT) pos = row_tok[begin_index + i] >= tsb ? max(pos, i + 1) : pos;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pos = max(pos, __shfl_xor(pos, o));
        if ((tid & 63) == 0) redi[tid >> 6] = pos;
        __syncthreads();
        pos = 0;
        for (int i = 0; i < SEL_NT / 64; ++i) pos = max(pos, redi[i]);
        any_ts = pos > 0;
        if (any_ts) {
            const int last_val = (int)row_tok[begin_index + pos - 1];
            ts_last = (last_ts && !pen_ts) ? last_val : last_val + 1;
        }
    }
    // Every rule is a predicate on the column alone once the row state is known: two allowed id intervals (text /
    // special ids below the first timestamp, timestamp ids), two single banned ids, and the byte masks.
    int tlo = 0, thi = ts_mode ? tsb : V, slo = V, shi = V;       // allowed: [tlo, thi) and [slo, shi)
    const int ban_eos = no_eos ? eos : -1, ban_nots = ts_mode ? tsb - 1 : -1;
    if (ts_mode) {
        if (L >= 1) {
            if (last_ts && pen_ts) { slo = shi = V; }                           // after a closed pair: text only
            else {
                slo = any_ts ? max(tsb, ts_last) : tsb;                          // timestamps never decrease
                if (last_ts) tlo = eos;                                         // after text + timestamp: timestamp / EOS
            }
        } else {
            tlo = thi = 0;                                                      // the first sampled token is a timestamp
            slo = tsb;
            shi = max_initial >= 0 ? min(V, tsb + max_initial + 1) : V;
        }
    }
    auto allowed = [&](int c) -> bool {               // (used by the probability-mass pass below)
        if (suppress && suppress[c]) return false;
        if (first && begin_suppress && begin_suppress[c]) return false;
        return ((c >= tlo && c < thi) || (c >= slo && c < shi)) && c != ban_eos && c != ban_nots;
    };
    // ---- pass 1: best allowed text token and best allowed timestamp token ----
    // (the byte masks are fetched four columns at a time; the next chunk is requested before the current one is judged)
    const bool word_masks = (((uintptr_t)suppress | (uintptr_t)begin_suppress) & 3) == 0;
    Best bt = {-INFINITY, 0x7fffffff}, bs = {-INFINITY, 0x7fffffff};
    auto masks_of = [&](int c0) -> unsigned {          // byte e != 0: column c0 + e is suppressed
        unsigned mask = 0;
        if (word_masks && c0 + 3 < V) {                 // (uniform except in the last chunk of a row)
            if (suppress) mask |= *(const unsigned*)(suppress + c0);
            if (first && begin_suppress) mask |= *(const unsigned*)(begin_suppress + c0);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c0 + e < V && ((suppress && suppress[c0 + e]) || (first && begin_suppress && begin_suppress[c0 + e])))
                    mask |= 0xffu << (8 * e);
        }
        return mask;
    };
    // A row lives on ONE CU (one workgroup), so the kernel is bound by instructions per column, not by bytes: a chunk of
    // four columns that lies inside one allowed interval with no mask bit and no banned id (almost every chunk) takes
    // the short path -- a compare and two selects per column; ascending order within a thread makes the strict compare
    // keep the smallest index among equal values.
    auto judge = [&](int c0, const bf16x4& x, unsigned mask, bool live) {
        const int c3 = c0 + 3;
        const bool in_text = c0 >= tlo && c3 < thi, in_ts = c0 >= slo && c3 < shi;
        const bool clean = live && mask == 0 && c3 < V && (in_text || in_ts) && !(ban_eos >= c0 && ban_eos <= c3) &&
                           !(ban_nots >= c0 && ban_nots <= c3);
        if (clean) {
            float bv = in_text ? bt.v : bs.v;
            int bi = in_text ? bt.i : bs.i;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = bf2f(x[e]);
                if (v > bv) { bv = v; bi = c0 + e; }
            }
            if (in_text) { bt.v = bv; bt.i = bi; } else { bs.v = bv; bs.i = bi; }
            return;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = c0 + e;
            const bool ok = live && c < V && !((mask >> (8 * e)) & 0xffu) &&
                            ((c >= tlo && c < thi) || (c >= slo && c < shi)) && c != ban_eos && c != ban_nots;
            if (ok) {
                const Best cand = {bf2f(x[e]), c};
                if (c < tsb) bt = better(bt, cand); else bs = better(bs, cand);
            }
        }
    };
    constexpr int NPRE = 13;                           // 13 x 4096 columns cover every Whisper vocabulary (51 866)
    if (V <= NPRE * SEL_NT * 4) {
        // The whole row is requested before anything is judged, with clamped addresses instead of branches around the
        // loads: as a loop, a thread's 13 chunks were 13 dependent L2 round trips (10 of the kernel's 19 us).
        const int clast = (V - 1) & ~3;
        bf16x4 xr[NPRE];
        unsigned mr[NPRE];
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int c0 = min(tid * 4 + i * SEL_NT * 4, clast);
            xr[i] = *(const bf16x4*)(row + c0);
            mr[i] = masks_of(c0);
        }
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int c0 = tid * 4 + i * SEL_NT * 4;
            judge(min(c0, clast), xr[i], mr[i], c0 < V);
        }
    } else {
        for (int c0 = tid * 4; c0 < V; c0 += SEL_NT * 4) judge(c0, *(const bf16x4*)(row + c0), masks_of(c0), true);
    }
    bt = block_best(bt, red);
    bs = block_best(bs, red);
    Best pick = better(bt, bs);
    if (ts_mode && bs.v > -INFINITY) {
        // sampled mass rule: if logsumexp over the allowed timestamps exceeds the best text logit, a timestamp is taken
        float sum = 0.f;
        for (int c = tsb + tid; c < V; c += SEL_NT)
            if (allowed(c)) sum += __expf(bf2f(row[c]) - bs.v);
        sum = wave_sum(sum);
        __syncthreads();
        if ((tid & 63) == 0) redf[tid >> 6] = sum;
        __syncthreads();
        sum = 0.f;
        for (int i = 0; i < SEL_NT / 64; ++i) sum += redf[i];
        if (bs.v + __logf(sum) > bt.v) pick = bs;
    }
    if (tid == 0) {
        long nxt = pick.i == 0x7fffffff ? 0 : pick.i;
        if (eos >= 0) {
            if (done[b]) nxt = fill;
            if (nxt == eos) done[b] = 1;
        }
        row_tok[n] = nxt;
        cur[b] = nxt;
    }
}

extern "C" int dw_greedy_select(const void* logits, int B, int V, int64_t ld, const uint8_t* suppress,
                                const uint8_t* begin_suppress, int first, int no_eos, int forced, int ts_begin,
                                int max_initial, int64_t* tokens, int64_t tok_ld, int n, int begin_index, int eos,
                                int fill, uint8_t* done, int64_t* cur, void* stream) {
    DW_CLEAR_ERR();
    if (!tokens || !cur || B <= 0 || n < 1 || n >= tok_ld) return DW_EINVAL;
    if (!forced) {
        if (!logits || V <= 0 || ld < V || (ld & 3) || ((uintptr_t)logits & 7)) return DW_EINVAL;
        if (eos >= 0 && !done) return DW_EINVAL;
        if (ts_begin >= 0 && (eos < 0 || begin_index < 1 || begin_index > n)) return DW_EINVAL;
    }
    hipLaunchKernelGGL(greedy_select_kernel, dim3(B), dim3(SEL_NT), 0, (hipStream_t)stream, (const bf16*)logits, V,
                       (long)ld, suppress, begin_suppress, first, no_eos, forced, ts_begin, max_initial, tokens,
                       (long)tok_ld, n, begin_index, eos, fill, done, cur);
    DW_CHECK_LAUNCH();
    return DW_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// Token step: attention with its projection inside (round 5).  One workgroup of 16 waves per (sequence, head), as in
// attn_decode_kernel (attention.hip); in front of the two passes over the keys the workgroup computes what the projection
// GEMV in front of the attention used to deliver for its head:
//   MODE 0 (cross-attention):  q_h = bf16(Wq[h] . bf16(LayerNorm(x_b)) + bq[h])                       (64 x D of weights)
//   MODE 1 (self-attention):   q_h, k_h, v_h from the fused QKV weight; k_h / v_h are appended to the cache row t of the
//                              sequence and enter the softmax from LDS (the row is not read back from memory)
// The weight rows of a head are shared by the `batch` workgroups of that head through L2; all 10 x 16-byte weight loads of a
// lane are requested at kernel entry and land while the row statistics are reduced.  What it removes is a dependent launch
// per attention (a kernel boundary costs ~5 us in the 20-launch token step, profiles/r4_decode_gap_histogram.md) and the
// q / k / v round trip through memory.  The passes over the keys request four key groups per lane before the first is used
// (64 KiB in flight per workgroup instead of 16: the single-load loop was a chain of HBM round trips).
// Same arithmetic as the separate launches: two-pass LayerNorm statistics, bf16 operands, fp32 sums (in another order than the
// MFMA GEMV's -- a q element may differ by one bf16 ulp), P rounded to bf16 in front of the PV product.
// ---------------------------------------------------------------------------------------------------------------------
struct DecAttnP {
    const void* x; long ldx;                   // residual stream [B][D] (f32 or bf16)
    const float* ln_g; const float* ln_b; float eps;
    const bf16* w; const float* bias;          // MODE 0: Wq [D][D], bq [D];  MODE 1: Wqkv [3D][D], bqkv [3D]
    const bf16* k; const bf16* v;              // keys / values of batch 0, head 0 (row pitch ldkv, batch pitch kv_rows rows)
    long ldkv, kv_rows;
    bf16* kv_app;                              // MODE 1: the cache (K | V per row) -- row t of every sequence is written
    bf16* o; long ldo;
    int D, Lk, t;                              // Lk counts the new key in MODE 1 (= t + 1)
    float scale;
};
template <bool XBF>
__device__ __forceinline__ void dec_ld_x8(const void* x, long off, float (&v)[8]) {
    if (XBF) {
        const bf16x8 t = *(const bf16x8*)((const bf16*)x + off);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = bf2f(t[e]);
    } else {
        const f32x4 a = *(const f32x4*)((const float*)x + off);
        const f32x4 b = *(const f32x4*)((const float*)x + off + 4);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    }
}
constexpr int DEC_NW = 8;                      // waves per workgroup (two workgroups per CU at <= 128 registers)
template <int MODE, bool XBF, int NI>        // NI = D / 64: 64-column steps of a weight row (compile time: the fragments live in registers)
__global__ __launch_bounds__(64 * DEC_NW, 4) void attn_decode_proj_kernel(const DecAttnP p) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    // [Lk scores (x4) | NW x 64 partial outputs | NW | NW | NW (sums) | 64 q | 64 k_new | 64 v_new | D bf16 normalised row | D f32 row]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = blockIdx.y, b = blockIdx.z;
    constexpr int D = NI * 64;
    float* sc = dsm;
    float* part = dsm + ((p.Lk + 3) & ~3);
    float* redm = part + DEC_NW * 64;
    float* redl = redm + DEC_NW;
    float* reds = redl + DEC_NW;
    float* qs = reds + DEC_NW;
    float* knew = qs + 64;
    float* vnew = knew + 64;
    bf16* lnb = (bf16*)(vnew + 64);
    float* xrow = (float*)(lnb + D);
    const float c = p.scale * 1.4426950408889634f;
    // ---- projection ----
    const int j = tid >> 3, l = tid & 7;                  // output j of the head (0..63), 8 lanes share it
    // All of a lane's weight fragments (20 x 16 bytes at D = 1280) are requested at kernel entry and land while the row statistics
    // are reduced.  (With 16 waves and a 64-register budget -- two 1024-thread workgroups per CU -- every split of the fragments
    // spilled, and a spilled fragment is a wait for its load in front of the reductions; eight waves at <= 128 registers keep the
    // same 64 KiB of key / value loads in flight per workgroup.)  The opaque touch of all fragments in dot() makes them live at one
    // point: without it the k / v passes become load -> wait -> use chains (twenty L2 round trips in a row).
    constexpr int NE = NI;
    bf16x8 wf[NI];
    const bf16* const wrow0 = p.w + (long)(h * 64 + j) * D + l * 8;
#pragma unroll
    for (int i = 0; i < NE; ++i) wf[i] = *(const bf16x8*)(wrow0 + i * 64);
    {
        // the row goes through LDS as fp32: nothing of it is held in registers across the two reductions, next to the 80 weight
        // registers of a lane
        if (tid < (D >> 3)) {
            float xv[8];
            dec_ld_x8<XBF>(p.x, (long)b * p.ldx + tid * 8, xv);
            *(f32x4*)(xrow + tid * 8) = f32x4{xv[0], xv[1], xv[2], xv[3]};
            *(f32x4*)(xrow + tid * 8 + 4) = f32x4{xv[4], xv[5], xv[6], xv[7]};
        }
        __syncthreads();
        const bool has = tid < (D >> 2);
        float s1 = 0.f;
        if (has) { const f32x4 a = *(const f32x4*)(xrow + tid * 4); s1 = (a[0] + a[1]) + (a[2] + a[3]); }
        const float mu = block_sum<64 * DEC_NW>(s1, reds) / (float)D;
        float s2 = 0.f;
        if (has) {
            const f32x4 a = *(const f32x4*)(xrow + tid * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float dd = a[e] - mu; s2 += dd * dd; }
        }
        const float rs = rsqrtf(block_sum<64 * DEC_NW>(s2, reds) / (float)D + p.eps);
        f32x4 g4 = {0.f, 0.f, 0.f, 0.f}, b4 = {0.f, 0.f, 0.f, 0.f};
        if (has) {                                        // (requested in front of the late fragments: the counter retires in order)
            g4 = *(const f32x4*)(p.ln_g + tid * 4);
            b4 = *(const f32x4*)(p.ln_b + tid * 4);
        }
        asm volatile("" : "+v"(g4), "+v"(b4));
#pragma unroll
        for (int i = NE; i < NI; ++i) wf[i] = *(const bf16x8*)(wrow0 + i * 64);
        if (has) {
            const f32x4 a = *(const f32x4*)(xrow + tid * 4);
            bf16x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = f2bf((a[e] - mu) * rs * g4[e] + b4[e]);
            *(bf16x4*)(lnb + tid * 4) = y;
        }
        __syncthreads();
    }
    auto dot = [&](bf16x8 (&wf)[NI]) __attribute__((always_inline)) -> float {
#pragma unroll
        for (int i = 0; i < NI; ++i) asm volatile("" : "+v"(wf[i]));
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const bf16x8 a = *(const bf16x8*)(lnb + (i * 8 + l) * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = fmaf(bf2f(wf[i][e]), bf2f(a[e]), acc);
        }
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        acc += __shfl_xor(acc, 4);
        return acc;
    };
    {
        const float qa = dot(wf);
        if (l == 0) qs[j] = bf2f(f2bf(qa + p.bias[h * 64 + j])) * c;
    }
    if constexpr (MODE == 1) {
#pragma unroll
        for (int part_i = 1; part_i <= 2; ++part_i) {       // 1: k, 2: v
            const bf16* wrow = p.w + ((long)part_i * D + h * 64 + j) * D + l * 8;
#pragma unroll
            for (int i = 0; i < NI; ++i) wf[i] = *(const bf16x8*)(wrow + i * 64);
            const float a = dot(wf);
            if (l == 0) {
                const bf16 r = f2bf(a + p.bias[part_i * D + h * 64 + j]);
                (part_i == 1 ? knew : vnew)[j] = bf2f(r);
                p.kv_app[((long)b * p.kv_rows + p.t) * p.ldkv + (part_i - 1) * D + h * 64 + j] = r;
            }
        }
    }
    __syncthreads();
    // ---- attention over Lk keys: 8 lanes share a key (16 bytes = 8 head dimensions each) ----
    const int sub = lane >> 3, ds = (lane & 7) * 8;
    const bf16* K = p.k + (long)b * p.kv_rows * p.ldkv + h * 64 + ds;
    const bf16* V = p.v + (long)b * p.kv_rows * p.ldkv + h * 64 + ds;
    float qv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[e] = qs[ds + e];
    const int nold = MODE == 1 ? p.Lk - 1 : p.Lk;        // keys that live in memory
    constexpr int UN = 8;                                // key groups requested ahead per lane
    float mx = NEG_BIG_D;
    for (int k0 = wave * 8; k0 < p.Lk; k0 += UN * DEC_NW * 8) {
        bf16x8 kr[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            int k = k0 + u * DEC_NW * 8 + sub;
            k = k < nold ? k : (nold > 0 ? nold - 1 : 0);
            kr[u] = nold > 0 ? ld_stream<2>((const bf16x8*)(K + (long)k * p.ldkv)) : bf16x8{};
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int k = k0 + u * DEC_NW * 8 + sub;
            float s = 0.f;
            if (MODE == 1 && k == nold) {
#pragma unroll
                for (int e = 0; e < 8; ++e) s = fmaf(qv[e], knew[ds + e], s);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) s = fmaf(qv[e], bf2f(kr[u][e]), s);
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
    for (int i = 1; i < DEC_NW; ++i) mx = fmaxf(mx, redm[i]);
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float lsum = 0.f;
    for (int k0 = wave * 8; k0 < p.Lk; k0 += UN * DEC_NW * 8) {
        bf16x8 vr[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            int k = k0 + u * DEC_NW * 8 + sub;
            k = k < nold ? k : (nold > 0 ? nold - 1 : 0);
            vr[u] = nold > 0 ? ld_stream<2>((const bf16x8*)(V + (long)k * p.ldkv)) : bf16x8{};
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int k = k0 + u * DEC_NW * 8 + sub;
            if (k < p.Lk) {
                const float pv = __builtin_amdgcn_exp2f(sc[k] - mx);
                lsum += pv;
                const float pb = round_bf16(pv);
                if (MODE == 1 && k == nold) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = fmaf(pb, vnew[ds + e], o[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = fmaf(pb, bf2f(vr[u][e]), o[e]);
                }
            }
        }
    }
    lsum = (lane & 7) == 0 ? lsum : 0.f;
    lsum = wave_sum(lsum);
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
    if (lane == 0) redl[wave] = lsum;
    __syncthreads();
    if (wave == 0) {
        float acc = 0.f, lt = 0.f;
#pragma unroll
        for (int i = 0; i < DEC_NW; ++i) { acc += part[i * 64 + lane]; lt += redl[i]; }
        p.o[(long)b * p.ldo + h * 64 + lane] = f2bf(acc / lt);
    }
}

template <int MODE, bool XBF>
static int launch_decode_proj_ni(const DecAttnP& p, const dim3& grid, const dim3& block, size_t smem, hipStream_t s) {
    switch (p.D >> 6) {
#define DW_NI(n) case n: hipLaunchKernelGGL((attn_decode_proj_kernel<MODE, XBF, n>), grid, block, smem, s, p); break;
        DW_NI(6) DW_NI(8) DW_NI(12) DW_NI(16) DW_NI(20)        // d_model 384 / 512 / 768 / 1024 / 1280
#undef DW_NI
        default: return DW_EINVAL;
    }
    DW_CHECK_LAUNCH();
    return DW_OK;
}
static bool decode_proj_dim_ok(int D) { return D == 384 || D == 512 || D == 768 || D == 1024 || D == 1280; }
static int launch_decode_proj(int mode, int x_dtype, const DecAttnP& p, int B, int H, hipStream_t s) {
    const size_t smem = ((((size_t)p.Lk + 3) & ~(size_t)3) + DEC_NW * 64 + 3 * DEC_NW + 3 * 64) * 4 + (size_t)p.D * 6;
    const dim3 grid(1, H, B), block(64 * DEC_NW);
    const bool xbf = x_dtype == DW_BF16;
    if (mode == 0 && xbf) return launch_decode_proj_ni<0, true>(p, grid, block, smem, s);
    if (mode == 0) return launch_decode_proj_ni<0, false>(p, grid, block, smem, s);
    if (xbf) return launch_decode_proj_ni<1, true>(p, grid, block, smem, s);
    return launch_decode_proj_ni<1, false>(p, grid, block, smem, s);
}

// ---------------------------------------------------------------------------------------------------------------------
// One decoder pass of cached greedy decoding as ONE C call: every launch of `WhisperDecoder.forward` on the cache
// branch (TF:modeling_whisper.py:690-795 with 312-335; reached from `generate`, run_eval.py:739,
// run_distillation.py:1524-1528, run_pseudo_labelling.py:861-996) is enqueued on the caller's stream: embedding,
// per layer LayerNorm -> fused QKV GEMM -> K/V appended to the cache in place -> self-attention over the cached prefix
// -> out-proj + residual -> LayerNorm -> Q GEMM -> cross-attention over the static encoder K/V -> out-proj + residual
// -> LayerNorm -> FC1 + GELU -> FC2 + residual, then the final LayerNorm and the tied LM head.  n_new = 1 is the
// token step (skinny weight-streaming GEMMs, streaming single-query attention); n_new > 1 scores several new
// positions against the cache with the bottom-right aligned causal mask (prompt prefill, the verify pass of assisted
// decoding).  Nothing is allocated; no host synchronisation: the call can be captured into a HIP graph.

__global__ __launch_bounds__(256) void kv_append_kernel(const bf16* qkv, bf16* cache, int n_new, int t, int max_len,
                                                         int D, long nvec) {
    // cache[b][t + j][0 .. 2D) = qkv[b * n_new + j][D .. 3D)   (8 bf16 per thread)
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nvec) return;
    const int per_row = (2 * D) >> 3;
    const long row = i / per_row;
    const int c = (int)(i - row * per_row) << 3;
    const long b = row / n_new;
    const int j = (int)(row - b * n_new);
    *(bf16x8*)(cache + ((b * max_len + t + j) * 2L * D) + c) = *(const bf16x8*)(qkv + row * 3L * D + D + c);
}

extern "C" int dw_decode_step(const DwDecodeStep* d, void* stream) {
    DW_CLEAR_ERR();
    if (!d || !d->ids || !d->tok_emb || !d->pos_emb || !d->layers || !d->lm_head || !d->x || !d->h || !d->qkv || !d->o ||
        !d->a || !d->logits)
        return DW_EINVAL;
    const int B = d->batch, n = d->n_new, D = d->d_model, H = d->heads, F = d->ffn, t = d->t;
    if (B <= 0 || n <= 0 || D != H * 64 || (D & 63) || (F & 63) || d->n_layers <= 0 || d->src_len <= 0 || t < 0 ||
        t + n > d->max_len || d->ldv < d->vocab || (d->ldv & 15))
        return DW_EINVAL;
    if (d->stream_dtype != DW_F32 && d->stream_dtype != DW_BF16) return DW_EINVAL;
    const int rows = B * n;
    const long ldx = d->cross_kv_ld > 0 ? d->cross_kv_ld : 2L * D;        // row pitch of the cross-attention K | V
    if (ldx < 2L * D || (ldx & 7)) return DW_EINVAL;
    const size_t es = d->stream_dtype == DW_F32 ? 4 : 2;
    int rc = dw_embed_fwd(d->ids, d->tok_emb, (const char*)d->pos_emb + (size_t)t * D * es, d->stream_dtype, d->x,
                          d->stream_dtype, B, n, D, stream);
    if (rc != DW_OK) return rc;
    // With few rows the LayerNorm in front of a projection is computed inside the weight-streaming GEMM and the new
    // K/V go straight into the cache (DwGemm.ln_x / kv_out): 9 launches per layer instead of 13.
    const bool fuse_ln = rows <= 32 && D <= 1280 && !(g_decode_fuse_off & 1), fuse_kv = rows <= 64 && !(g_decode_fuse_off & 2);
    // token step: the q (q / k / v) projection of a head is computed by the attention workgroup of that head
    const bool proj_ok = n == 1 && decode_proj_dim_ok(D);
    const bool proj_self = proj_ok && !(g_decode_fuse_off & 4) && t + 1 <= 8192;
    const bool proj_cross = proj_ok && !(g_decode_fuse_off & 8) && d->src_len <= 8192;
    auto gemm = [&](const void* a, long lda, const void* w, const float* bias, int N, int K, void* c, long ldc, int c_dtype,
                    int act, const void* r, const float* ln_g, const float* ln_b, void* kv) -> int {
        DwGemm g = {};
        g.a = a; g.b = w; g.c = c; g.bias = bias; g.r = r;
        g.lda = lda; g.ldb = K; g.ldc = ldc; g.ldr = ldc;
        g.m = rows; g.n = N; g.k = K;
        g.act = act; g.c_dtype = c_dtype; g.r_dtype = c_dtype; g.round_res = 1;
        if (ln_g) {                                   // a = the residual stream x, normalised on load
            g.a = nullptr; g.ln_x = a; g.ld_lnx = lda; g.ln_x_dtype = d->stream_dtype;
            g.ln_gamma = ln_g; g.ln_beta = ln_b; g.ln_eps = 1e-5f;
        }
        if (kv) {
            g.kv_out = kv; g.kv_ld = 2 * D; g.kv_split = D; g.kv_rows_per_batch = n; g.kv_batch_pitch = d->max_len;
            g.kv_row0 = t;
        }
        return dw_gemm_bf16(&g, stream);
    };
    auto ln = [&](const float* gam, const float* bet) -> int {
        return dw_layernorm_fwd(d->x, d->stream_dtype, gam, bet, d->h, nullptr, nullptr, rows, D, 1e-5f, stream);
    };
    for (int l = 0; l < d->n_layers; ++l) {
        const DwDecoderLayer& L = d->layers[l];
        if (!L.wqkv || !L.wo || !L.wq || !L.wo2 || !L.w1 || !L.w2 || !L.self_kv || !L.cross_kv) return DW_EINVAL;
        // ---- self-attention over the cached prefix ----
        if (proj_self && L.bqkv && L.ln1_g && L.ln1_b) {
            DecAttnP q = {};
            q.x = d->x; q.ldx = D; q.ln_g = L.ln1_g; q.ln_b = L.ln1_b; q.eps = 1e-5f;
            q.w = (const bf16*)L.wqkv; q.bias = L.bqkv;
            q.k = (const bf16*)L.self_kv; q.v = q.k + D; q.ldkv = 2 * D; q.kv_rows = d->max_len;
            q.kv_app = (bf16*)L.self_kv; q.o = (bf16*)d->o; q.ldo = D; q.D = D; q.Lk = t + 1; q.t = t; q.scale = 0.125f;
            if ((rc = launch_decode_proj(1, d->stream_dtype, q, B, H, (hipStream_t)stream)) != DW_OK) return rc;
        } else {
        if (fuse_ln) {
            rc = gemm(d->x, D, L.wqkv, L.bqkv, 3 * D, D, d->qkv, 3 * D, DW_BF16, 0, nullptr, L.ln1_g, L.ln1_b,
                      fuse_kv ? L.self_kv : nullptr);
        } else {
            if ((rc = ln(L.ln1_g, L.ln1_b)) != DW_OK) return rc;
            rc = gemm(d->h, D, L.wqkv, L.bqkv, 3 * D, D, d->qkv, 3 * D, DW_BF16, 0, nullptr, nullptr, nullptr,
                      fuse_kv ? L.self_kv : nullptr);
        }
        if (rc != DW_OK) return rc;
        if (!fuse_kv) {
            const long nvec = (long)rows * ((2 * D) >> 3);
            hipLaunchKernelGGL(kv_append_kernel, dim3((nvec + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                               (const bf16*)d->qkv, (bf16*)L.self_kv, n, t, d->max_len, D, nvec);
            DW_CHECK_LAUNCH();
        }
        const bf16* kc = (const bf16*)L.self_kv;
        if ((rc = dw_attn_fwd_ex(d->qkv, kc, kc + D, d->o, nullptr, B, H, n, t + n, 3 * D, 2 * D, 2 * D, D, n, d->max_len,
                                 n > 1 ? 2 : 0, 0.125f, stream)) != DW_OK) return rc;
        }
        if ((rc = gemm(d->o, D, L.wo, L.bo, D, D, d->x, D, d->stream_dtype, 0, d->x, nullptr, nullptr, nullptr)) != DW_OK)
            return rc;
        // ---- cross-attention over the static encoder K/V ----
        if (proj_cross && L.bq && L.ln2_g && L.ln2_b) {
            DecAttnP q = {};
            q.x = d->x; q.ldx = D; q.ln_g = L.ln2_g; q.ln_b = L.ln2_b; q.eps = 1e-5f;
            q.w = (const bf16*)L.wq; q.bias = L.bq;
            q.k = (const bf16*)L.cross_kv; q.v = q.k + D; q.ldkv = ldx; q.kv_rows = d->src_len;
            q.o = (bf16*)d->o; q.ldo = D; q.D = D; q.Lk = d->src_len; q.t = 0; q.scale = 0.125f;
            if ((rc = launch_decode_proj(0, d->stream_dtype, q, B, H, (hipStream_t)stream)) != DW_OK) return rc;
        } else {
        if (fuse_ln) {
            rc = gemm(d->x, D, L.wq, L.bq, D, D, d->qkv, 3 * D, DW_BF16, 0, nullptr, L.ln2_g, L.ln2_b, nullptr);
        } else {
            if ((rc = ln(L.ln2_g, L.ln2_b)) != DW_OK) return rc;
            rc = gemm(d->h, D, L.wq, L.bq, D, D, d->qkv, 3 * D, DW_BF16, 0, nullptr, nullptr, nullptr, nullptr);
        }
        if (rc != DW_OK) return rc;
        const bf16* kx = (const bf16*)L.cross_kv;
        if ((rc = dw_attn_fwd_ex(d->qkv, kx, kx + D, d->o, nullptr, B, H, n, d->src_len, 3 * D, ldx, ldx, D, n,
                                 d->src_len, 0, 0.125f, stream)) != DW_OK) return rc;
        }
        if ((rc = gemm(d->o, D, L.wo2, L.bo2, D, D, d->x, D, d->stream_dtype, 0, d->x, nullptr, nullptr, nullptr)) != DW_OK)
            return rc;
        // ---- feed-forward ----
        if (fuse_ln) {
            rc = gemm(d->x, D, L.w1, L.b1, F, D, d->a, F, DW_BF16, 1, nullptr, L.ln3_g, L.ln3_b, nullptr);
        } else {
            if ((rc = ln(L.ln3_g, L.ln3_b)) != DW_OK) return rc;
            rc = gemm(d->h, D, L.w1, L.b1, F, D, d->a, F, DW_BF16, 1, nullptr, nullptr, nullptr, nullptr);
        }
        if (rc != DW_OK) return rc;
        if ((rc = gemm(d->a, F, L.w2, L.b2, D, F, d->x, D, d->stream_dtype, 0, d->x, nullptr, nullptr, nullptr)) != DW_OK)
            return rc;
    }
    if ((rc = dw_layernorm_fwd(d->x, d->stream_dtype, d->lnf_g, d->lnf_b, d->h, nullptr, nullptr, rows, D, 1e-5f,
                               stream)) != DW_OK) return rc;
    return gemm(d->h, D, d->lm_head, nullptr, d->ldv, D, d->logits, d->ldv, DW_BF16, 0, nullptr, nullptr, nullptr, nullptr);
}
