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

#define SEL_NT 1024

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
    auto allowed = [&](int c) -> bool {
        if (suppress && suppress[c]) return false;
        if (first && begin_suppress && begin_suppress[c]) return false;
        if (no_eos && c == eos) return false;
        if (ts_mode) {
            if (c == tsb - 1) return false;                                  // <|notimestamps|> is never sampled
            if (L >= 1) {
                if (last_ts && pen_ts && c >= tsb) return false;             // after a closed pair: text only
                if (last_ts && !pen_ts && c < eos) return false;             // after text + timestamp: timestamp / EOS
                if (any_ts && c >= tsb && c < ts_last) return false;         // timestamps never decrease
            } else {
                if (c < tsb) return false;                                   // the first sampled token is a timestamp
                if (max_initial >= 0 && c > tsb + max_initial) return false;
            }
        }
        return true;
    };
    // ---- pass 1: best allowed text token and best allowed timestamp token ----
    Best bt = {-INFINITY, 0x7fffffff}, bs = {-INFINITY, 0x7fffffff};
    for (int c0 = tid * 4; c0 < V; c0 += SEL_NT * 4) {
        const bf16x4 x = *(const bf16x4*)(row + c0);                          // (ld is a multiple of 4; pad columns are never used)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = c0 + e;
            if (c < V && allowed(c)) {
                const Best cand = {bf2f(x[e]), c};
                if (c < tsb) bt = better(bt, cand); else bs = better(bs, cand);
            }
        }
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
    const size_t es = d->stream_dtype == DW_F32 ? 4 : 2;
    int rc = dw_embed_fwd(d->ids, d->tok_emb, (const char*)d->pos_emb + (size_t)t * D * es, d->stream_dtype, d->x,
                          d->stream_dtype, B, n, D, stream);
    if (rc != DW_OK) return rc;
    auto gemm = [&](const void* a, long lda, const void* w, const float* bias, int N, int K, void* c, long ldc, int c_dtype,
                    int act, const void* r) -> int {
        DwGemm g = {};
        g.a = a; g.b = w; g.c = c; g.bias = bias; g.r = r;
        g.lda = lda; g.ldb = K; g.ldc = ldc; g.ldr = ldc;
        g.m = rows; g.n = N; g.k = K;
        g.act = act; g.c_dtype = c_dtype; g.r_dtype = c_dtype; g.round_res = 1;
        return dw_gemm_bf16(&g, stream);
    };
    for (int l = 0; l < d->n_layers; ++l) {
        const DwDecoderLayer& L = d->layers[l];
        if (!L.wqkv || !L.wo || !L.wq || !L.wo2 || !L.w1 || !L.w2 || !L.self_kv || !L.cross_kv) return DW_EINVAL;
        // ---- self-attention over the cached prefix ----
        if ((rc = dw_layernorm_fwd(d->x, d->stream_dtype, L.ln1_g, L.ln1_b, d->h, nullptr, nullptr, rows, D, 1e-5f,
                                   stream)) != DW_OK) return rc;
        if ((rc = gemm(d->h, D, L.wqkv, L.bqkv, 3 * D, D, d->qkv, 3 * D, DW_BF16, 0, nullptr)) != DW_OK) return rc;
        {
            const long nvec = (long)rows * ((2 * D) >> 3);
            hipLaunchKernelGGL(kv_append_kernel, dim3((nvec + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                               (const bf16*)d->qkv, (bf16*)L.self_kv, n, t, d->max_len, D, nvec);
            DW_CHECK_LAUNCH();
        }
        const bf16* kc = (const bf16*)L.self_kv;
        if ((rc = dw_attn_fwd_ex(d->qkv, kc, kc + D, d->o, nullptr, B, H, n, t + n, 3 * D, 2 * D, 2 * D, D, n, d->max_len,
                                 n > 1 ? 2 : 0, 0.125f, stream)) != DW_OK) return rc;
        if ((rc = gemm(d->o, D, L.wo, L.bo, D, D, d->x, D, d->stream_dtype, 0, d->x)) != DW_OK) return rc;
        // ---- cross-attention over the static encoder K/V ----
        if ((rc = dw_layernorm_fwd(d->x, d->stream_dtype, L.ln2_g, L.ln2_b, d->h, nullptr, nullptr, rows, D, 1e-5f,
                                   stream)) != DW_OK) return rc;
        if ((rc = gemm(d->h, D, L.wq, L.bq, D, D, d->qkv, 3 * D, DW_BF16, 0, nullptr)) != DW_OK) return rc;
        const bf16* kx = (const bf16*)L.cross_kv;
        if ((rc = dw_attn_fwd_ex(d->qkv, kx, kx + D, d->o, nullptr, B, H, n, d->src_len, 3 * D, 2 * D, 2 * D, D, n,
                                 d->src_len, 0, 0.125f, stream)) != DW_OK) return rc;
        if ((rc = gemm(d->o, D, L.wo2, L.bo2, D, D, d->x, D, d->stream_dtype, 0, d->x)) != DW_OK) return rc;
        // ---- feed-forward ----
        if ((rc = dw_layernorm_fwd(d->x, d->stream_dtype, L.ln3_g, L.ln3_b, d->h, nullptr, nullptr, rows, D, 1e-5f,
                                   stream)) != DW_OK) return rc;
        if ((rc = gemm(d->h, D, L.w1, L.b1, F, D, d->a, F, DW_BF16, 1, nullptr)) != DW_OK) return rc;
        if ((rc = gemm(d->a, F, L.w2, L.b2, D, F, d->x, D, d->stream_dtype, 0, d->x)) != DW_OK) return rc;
    }
    if ((rc = dw_layernorm_fwd(d->x, d->stream_dtype, d->lnf_g, d->lnf_b, d->h, nullptr, nullptr, rows, D, 1e-5f,
                               stream)) != DW_OK) return rc;
    return gemm(d->h, D, d->lm_head, nullptr, d->ldv, D, d->logits, d->ldv, DW_BF16, 0, nullptr);
}
