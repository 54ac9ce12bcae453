/*
 * dwamd.h -- C ABI of libdwamd.so: the MI355X (gfx950 / CDNA4) kernels of the Whisper distillation hot path.
 *
 * The reference (huggingface/distil-whisper) has no FFI layer of its own: its hot path calls the Python surface
 * of `transformers` (WhisperFeatureExtractor + WhisperForConditionalGeneration) which dispatches to ATen/rocBLAS/
 * MIOpen kernels.  Each entry point below names the reference site it replaces (paths relative to the reference
 * tree `training/`, `TF:` = transformers/models/whisper of the pinned transformers==5.15.0).
 *
 * Conventions
 *   - every function returns 0 on success, a hipError_t value (>0) if a launch failed, or a negative DW_E* code
 *     for an invalid argument; nothing throws across this boundary;
 *   - the caller owns every buffer (device pointers from its own allocator); kernels never allocate;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued asynchronously on it, no host sync inside;
 *   - bf16 = 16-bit brain float (the upper half of an IEEE binary32), row-major everywhere, "ld" = leading
 *     dimension in ELEMENTS;
 *   - re-entrant: may be called from the Python main thread (forward) and the autograd engine thread (backward).
 */
#ifndef DWAMD_H
#define DWAMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DW_OK 0
#define DW_EINVAL (-1)  /* bad argument (alignment, K %% 64, null pointer ...) */
#define DW_EUNSUP (-2)  /* unsupported shape (e.g. head_dim != 64)            */

#define DW_F32 0
#define DW_BF16 1

int dw_version(void);

/* ---- a1: log-mel front end --------------------------------------------------------------------------------------
 * Replaces WhisperFeatureExtractor._torch_extract_fbank_features (TF:feature_extraction_whisper.py:135-168), called
 * from run_distillation.py:1176,1234 and run_eval.py:628-642.
 * audio [batch][n_samples] f32 (n_samples = 480000), hann window 400 (periodic), hop 160, reflect-pad centre STFT,
 * power spectrum, mel_filters [201][n_mels] f32 (HF layout), log10(clamp 1e-10), max(x, clipmax-8), (x+4)/4.
 * out [batch][n_mels][n_frames] f32 with n_frames = n_samples/160.  twiddle [400][2] f32 = (cos, sin)(2*pi*i/400),
 * window [400] f32; clipmax [batch] f32 scratch (overwritten). */
int dw_logmel(const float* audio, int batch, int n_samples, const float* mel_filters, int n_mels,
              const float* twiddle, const float* window, float* out, float* clipmax, void* stream);

/* ---- a3/a4/a5/a6: dense projections on the bf16 MFMA ------------------------------------------------------------
 * Replaces nn.Linear / nn.Conv1d (after im2col) forward+backward GEMMs (TF:modeling_whisper.py:279-282, 375-376,
 * 444-445, 566-567, 970).   C[M,N] = epilogue( op(A)[M,K] . op(B)[K,N] ), fp32 accumulation.
 *   trans_a = 0: A stored [M][lda] (K contiguous)      trans_a = 1: A stored [K][lda] (M contiguous)
 *   trans_b = 0: B stored [N][ldb] (K contiguous, i.e. the nn.Linear weight layout)
 *   trans_b = 1: B stored [K][ldb] (N contiguous)
 * Requirements: K %% 64 == 0 (pad with zeros), lda/ldb multiples of 8, base pointers 16-byte aligned.
 * Epilogue, in this order (each step optional):
 *   v = acc + bias[n]                      (bias f32 [N])
 *   Z[m][n] = bf16(v)                      (z_out, ldz; the pre-activation kept for backward)
 *   v = gelu(round_bf16(v))                (act == 1; exact erf GELU on the bf16-rounded value, as autocast does)
 *   v = v * gelu'(Zin[m][n])               (zgrad_in bf16, ldzg: fused GELU backward)
 *   v = (round_res ? round_bf16(v) : v) + R[m %% r_row_mod or m][n]     (residual R, f32 or bf16, ldr)
 *   C[m][n] = v as f32 or bf16             (c_dtype, ldc)
 * R may alias C (in-place accumulation: every element is read and written by the same lane). */
typedef struct DwGemm {
    const void* a;      /* bf16 */
    const void* b;      /* bf16 */
    void* c;            /* f32 or bf16 */
    const float* bias;  /* f32 [N] or NULL */
    void* z_out;        /* bf16 or NULL */
    const void* zgrad_in; /* bf16 or NULL */
    const void* r;      /* residual or NULL */
    int64_t lda, ldb, ldc, ldz, ldzg, ldr;
    int32_t m, n, k;
    int32_t trans_a, trans_b;
    int32_t act;        /* 0 none, 1 gelu */
    int32_t c_dtype;    /* DW_F32 / DW_BF16 */
    int32_t r_dtype;    /* DW_F32 / DW_BF16 */
    int32_t r_row_mod;  /* 0: R row = m; >0: R row = m %% r_row_mod (positional table broadcast) */
    int32_t round_res;  /* 1: round v to bf16 before adding R (autocast semantics) */
    int32_t tile;       /* 0 auto, 128 or 256: force the block tile; 16: the skinny-M (m <= 64) weight-streaming kernel;
                           129 (probing): like 0 for outputs below two rounds of 256-row tiles, refused for k-major A / split K */
    int32_t split_k;    /* > 1: the K range is cut into that many slices, combined either by atomic_acc or by
                           storing fp32 partials at c + slice * slice_stride (then call dw_reduce_slices) */
    int32_t atomic_acc; /* 1: C (f32, plain epilogue) += result with float atomics (gradient accumulation) */
    int64_t slice_stride; /* elements between the partial outputs of consecutive K slices (split_k > 1, no atomics) */
    /* Decode-step fusions, honoured by the skinny-M kernel only (m <= 32 for ln_x, m <= 64 for kv_out; any other launch
     * with these fields set is rejected with DW_EINVAL).  All zero = off. */
    const void* ln_x;        /* A = bf16(LayerNorm(ln_x)) computed on load over the K columns (`a` is ignored; k <= 1280):
                                the LayerNorm in front of a projection (TF:modeling_whisper.py:459, 474, 491) */
    const float* ln_gamma;   /* f32 [k] */
    const float* ln_beta;    /* f32 [k] */
    void* kv_out;            /* bf16 K/V cache: output columns >= kv_split of row m are stored at
                                kv_out[((m / kv_rows_per_batch) * kv_batch_pitch + kv_row0 + m % kv_rows_per_batch) * kv_ld
                                       + (n - kv_split)] instead of C (the in-place cache append of
                                TF:modeling_whisper.py:312-335); requires a bf16 output without activation / residual */
    int64_t ld_lnx, kv_ld;
    int32_t ln_x_dtype;      /* DW_F32 / DW_BF16 */
    int32_t kv_split, kv_rows_per_batch, kv_batch_pitch, kv_row0;
    float ln_eps;
    int32_t z_is_gelu_grad;  /* 1: z_out (with act = 1) receives gelu'(z) in fp16 instead of z in bf16, and zgrad_in is read as
                                such: the backward epilogue multiplies by the stored derivative instead of evaluating it
                                (same bytes; the derivative is rounded to 11 bits where the reference keeps fp32) */
    float* colsum_out;       /* f32 [N] or NULL: the column sums of the stored C (as rounded to its dtype) are ADDED to it with
                                float atomics -- the bias gradient of the Linear whose output gradient this GEMM produces
                                (dX of fc2 -> fc1.bias), without a separate pass over C.  Tile kernels only, no K slices. */
} DwGemm;
int dw_gemm_bf16(const DwGemm* g, void* stream);
/* out[i] (+)= sum over slices of part[s*stride + i]; n, stride multiples of 4 (split-K combination, deterministic). */
int dw_reduce_slices(const float* part, int64_t stride, int slices, float* out, int64_t n, int accumulate, void* stream);
/* The same over [rows][cols] matrices whose rows are ld_part (slabs, slice_stride elements apart) / ld_out elements apart. */
int dw_reduce_slices_ld(const float* part, int64_t slice_stride, int64_t ld_part, int slices, float* out, int64_t ld_out,
                        int rows, int cols, int accumulate, void* stream);

/* ---- LayerNorm (TF:modeling_whisper.py:371,377,434,443,446,573,682), eps 1e-5, statistics in f32 ---------------
 * x [rows][cols] f32 or bf16 (x_dtype), y bf16 [rows][cols]; mean/rstd f32 [rows] (may be NULL for inference). */
int dw_layernorm_fwd(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, float* mean,
                     float* rstd, int rows, int cols, float eps, void* stream);
/* dx = LN'(dy); if accumulate: dres[rows][cols] (f32) += dx else dres = dx.  dgamma/dbeta f32 [cols] are ADDED to
 * (atomics; zero them first).  Optional fused outputs: dres_lowp (bf16 [rows][cols]) = bf16(dres) -- the gradient the
 * next residual branch's GEMMs consume -- and dres_colsum (f32 [cols], ADDED to) = its column sums = the bias gradient
 * of that branch's output projection. */
int dw_layernorm_bwd(const void* dy_bf16, const void* x, int x_dtype, const float* mean, const float* rstd,
                     const float* gamma, float* dres, int accumulate, float* dgamma, float* dbeta, void* dres_lowp,
                     float* dres_colsum, int rows, int cols, void* stream);
/* The same two with row pitches in elements (>= cols, multiples of 4; 8 for bf16 x): activation buffers whose rows are padded
 * by 128 bytes so that the 128-byte pieces a GEMM tile reads from 256-320 consecutive rows do not fall on two of an XCD's sixteen
 * L2 channels (rows 2 560 / 10 240 bytes apart do; engine.row_pad, tools/gemm_stride_probe.py). */
int dw_layernorm_fwd_ld(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, float* mean,
                        float* rstd, int rows, int cols, float eps, int64_t ldx, int64_t ldy, void* stream);
int dw_layernorm_bwd_ld(const void* dy_bf16, const void* x, int x_dtype, const float* mean, const float* rstd,
                        const float* gamma, float* dres, int accumulate, float* dgamma, float* dbeta, void* dres_lowp,
                        float* dres_colsum, int rows, int cols, int64_t lddy, int64_t ldx, int64_t lddres,
                        int64_t ldlowp, void* stream);

/* ---- attention core (TF:modeling_whisper.py:215-238 / integrations/sdpa_attention.py), head_dim 64 --------------
 * q [B*Lq rows], k,v [B*Lk rows]: bf16, head h of a row at element offset h*64, row strides ldq/ldk/ldv/ldo
 * (elements).  o bf16 same addressing with ldo; lse f32 [B][H][Lq] (natural-log logsumexp of scale*q.k).
 * softmax(scale * q k^T [+ causal mask]) v; the reference pre-scales q by 0.125 and passes scaling=1.0, which is
 * bit-identical to scale=0.125 here (power of two). */
int dw_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Lq, int Lk,
                int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int causal, float scale, void* stream);
/* Same kernel with explicit batch pitches (rows between consecutive batches of q/o and of k/v): reads a padded KV
 * cache in place during greedy decoding (TF:modeling_whisper.py:312-335 EncoderDecoderCache).  lse may be NULL.
 * causal: 0 none, 1 query i sees keys <= i, 2 bottom-right aligned (query i sees keys <= i + Lk - Lq: several new
 * queries against a longer cache -- the multi-token verify step of speculative decoding, run_eval.py:578-599). */
int dw_attn_fwd_ex(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Lq, int Lk,
                   int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t q_batch_rows, int64_t kv_batch_rows,
                   int causal, float scale, void* stream);
/* Forward over RAGGED batches of packed rows (forward-only passes: the frozen teacher's decoder over the live positions of a
 * batch, run_distillation.py:1472-1480 with the dead decoder positions left out; no lse).  n_seq sequences; sequence i has
 * q_len[i] (1 <= q_len[i] <= max_q) queries at rows q_start[i], q_start[i] + 1, ... of q / o (int32 arrays on the device).
 * Self-attention: kv_start = q_start and kv_len = q_len -- keys / values are rows of the same packed buffers (causal 0 / 1).
 * Cross-attention: kv_start = kv_len = NULL -- k / v are [kv_batches][kv_batch_rows] rectangles with Lk valid rows each and
 * sequence i reads batch min(i, kv_batches - 1) (n_seq may exceed kv_batches by filler sequences whose results are unused).
 * Same arithmetic per query as dw_attn_fwd over the rectangular layout (bit-identical rows). */
int dw_attn_fwd_varlen(const void* q, const void* k, const void* v, void* o, int n_seq, int H, int max_q, int Lk, int64_t ldq,
                       int64_t ldk, int64_t ldv, int64_t ldo, const int32_t* q_start, const int32_t* q_len,
                       const int32_t* kv_start, const int32_t* kv_len, int kv_batches, int64_t kv_batch_rows, int causal,
                       float scale, void* stream);
/* delta: caller-owned f32 scratch of 2*B*H*Lq elements (the dQ kernel stores -rowsum(dO*O) and -lse/scale of its queries
 * there; the dK/dV kernel, launched behind it, starts its accumulators from them).  dq/dk/dv bf16 with row strides
 * lddq/lddk/lddv. */
int dw_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                float* delta, void* dq, void* dk, void* dv, int B, int H, int Lq, int Lk, int64_t ldq, int64_t ldk,
                int64_t ldv, int64_t ldo, int64_t lddo, int64_t lddq, int64_t lddk, int64_t lddv, int causal,
                float scale, void* stream);
/* The same, plus (optional, may be NULL) the bias gradients of the projections that produced q and v: dq_colsum /
 * dv_colsum f32 [B][H*64] receive (ADDED with float atomics) the PER-BATCH-ROW column sums of the dq / dv matrices as
 * stored, from the kernels that write dq / dv instead of a separate pass over them; the caller adds the B partial rows
 * (dw_reduce_slices) into q_proj.bias.grad / v_proj.bias.grad.  Per batch row because every 32-row wave tile of a
 * (batch, head) adds to the same 64 addresses: one [H*64] target for the whole batch serialises 1 500 atomics per
 * address and cost more than the pass it replaced.  (k_proj has no bias, TF:modeling_whisper.py:279.) */
int dw_attn_bwd_ex(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                   float* delta, void* dq, void* dk, void* dv, int B, int H, int Lq, int Lk, int64_t ldq, int64_t ldk,
                   int64_t ldv, int64_t ldo, int64_t lddo, int64_t lddq, int64_t lddk, int64_t lddv, int causal,
                   float scale, float* dq_colsum, float* dv_colsum, void* stream);

/* ---- a7: fused CE + temperature-KL distillation loss (run_distillation.py:1453-1462, 1486-1493 and
 * TF:modeling_whisper.py:1083-1087).  s/t logits bf16 [rows][ld] (V valid columns), labels int64 [rows] (-100 =
 * ignore).  losses f32[4] = {ce, kl*T^2, 0.8-weighted total, n_valid}.  If dlogits != NULL the gradient of
 * grad_scale * total w.r.t. the student logits is written there (bf16 [rows][ld], may alias s_logits; columns
 * V..ld-1 are zeroed).  row_ce/row_kl f32 [rows] scratch; counts int32[2] scratch. */
int dw_distill_loss(const void* s_logits, const void* t_logits, const int64_t* labels, int rows, int V, int64_t ld,
                    float temperature, float ce_weight, float kl_weight, float grad_scale, float* losses,
                    void* dlogits, float* row_ce, float* row_kl, int32_t* counts, void* stream);
/* The same with the mix {ce_weight, kl_weight} read from DEVICE memory (`weights`: f32[2]) by the gradient pass and the
 * total: for a caller whose weights are device scalars -- the upstream gradients autograd hands to the reference's own loss
 * lines `0.8 * ce_loss + kl_weight * kl_loss` (run_distillation.py:1486-1493) when they run over the drop-in modules' lazy
 * logits (distil_whisper_amd.modeling.LazyLogits) -- no host round trip, graph-capturable. */
int dw_distill_loss_w(const void* s_logits, const void* t_logits, const int64_t* labels, int rows, int V, int64_t ld,
                      float temperature, const float* weights, float grad_scale, float* losses, void* dlogits,
                      float* row_ce, float* row_kl, int32_t* counts, void* stream);

/* ---- embeddings (TF:modeling_whisper.py:675-676, 736-762) ------------------------------------------------------
 * out[b*T+t][:] = tok[ids[b*T+t]][:] + pos[t][:]; tables f32 or bf16 (tab_dtype); out f32 or bf16. */
int dw_embed_fwd(const int64_t* ids, const void* tok, const void* pos, int tab_dtype, void* out, int out_dtype,
                 int B, int T, int D, void* stream);
/* dtok[ids[r]] += dx[r] (atomic f32), dpos[t] += sum_b dx[b*T+t] (skipped if dpos NULL). */
int dw_embed_bwd(const float* dx, const int64_t* ids, float* dtok, float* dpos, int B, int T, int D, void* stream);

/* ---- conv front end helpers (TF:modeling_whisper.py:566-567, 618-625) ------------------------------------------
 * conv1 (k3,p1,s1) and conv2 (k3,p1,s2) run as im2col + dw_gemm_bf16 with W' [D][3*C] (W'[d][k*C+c] = W[d][c][k]).
 * im2col_mel: mel f32 [B][C][T] -> xcol bf16 [B*T][kpad], xcol[(b,t)][k*C+c] = mel[b][c][t+k-1], zero padded. */
int dw_im2col_mel(const float* mel, void* xcol, int B, int C, int T, int kpad, void* stream);
/* im2col_s2: a bf16 [B*T][C] -> xcol bf16 [B*T/2][3*C], xcol[(b,t)][k*C+c] = a[b][2t+k-1][c]. */
int dw_im2col_s2(const void* a, void* xcol, int B, int T, int C, void* stream);
/* col2im_s2 + GELU backward: dz[b][r][c] = gelu'(z[b][r][c]) * sum_{2t+k-1=r} dxcol[(b,t)][k*C+c]. */
int dw_col2im_s2_gelu_bwd(const void* dxcol, const void* z, void* dz, int B, int T, int C, void* stream);
/* dz = bf16(dy) * gelu'(z): backward of the GELU after conv2 (dy f32 or bf16, z/dz bf16, n %% 4 == 0). */
int dw_gelu_bwd(const void* dy, int dy_dtype, const void* z, void* dz, int64_t n, void* stream);
/* conv weight <-> GEMM weight layouts: w [D][C][3] f32 <-> wp [D][kpad] (bf16 pack / f32 grad unpack-accumulate). */
int dw_pack_conv_weight(const float* w, void* wp_bf16, int D, int C, int kpad, void* stream);
int dw_unpack_conv_grad(const float* gwp, float* gw, int D, int C, int kpad, int accumulate, void* stream);

/* ---- small streaming kernels -----------------------------------------------------------------------------------*/
int dw_cast_f32_bf16(const float* x, void* y, int64_t n, void* stream);
int dw_cast_bf16_f32(const void* x, float* y, int64_t n, void* stream);
/* out[n] (+)= sum_r x[r][n], x bf16 [rows][ld] (bias gradients). */
int dw_colsum_bf16(const void* x, int64_t ld, int rows, int cols, float* out, int accumulate, void* stream);
/* Rows through an index list (int32, device): gather (scatter = 0) dst[i][:] = src[idx[i]][:], scatter dst[idx[i]][:] =
 * src[i][:], i < n; row_bytes and both pitches multiples of 16.  The padding-free decoder pass keeps [live rows][D]
 * activations between the attention calls -- the reference computes all 448 padded positions of every row of the batch
 * (collator, run_distillation.py:405-478; decoder, TF:modeling_whisper.py:690-795). */
int dw_move_rows(const void* src, int64_t src_pitch_bytes, void* dst, int64_t dst_pitch_bytes, const int32_t* idx, int n,
                 int row_bytes, int scatter, void* stream);
/* y = a (bf16/f32) + b (bf16/f32) elementwise into f32 or bf16 */
int dw_add(const void* a, int a_dtype, const void* b, int b_dtype, void* y, int y_dtype, int64_t n, void* stream);

/* ---- a9: global-norm clip + AdamW (run_distillation.py:1377-1407, 1611-1614; torch.optim.AdamW semantics) -------
 * sumsq: out[0] += sum(g^2) (zero it first).  adamw: clip = min(1, max_norm/(sqrt(sumsq[0])+1e-6)) (max_norm <= 0
 * disables), g' = g*clip*grad_mul; p *= 1-lr*wd; m,v update; p -= lr/bc1 * m/(sqrt(v)/sqrt(bc2)+eps); the bf16
 * shadow copy used by the GEMMs is refreshed in the same pass (shadow may be NULL). */
#define DW_SUMSQ_PARTIALS 2048
/* partials: caller-owned scratch of DW_SUMSQ_PARTIALS floats.  Two deterministic stages (per-block partials, then one
 * block adds them in a fixed order; no float atomics): data-parallel replicas must derive bit-identical clip
 * coefficients from their bit-identical all-reduced gradients. */
int dw_sumsq_f32(const float* g, int64_t n, float* out, float* partials, void* stream);
int dw_adamw(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, const float* sumsq,
             float max_norm, float grad_mul, double lr, double beta1, double beta2, double eps, double weight_decay,
             int step, void* stream);
/* The same update with the optimizer's scalar state RESIDENT ON THE DEVICE, so that one whole training step (forward,
 * backward, clip, AdamW) is a fixed launch sequence that can be captured in a HIP graph and replayed: nothing the host
 * passes by value changes from step to step.  state: 8 doubles, caller-owned: [0] lr (the host rewrites it before a step
 * when an LR scheduler runs: get_scheduler, run_distillation.py:1410-1415), [1] number of optimizer steps taken so far,
 * [2] beta1, [3] beta2, [4..6] written by dw_adam_tick for the update kernels of this step, [7] unused.
 * dw_adam_tick: once per optimizer step before the dw_adamw_dev calls of that step: advances state[1] and derives
 * lr/(1-beta1^step), sqrt(1-beta2^step).  gate (optional, f32 device scalar): when *gate <= 0 the whole step is skipped
 * (nothing advances, no parameter changes) -- the trainer passes the loss kernel's n_valid so that a batch without a
 * single label does not move the weights by momentum / weight decay alone. */
int dw_adam_tick(double* state, const float* gate, void* stream);
int dw_adamw_dev(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, const float* sumsq,
                 float max_norm, float grad_mul, const double* state, double eps, double weight_decay, void* stream);

/* ---- a11: token selection of one greedy-decoding step for the whole batch (TF:generation/logits_process.py processors
 * MinNewTokensLength, SuppressTokensAtBegin, SuppressTokens, WhisperTimeStamp as installed by
 * TF:models/whisper/generation_whisper.py:1774-1812, then argmax and the EOS / pad bookkeeping of GenerationMixin).
 * logits bf16 [B][ld] (V valid columns); suppress / begin_suppress: uint8 [V] masks (1 = never sampled) or NULL,
 * begin_suppress applies when first != 0; no_eos != 0 masks eos (min_new_tokens not reached); forced != 0: position n is
 * still inside the forced prefix (cur = tokens[b][n], nothing else happens).  Timestamp rules: ts_begin =
 * no_timestamps_token_id + 1 (< 0 disables), max_initial = max_initial_timestamp_index (< 0 none), begin_index = length
 * of the forced prefix.  tokens int64 [B][tok_ld]: history in [0, n), the selected token is written to [n]; cur int64 [B]
 * receives it as well; done uint8 [B] in/out (finished rows get `fill`); eos < 0 disables the bookkeeping. */
int dw_greedy_select(const void* logits, int B, int V, int64_t ld, const uint8_t* suppress,
                     const uint8_t* begin_suppress, int first, int no_eos, int forced, int ts_begin, int max_initial,
                     int64_t* tokens, int64_t tok_ld, int n, int begin_index, int eos, int fill, uint8_t* done,
                     int64_t* cur, void* stream);

/* ---- a11: one decoder pass of cached greedy decoding as ONE call (the `decode_step` entry of SURVEY.md 8b).
 * Replaces `WhisperDecoder.forward` + `proj_out` on the cache branch (TF:modeling_whisper.py:690-795, 312-335, 1080)
 * as reached from `generate` (run_eval.py:739, run_distillation.py:1524-1528, run_pseudo_labelling.py:861-996).
 * Every launch of the pass is enqueued on `stream`; nothing is allocated, nothing synchronises (HIP-graph capturable).
 * All matrices bf16 in nn.Linear layout [out][in]; biases and LayerNorm parameters f32; q scaling 0.125 is applied
 * inside attention.  The caller owns every buffer (weights, caches, workspace). */
typedef struct DwDecoderLayer {
    const float *ln1_g, *ln1_b;  /* self_attn_layer_norm */
    const void* wqkv;            /* [3D][D]: q_proj | k_proj | v_proj weights stacked */
    const float* bqkv;           /* [3D] (the k part is zero: k_proj has no bias) */
    const void* wo;  const float* bo;    /* self_attn.out_proj */
    const float *ln2_g, *ln2_b;  /* encoder_attn_layer_norm */
    const void* wq;  const float* bq;    /* encoder_attn.q_proj */
    const void* wo2; const float* bo2;   /* encoder_attn.out_proj */
    const float *ln3_g, *ln3_b;  /* final_layer_norm */
    const void* w1;  const float* b1;    /* fc1 [ffn][D] */
    const void* w2;  const float* b2;    /* fc2 [D][ffn] */
    void* self_kv;               /* bf16 [batch][max_len][2D]: K | V of the generated prefix, appended in place */
    const void* cross_kv;        /* bf16 [batch*src_len][2D]: K | V of the encoder states (projected once per batch); rows
                                    DwDecodeStep.cross_kv_ld elements apart */
} DwDecoderLayer;
typedef struct DwDecodeStep {
    int32_t batch, n_new;        /* n_new new positions per sequence (1 = token step; > 1 = prefill / verify pass) */
    int32_t d_model, heads, ffn, n_layers;
    int32_t src_len, max_len;    /* encoder positions (1500); capacity of the self-attention cache */
    int32_t t;                   /* position of the first new token = number of cached positions */
    int32_t vocab, ldv;          /* LM-head rows used / padded (multiple of 16) = row pitch of logits */
    int32_t stream_dtype;        /* residual stream and embedding tables: DW_F32 (autocast student) or DW_BF16 */
    int32_t cross_kv_ld;         /* row pitch of every layer's cross_kv in elements; 0 = 2 * d_model.  (Rows padded by 128 bytes
                                    stream faster: the 128-byte K / V pieces of a head in consecutive 5 120-byte rows fall on four
                                    of an XCD's sixteen L2 channels -- 29.8 -> 26.4 us per layer at batch 16.) */
    const int64_t* ids;          /* [batch][n_new] */
    const void* tok_emb;         /* [vocab..][D] */
    const void* pos_emb;         /* [max_target_positions][D] (row t .. t+n_new-1 are used) */
    const float *lnf_g, *lnf_b;  /* decoder.layer_norm */
    const void* lm_head;         /* bf16 [ldv][D] (tied embedding, zero-padded rows) */
    const DwDecoderLayer* layers;
    void *x, *h, *qkv, *o, *a;   /* workspace, rows = batch*n_new: x [rows][D] stream dtype; h, o [rows][D], qkv
                                    [rows][3D], a [rows][ffn] bf16 */
    void* logits;                /* out: bf16 [rows][ldv] */
} DwDecodeStep;
int dw_decode_step(const DwDecodeStep* d, void* stream);

/* ---- self tests (diagnostics for bring-up; not on the hot path) --------------------------------------------------
 * Runs ds_read_b64_tr_b16 on a known LDS image: out int32 [64][4] = element ids received by each lane. */
int dw_selftest_tr16(int32_t* out, void* stream);
/* Tuning knobs for kernel A/B experiments and tests; not part of the hot path (defaults are the measured best).
 *   key 0   GEMM kernel selection, bit mask (default 2163 = 115 | 2048): bits 0-1 base 16-wave tile kernel; bit 2
 *           phase-pipelined kernel for dX GEMMs with K >= 3840; bits 4/5/6 8-wave software-pipelined kernel for row-major /
 *           k-major-B / both-k-major operands; bit 7 phase-pipelined kernel for every dX GEMM; bit 8 row-major kernel with
 *           the operand DMA through the compiler builtin (A/B reference); bit 11 (2048) 320 x 256 block tiles for
 *           row-major A where they pay (M % 320 == 0, N % 256 == 0; K >= 2560, or row-major B with N >= 2560; a grid
 *           of one round of them instead of 128-tiles), bit 12 (4096) wherever eligible.  All bit-identical.
 *           Bits 9 / 10 (512 / 1024): main-loop ablations of the row-major kernel for profiling -- no fragment reads /
 *           no operand DMA in the K loop (WRONG results by construction; tools/gemm_overhead.py, gemm_power_probe.py).
 *   key 1   rasterisation strip width override (0 = rule);   key 6  strip L2 budget in 512 KiB units (default 8)
 *   key 2   persistent workgroups on/off;   key 9  persistent grid size in CUs (multiple of 8, default 256)
 *   key 10  dynamic per-XCD tile hand-out (default 1);   key 11  profiling: bit 4 (16) makes the software-pipelined
 *           kernels skip their epilogue (nothing is stored; tools/gemm_overhead.py); other bits unused (default 1)
 *   key 3   attention backward variant (bit 0 dQ, bit 1 dK/dV fast tile staging, bit 2 dK/dV at 3 waves per SIMD -- bits 1
 *           and 2 apply to the NON-causal dK/dV kernel only: the causal launch always runs two waves per SIMD; default 5);   key 4  single-query attention kernel (bit 0 on [default], bit 1 all-loads-up-front variant)
 *   key 5   log-mel DFT on the matrix cores (default 1);   key 7  decode-step fusions off (bit 0 LayerNorm-on-load,
 *           bit 1 K/V append, bit 2 self-attention with its q / k / v projection inside, bit 3 cross-attention with its q
 *           projection inside; default 4);   key 8  token-step GEMVs, bit mask (default 5): bit 0 wide LM-head kernel; bit 1 LayerNorm-on-load kernels keep one
 *           column block per workgroup; bits 2-3 columns per workgroup of the projections back to d_model (4: 8, 0: 4, 12: 16)
 *   key 19  row-major 256-row GEMMs run main-loop ablation `value` (1 no fragment reads, 2 no operand DMA, 3 both, 4 DMA that
 *           always hits L2; WRONG results by construction; tools/gemm_dma_diag.py);   key 20  bit mask: software-pipelined GEMM
 *           kernels on v_mfma_f32_16x16x32_bf16 (1 row-major, 2 k-major B, 4 both k-major, 32 row-major with K <= 2560 and
 *           N >= 3840 on the 256-row tile; default 36; bit-identical results;
 *           8 / 16: the four-wave 128 x 128-per-wave experiment for row-major / k-major B, gemm_wp16_w4.hip)
 *   key 21  LayerNorm kernels: bit 0 persistent fp32-input forward with next-row prefetch (bits 8-11: workgroups per CU),
 *           bit 1 backward with next-row / residual-gradient prefetch at two waves per SIMD (default 3; tools/ln_ab.py)
 *   key 22  wide row-major 256-row launches hand a partial last row block (<= 128 rows) to the 128-tile kernel when the full
 *           row blocks alone need one round of the CUs less (default 1; bit-identical)
 *   key 23  attention forward: threshold (in powers of two) by which a tile maximum must exceed the running reference of
 *           the online softmax before the reference moves (default 8; 0 = the exact running maximum, the A/B leg of
 *           tests/test_sharp_parity_gpu.py)
 *   key 24  128-tile launches on the four-wave tile (2 x 2 waves of 64 x 64; 1 plain K loop, 2 register double buffer;
 *           default 0 = eight waves of 64 x 32; bit-identical)
 *   key 25  outputs with fewer than two rounds of 256-row tiles (the decoders' M = 32 x live positions): kernel chosen by the
 *           rounds of the CUs it needs among 128 x 256 tiles in a three-stage operand ring (csrc/gemm_wp8_m128.hip), 256 x 256
 *           on 16x16x32 and 320 x 256 (default 1; 0 = the lock-step 128 x 128 kernel; 2 = as 1, and DwGemm.tile 129 forces the
 *           128-row kernel at any size; bit-identical)
 *   key 26  non-causal attention forward on the software-pipelined kernel (scores of key tile t+1 under the softmax of tile
 *           t, three K / V stages; 1: three waves per SIMD, 2: two).  Default 0: measured 8 % / 15 % SLOWER than the four-waves-
 *           per-SIMD kernel at the encoder shape (tools/attn_pipe_ab.py), bit-identical -- kept as the A/B of that design
 */
int dw_debug_set(int key, int value);

#ifdef __cplusplus
}
#endif
#endif
