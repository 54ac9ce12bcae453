// Fused cross-entropy + temperature-scaled KL distillation loss and its gradient (HBM-bound, one block per token row).
//
// Replaces, in one pass structure over the two bf16 logit rows:
//   * CrossEntropyLoss(ignore_index=-100, mean) on the student logits (TF:modeling_whisper.py:1083-1087),
//   * softmax(teacher/T), log_softmax(student/T), nn.KLDivLoss(reduction="none"), labels>=0 mask, sum / n_valid, * T^2
//     (run_distillation.py:1453-1462, 1486-1490) -- the reference materialises three fp32 [B,447,V] temporaries,
//   * loss = 0.8*ce + kl_weight*kl (run_distillation.py:1493) and d(loss)/d(student logits).
// Math (per row, z_s / z_t = student / teacher logits, m = row max, T = temperature):
//   ce  = m_s + log sum exp(z_s - m_s) - z_s[label]
//   kl  = [sum_v e_t(v) (z_t(v) - z_s(v))] / (T * Z_t) - (m_t/T + log Z_t) + (m_s/T + log Z_sT),
//         e_t(v) = exp((z_t(v)-m_t)/T), Z_t = sum e_t, Z_sT = sum exp((z_s-m_s)/T)
//   grad(v) = gs * [ cw/n_ce * (softmax(z_s)(v) - [v==label]) + kw*T/n_kl * (softmax(z_s/T)(v) - softmax(z_t/T)(v)) ]
#include "common.h"
#include "../../include/dwamd.h"

#define LOSS_NT 256
#define NEG_INF_F() (-3.0e38f)

__global__ __launch_bounds__(LOSS_NT) void loss_count_kernel(const int64_t* labels, int rows, int32_t* counts) {
    __shared__ float red[LOSS_NT / 64];
    float n_ce = 0.f, n_kl = 0.f;
    for (int i = threadIdx.x; i < rows; i += LOSS_NT) {
        const int64_t l = labels[i];
        n_ce += (l != -100) ? 1.f : 0.f;
        n_kl += (l >= 0) ? 1.f : 0.f;
    }
    n_ce = block_sum<LOSS_NT>(n_ce, red);
    n_kl = block_sum<LOSS_NT>(n_kl, red);
    if (threadIdx.x == 0) { counts[0] = (int32_t)n_ce; counts[1] = (int32_t)n_kl; }
}

__global__ __launch_bounds__(LOSS_NT) void loss_row_kernel(const bf16* s_logits, const bf16* t_logits,
                                                           const int64_t* labels, int V, long ld, float T,
                                                           float ce_w, float kl_w, float grad_scale, bf16* dlogits,
                                                           float* row_ce, float* row_kl, const int32_t* counts,
                                                           const float* weights) {
    __shared__ float red[LOSS_NT / 64];
    const int row = blockIdx.x;
    const int tid = threadIdx.x;
    const bf16* zs = s_logits + (long)row * ld;
    const bf16* zt = t_logits + (long)row * ld;
    const int64_t label = labels[row];
    const bool ce_ok = label != -100 && label >= 0 && label < V;
    const bool kl_ok = label >= 0;
    const float invT = 1.0f / T;
    const float z_label = ce_ok ? bf2f(zs[label]) : 0.f;  // read before any in-place gradient write

    // pass A: row maxima
    float ms = NEG_INF_F(), mt = NEG_INF_F();
    for (int i = tid * 8; i < V; i += LOSS_NT * 8) {
        const bf16x8 a = *(const bf16x8*)(zs + i);
        const bf16x8 b = *(const bf16x8*)(zt + i);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (i + e < V) { ms = fmaxf(ms, bf2f(a[e])); mt = fmaxf(mt, bf2f(b[e])); }
    }
    ms = block_max<LOSS_NT>(ms, red);
    mt = block_max<LOSS_NT>(mt, red);

    // pass B: partition functions and the KL cross term
    float zs1 = 0.f, zsT = 0.f, ztT = 0.f, w = 0.f;
    for (int i = tid * 8; i < V; i += LOSS_NT * 8) {
        const bf16x8 a = *(const bf16x8*)(zs + i);
        const bf16x8 b = *(const bf16x8*)(zt + i);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (i + e < V) {
                const float s = bf2f(a[e]), t = bf2f(b[e]);
                zs1 += __expf(s - ms);
                zsT += __expf((s - ms) * invT);
                const float et = __expf((t - mt) * invT);
                ztT += et;
                w += et * (t - s);
            }
    }
    zs1 = block_sum<LOSS_NT>(zs1, red);
    zsT = block_sum<LOSS_NT>(zsT, red);
    ztT = block_sum<LOSS_NT>(ztT, red);
    w = block_sum<LOSS_NT>(w, red);

    if (tid == 0) {
        const float ce = ce_ok ? (ms + __logf(zs1) - z_label) : 0.f;
        const float kl = kl_ok ? (w / ztT * invT - (mt * invT + __logf(ztT)) + (ms * invT + __logf(zsT))) : 0.f;
        row_ce[row] = ce;
        row_kl[row] = kl;
    }
    if (!dlogits) return;

    // pass C: gradient w.r.t. the student logits (written in place of / next to them)
    if (weights) { ce_w = weights[0]; kl_w = weights[1]; }      // device-resident mix (dw_distill_loss_w)
    const float n_ce = (float)max(counts[0], 1), n_kl = (float)max(counts[1], 1);
    const float gce = ce_ok ? grad_scale * ce_w / n_ce : 0.f;
    const float gkl = kl_ok ? grad_scale * kl_w * T / n_kl : 0.f;  // T^2 * (1/T)
    const float i1 = 1.0f / zs1, iS = 1.0f / zsT, iT = 1.0f / ztT;
    bf16* dz = dlogits + (long)row * ld;
    for (int i = tid * 8; i < ld; i += LOSS_NT * 8) {
        bf16x8 o;
        if (i < V) {
            const bf16x8 a = *(const bf16x8*)(zs + i);
            const bf16x8 b = *(const bf16x8*)(zt + i);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float g = 0.f;
                if (i + e < V) {
                    const float s = bf2f(a[e]), t = bf2f(b[e]);
                    const float p1 = __expf(s - ms) * i1;
                    const float pS = __expf((s - ms) * invT) * iS;
                    const float pT = __expf((t - mt) * invT) * iT;
                    g = gce * (p1 - ((int64_t)(i + e) == label ? 1.f : 0.f)) + gkl * (pS - pT);
                }
                o[e] = f2bf(g);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf(0.f);
        }
#ifdef DW_NT_LOSS        // (experiment)
        __builtin_nontemporal_store(o, (bf16x8*)(dz + i));
#else
        *(bf16x8*)(dz + i) = o;
#endif
    }
}

__global__ __launch_bounds__(LOSS_NT) void loss_final_kernel(const float* row_ce, const float* row_kl, int rows,
                                                             const int32_t* counts, float T, float ce_w, float kl_w,
                                                             float* losses, const float* weights) {
    __shared__ float red[LOSS_NT / 64];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < rows; i += LOSS_NT) { a += row_ce[i]; b += row_kl[i]; }
    a = block_sum<LOSS_NT>(a, red);
    b = block_sum<LOSS_NT>(b, red);
    if (weights) { ce_w = weights[0]; kl_w = weights[1]; }
    if (threadIdx.x == 0) {
        // no valid label in the batch: the reference's means are 0/0 = NaN (CrossEntropyLoss over zero tokens,
        // kl.sum() / padding_mask.sum()); report NaN as well so that the empty batch is visible (the gradient written
        // by the row kernel is zero in that case; losses[3] = 0 is the gate of dw_adam_tick, which then skips the whole
        // optimizer step on the device: zero gradients alone would still move the weights by momentum and weight decay)
        const float ce = a / (float)counts[0];
        const float kl = b / (float)counts[1] * T * T;
        losses[0] = ce;
        losses[1] = kl;
        losses[2] = ce_w * ce + kl_w * kl;
        losses[3] = (float)counts[0];
    }
}

static int distill_loss_launch(const void* s_logits, const void* t_logits, const int64_t* labels, int rows, int V,
                               int64_t ld, float temperature, float ce_weight, float kl_weight, const float* weights,
                               float grad_scale, float* losses, void* dlogits, float* row_ce, float* row_kl,
                               int32_t* counts, void* stream) {
    DW_CLEAR_ERR();
    if (!s_logits || !t_logits || !labels || !losses || !row_ce || !row_kl || !counts) return DW_EINVAL;
    if (rows <= 0 || V <= 0 || ld < V || (ld & 7) || temperature <= 0.f) return DW_EINVAL;
    if (((uintptr_t)s_logits & 15) || ((uintptr_t)t_logits & 15) || ((uintptr_t)dlogits & 15)) return DW_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(loss_count_kernel, dim3(1), dim3(LOSS_NT), 0, s, labels, rows, counts);
    hipLaunchKernelGGL(loss_row_kernel, dim3(rows), dim3(LOSS_NT), 0, s, (const bf16*)s_logits, (const bf16*)t_logits,
                       labels, V, (long)ld, temperature, ce_weight, kl_weight, grad_scale, (bf16*)dlogits, row_ce,
                       row_kl, counts, weights);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(LOSS_NT), 0, s, row_ce, row_kl, rows, counts, temperature,
                       ce_weight, kl_weight, losses, weights);
    DW_CHECK_LAUNCH();
    return DW_OK;
}

extern "C" int dw_distill_loss(const void* s_logits, const void* t_logits, const int64_t* labels, int rows, int V,
                               int64_t ld, float temperature, float ce_weight, float kl_weight, float grad_scale,
                               float* losses, void* dlogits, float* row_ce, float* row_kl, int32_t* counts,
                               void* stream) {
    return distill_loss_launch(s_logits, t_logits, labels, rows, V, ld, temperature, ce_weight, kl_weight, nullptr, grad_scale,
                               losses, dlogits, row_ce, row_kl, counts, stream);
}

// The same with the loss mix (ce_weight, kl_weight) read from DEVICE memory (f32[2]): the weights of a caller whose mix is the
// result of device arithmetic -- autograd's upstream gradients of the reference's own loss lines (run_distillation.py:1486-1493
// over distil_whisper_amd.modeling.LazyLogits) -- without a host round trip.
extern "C" int dw_distill_loss_w(const void* s_logits, const void* t_logits, const int64_t* labels, int rows, int V,
                                 int64_t ld, float temperature, const float* weights, float grad_scale, float* losses,
                                 void* dlogits, float* row_ce, float* row_kl, int32_t* counts, void* stream) {
    if (!weights || ((uintptr_t)weights & 3)) return DW_EINVAL;
    return distill_loss_launch(s_logits, t_logits, labels, rows, V, ld, temperature, 0.f, 0.f, weights, grad_scale, losses,
                               dlogits, row_ce, row_kl, counts, stream);
}
