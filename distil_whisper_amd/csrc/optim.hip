// Global-norm gradient clipping + AdamW over flat fp32 parameter / gradient / moment buffers (HBM-bound: 16 B read +
// 14 B written per parameter), refreshing the bf16 shadow weights the MFMA GEMMs consume in the same pass.
// Replaces accelerator.clip_grad_norm_ + torch.optim.AdamW.step (run_distillation.py:1377-1407, 1611-1614).
#include "common.h"
#include "../../include/dwamd.h"

// Two deterministic stages (no float atomics): every block stores its partial sum, one block adds the partials in a
// fixed order.  The clip coefficient derived from this norm is baked into every parameter update, so data-parallel
// replicas (which see bit-identical all-reduced gradients) must compute bit-identical norms -- an atomicAdd over up to
// 2048 blocks would make the summation order, and with it the replicas' parameters, run dependent.
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* g, long n, float* partials) {
    __shared__ float red[4];
    float acc = 0.f;
    const long stride = (long)gridDim.x * 256 * 4;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            const f32x4 v = *(const f32x4*)(g + i);
            acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        } else {
            for (long j = i; j < n; ++j) acc += g[j] * g[j];
        }
    }
    acc = block_sum<256>(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* partials, int nb, float* out) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) acc += partials[i];
    acc = block_sum<256>(acc, red);
    if (threadIdx.x == 0) out[0] += acc;
}

__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v, bf16* shadow, long n,
                                                    const float* sumsq, float max_norm, float grad_mul,
                                                    float step_size, float decay, float beta1, float omb1,
                                                    float beta2, float omb2, float eps, float bc2_sqrt) {
    float clip = grad_mul;
    if (max_norm > 0.f && sumsq) {
        const float norm = sqrtf(sumsq[0]) * fabsf(grad_mul);
        const float coef = max_norm / (norm + 1e-6f);
        clip *= coef < 1.f ? coef : 1.f;
    }
    const long stride = (long)gridDim.x * 256 * 4;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            f32x4 pv = *(f32x4*)(p + i);
            const f32x4 gv = *(const f32x4*)(g + i);
            f32x4 mv = *(f32x4*)(m + i);
            f32x4 vv = *(f32x4*)(v + i);
            bf16x4 sh;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gg = gv[e] * clip;
                pv[e] *= decay;
                mv[e] = beta1 * mv[e] + omb1 * gg;
                vv[e] = beta2 * vv[e] + omb2 * (gg * gg);
                const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
                pv[e] -= step_size * (mv[e] / denom);
                sh[e] = f2bf(pv[e]);
            }
            *(f32x4*)(p + i) = pv;
            *(f32x4*)(m + i) = mv;
            *(f32x4*)(v + i) = vv;
            if (shadow) *(bf16x4*)(shadow + i) = sh;
        } else {
            for (long j = i; j < n; ++j) {
                const float gg = g[j] * clip;
                float pj = p[j] * decay;
                const float mj = beta1 * m[j] + omb1 * gg;
                const float vj = beta2 * v[j] + omb2 * (gg * gg);
                pj -= step_size * (mj / (sqrtf(vj) / bc2_sqrt + eps));
                p[j] = pj; m[j] = mj; v[j] = vj;
                if (shadow) shadow[j] = f2bf(pj);
            }
        }
    }
}

extern "C" int dw_sumsq_f32(const float* g, int64_t n, float* out, float* partials, void* stream) {
    DW_CLEAR_ERR();
    if (!g || !out || !partials || n <= 0 || ((uintptr_t)g & 15)) return DW_EINVAL;
    long nb = (n / 4 + 255) / 256;
    if (nb > DW_SUMSQ_PARTIALS) nb = DW_SUMSQ_PARTIALS;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, g, (long)n, partials);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, (int)nb, out);
    DW_CHECK_LAUNCH();
    return DW_OK;
}

extern "C" int dw_adamw(float* p, const float* g, float* m, float* v, void* shadow, int64_t n, const float* sumsq,
                        float max_norm, float grad_mul, double lr, double beta1, double beta2, double eps,
                        double weight_decay, int step, void* stream) {
    DW_CLEAR_ERR();
    if (!p || !g || !m || !v || n <= 0 || step < 1) return DW_EINVAL;
    if (((uintptr_t)p & 15) || ((uintptr_t)g & 15) || ((uintptr_t)m & 15) || ((uintptr_t)v & 15) ||
        ((uintptr_t)shadow & 7))
        return DW_EINVAL;
    // scalar hyper-parameter arithmetic in double, exactly as torch.optim.AdamW does it on the host
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    long nb = (n / 4 + 255) / 256;
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(adamw_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16*)shadow, (long)n,
                       sumsq, max_norm, grad_mul, (float)(lr / bc1), (float)(1.0 - lr * weight_decay), (float)beta1,
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)sqrt(bc2));
    DW_CHECK_LAUNCH();
    return DW_OK;
}

// ---- device-resident optimizer state: the whole training step (including this update) can be captured in a HIP graph --
// state (8 doubles, caller-owned): [0] lr (host writes it before every step: LR scheduler), [1] step count (advanced
// HERE), [2] beta1, [3] beta2; derived by the tick kernel for the update kernels of this step: [4] lr / (1 - beta1^step),
// [5] sqrt(1 - beta2^step), [6] applied (1.0 / 0.0).
__global__ void adam_tick_kernel(double* st, const float* gate) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const bool apply = !gate || gate[0] > 0.f;      // gate = n_valid of the loss kernel: a batch without labels is skipped
    if (apply) st[1] += 1.0;
    const double step = st[1] < 1.0 ? 1.0 : st[1];
    st[4] = st[0] / (1.0 - pow(st[2], step));
    st[5] = sqrt(1.0 - pow(st[3], step));
    st[6] = apply ? 1.0 : 0.0;
}

__global__ __launch_bounds__(256) void adamw_dev_kernel(float* p, const float* g, float* m, float* v, bf16* shadow, long n,
                                                        const float* sumsq, float max_norm, float grad_mul,
                                                        const double* st, double weight_decay, float eps) {
    if (st[6] == 0.0) return;
    const float step_size = (float)st[4], bc2_sqrt = (float)st[5];
    const float decay = (float)(1.0 - st[0] * weight_decay);
    const float beta1 = (float)st[2], omb1 = (float)(1.0 - st[2]), beta2 = (float)st[3], omb2 = (float)(1.0 - st[3]);
    float clip = grad_mul;
    if (max_norm > 0.f && sumsq) {
        const float norm = sqrtf(sumsq[0]) * fabsf(grad_mul);
        const float coef = max_norm / (norm + 1e-6f);
        clip *= coef < 1.f ? coef : 1.f;
    }
    const long stride = (long)gridDim.x * 256 * 4;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
#ifdef DW_NT_ADAMW      // (experiment: everything this kernel touches is read once and written once per step)
            f32x4 pv = __builtin_nontemporal_load((f32x4*)(p + i));
            const f32x4 gv = __builtin_nontemporal_load((const f32x4*)(g + i));
            f32x4 mv = __builtin_nontemporal_load((f32x4*)(m + i));
            f32x4 vv = __builtin_nontemporal_load((f32x4*)(v + i));
#else
            f32x4 pv = *(f32x4*)(p + i);
            const f32x4 gv = *(const f32x4*)(g + i);
            f32x4 mv = *(f32x4*)(m + i);
            f32x4 vv = *(f32x4*)(v + i);
#endif
            bf16x4 sh;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gg = gv[e] * clip;
                pv[e] *= decay;
                mv[e] = beta1 * mv[e] + omb1 * gg;
                vv[e] = beta2 * vv[e] + omb2 * (gg * gg);
                const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
                pv[e] -= step_size * (mv[e] / denom);
                sh[e] = f2bf(pv[e]);
            }
#ifdef DW_NT_ADAMW
            __builtin_nontemporal_store(pv, (f32x4*)(p + i));
            __builtin_nontemporal_store(mv, (f32x4*)(m + i));
            __builtin_nontemporal_store(vv, (f32x4*)(v + i));
            if (shadow) __builtin_nontemporal_store(sh, (bf16x4*)(shadow + i));
#else
            *(f32x4*)(p + i) = pv;
            *(f32x4*)(m + i) = mv;
            *(f32x4*)(v + i) = vv;
            if (shadow) *(bf16x4*)(shadow + i) = sh;
#endif
        } else {
            for (long j = i; j < n; ++j) {
                const float gg = g[j] * clip;
                float pj = p[j] * decay;
                const float mj = beta1 * m[j] + omb1 * gg;
                const float vj = beta2 * v[j] + omb2 * (gg * gg);
                pj -= step_size * (mj / (sqrtf(vj) / bc2_sqrt + eps));
                p[j] = pj; m[j] = mj; v[j] = vj;
                if (shadow) shadow[j] = f2bf(pj);
            }
        }
    }
}

extern "C" int dw_adam_tick(double* state, const float* gate, void* stream) {
    DW_CLEAR_ERR();
    if (!state || ((uintptr_t)state & 7)) return DW_EINVAL;
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, gate);
    DW_CHECK_LAUNCH();
    return DW_OK;
}

extern "C" int dw_adamw_dev(float* p, const float* g, float* m, float* v, void* shadow, int64_t n, const float* sumsq,
                            float max_norm, float grad_mul, const double* state, double eps, double weight_decay,
                            void* stream) {
    DW_CLEAR_ERR();
    if (!p || !g || !m || !v || !state || n <= 0) return DW_EINVAL;
    if (((uintptr_t)p & 15) || ((uintptr_t)g & 15) || ((uintptr_t)m & 15) || ((uintptr_t)v & 15) ||
        ((uintptr_t)shadow & 7) || ((uintptr_t)state & 7))
        return DW_EINVAL;
    long nb = (n / 4 + 255) / 256;
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(adamw_dev_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16*)shadow, (long)n,
                       sumsq, max_norm, grad_mul, state, weight_decay, (float)eps);
    DW_CHECK_LAUNCH();
    return DW_OK;
}
