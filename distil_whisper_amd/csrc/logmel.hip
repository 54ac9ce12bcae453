// Whisper log-mel front end on gfx950: reflect-padded centre STFT (n_fft 400, hop 160, periodic Hann), power
// spectrum, mel projection, log10, per-clip dynamic-range clip and affine -- audio in HBM -> features in HBM with
// one coalesced read of the waveform and one coalesced write of the features.
//
// Replaces WhisperFeatureExtractor._torch_extract_fbank_features (TF:feature_extraction_whisper.py:135-168; in-repo
// twin flax/run_distillation.py:988-1007), which the reference runs on CPU dataloader workers.
//
// Each block owns 32 consecutive frames of one clip: the 5360-sample span they cover is staged once in LDS
// (coalesced), every lane owns one DFT bin k and walks n with an exact 400-entry twiddle table (index k*n mod 400,
// no recurrences, so the error is plain fp32 accumulation), reading each sample quad as one LDS broadcast b128.
// fp32 throughout: the reference output is fp32 and the parity target is 1e-4 absolute.
#include "common.h"
#include "../../include/dwamd.h"

#define LM_FR 32            // frames per block
#define LM_SPAN (31 * 160 + 400)  // 5360 samples
#define LM_NBIN 201
#define LM_PWLD 204

__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
    if (v >= 0.f) atomicMax((int*)addr, __float_as_int(v));
    else atomicMin((unsigned int*)addr, __float_as_uint(v));
}

__global__ void logmel_init_kernel(float* clipmax, int batch) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < batch) clipmax[i] = -INFINITY;
}

__global__ __launch_bounds__(256) void logmel_kernel(const float* audio, int n_samples, const float* mel_filters,
                                                     int n_mels, const float* twiddle, const float* window,
                                                     float* out, float* clipmax, int n_frames) {
    __shared__ __attribute__((aligned(16))) float s_x[LM_SPAN];
    __shared__ __attribute__((aligned(16))) float s_tw[800];
    __shared__ __attribute__((aligned(16))) float s_win[400];
    __shared__ __attribute__((aligned(16))) float s_pw[256 * 33];  // power spectrum [32][204], later log-mel [mel][33]
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * LM_FR;
    const float* x = audio + (long)b * n_samples;

    for (int i = tid; i < LM_SPAN; i += 256) {
        int sidx = t0 * 160 - 200 + i;
        if (sidx < 0) sidx = -sidx;                              // reflect (no edge repeat), as torch.stft center=True
        if (sidx >= n_samples) sidx = 2 * (n_samples - 1) - sidx;
        sidx = sidx < 0 ? 0 : (sidx >= n_samples ? n_samples - 1 : sidx);
        s_x[i] = x[sidx];
    }
    for (int i = tid; i < 800; i += 256) s_tw[i] = twiddle[i];
    for (int i = tid; i < 400; i += 256) s_win[i] = window[i];
    __syncthreads();

    // ---- DFT: lane k accumulates re/im of bin k for the block's 32 frames ----
    const int k = tid;
    if (k < LM_NBIN) {
        float re[LM_FR], im[LM_FR];
#pragma unroll
        for (int f = 0; f < LM_FR; ++f) { re[f] = 0.f; im[f] = 0.f; }
        int idx = 0;  // (k * n) mod 400
        for (int n = 0; n < 400; n += 4) {
            float cw[4], sw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float w = s_win[n + u];
                cw[u] = s_tw[2 * idx] * w;
                sw[u] = s_tw[2 * idx + 1] * w;
                idx += k;
                idx = idx >= 400 ? idx - 400 : idx;
            }
#pragma unroll
            for (int f = 0; f < LM_FR; ++f) {
                const f32x4 v = *(const f32x4*)(s_x + f * 160 + n);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    re[f] = fmaf(v[u], cw[u], re[f]);
                    im[f] = fmaf(v[u], sw[u], im[f]);
                }
            }
        }
#pragma unroll
        for (int f = 0; f < LM_FR; ++f) s_pw[f * LM_PWLD + k] = re[f] * re[f] + im[f] * im[f];
    }
    __syncthreads();

    // ---- mel projection + log10: lane j owns mel bin j ----
    float lm[LM_FR];
    float bmax = -INFINITY;
    for (int j0 = 0; j0 < n_mels; j0 += 256) {
        const int j = j0 + tid;
        if (j < n_mels) {
#pragma unroll
            for (int f = 0; f < LM_FR; ++f) lm[f] = 0.f;
            for (int kk = 0; kk < LM_NBIN; ++kk) {
                const float fv = mel_filters[kk * n_mels + j];
#pragma unroll
                for (int f = 0; f < LM_FR; ++f) lm[f] = fmaf(fv, s_pw[f * LM_PWLD + kk], lm[f]);
            }
#pragma unroll
            for (int f = 0; f < LM_FR; ++f) {
                lm[f] = log10f(fmaxf(lm[f], 1e-10f));
                if (t0 + f < n_frames) bmax = fmaxf(bmax, lm[f]);
            }
        }
        __syncthreads();  // everyone is done reading the power spectrum
        if (j < n_mels) {
#pragma unroll
            for (int f = 0; f < LM_FR; ++f) s_pw[(j - j0) * 33 + f] = lm[f];
        }
        __syncthreads();
        // coalesced store: 32 consecutive frames (128 B) per mel row
        const int nj = min(256, n_mels - j0);
        for (int e = tid; e < nj * LM_FR; e += 256) {
            const int jj = e >> 5, f = e & 31;
            if (t0 + f < n_frames) out[((long)b * n_mels + j0 + jj) * n_frames + t0 + f] = s_pw[jj * 33 + f];
        }
        __syncthreads();
    }
    bmax = block_max<256>(bmax, red);
    if (tid == 0) atomic_max_f32(clipmax + b, bmax);
}

__global__ __launch_bounds__(256) void logmel_norm_kernel(float* out, const float* clipmax, long per_clip, long total) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= total) return;
    const float lo = clipmax[i / per_clip] - 8.0f;
    f32x4 v = *(f32x4*)(out + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (fmaxf(v[e], lo) + 4.0f) * 0.25f;
    *(f32x4*)(out + i) = v;
}

extern "C" int dw_logmel(const float* audio, int batch, int n_samples, const float* mel_filters, int n_mels,
                         const float* twiddle, const float* window, float* out, float* clipmax, void* stream) {
    DW_CLEAR_ERR();
    if (!audio || !mel_filters || !twiddle || !window || !out || !clipmax) return DW_EINVAL;
    if (batch <= 0 || n_samples < 400 || (n_samples % 160) || n_mels <= 0 || n_mels > 256) return DW_EINVAL;
    const int n_frames = n_samples / 160;
    if (((long)n_mels * n_frames) & 3) return DW_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(logmel_init_kernel, dim3((batch + 63) / 64), dim3(64), 0, s, clipmax, batch);
    hipLaunchKernelGGL(logmel_kernel, dim3((n_frames + LM_FR - 1) / LM_FR, batch), dim3(256), 0, s, audio, n_samples,
                       mel_filters, n_mels, twiddle, window, out, clipmax, n_frames);
    const long per_clip = (long)n_mels * n_frames, total = per_clip * batch;
    hipLaunchKernelGGL(logmel_norm_kernel, dim3((total / 4 + 255) / 256), dim3(256), 0, s, out, clipmax, per_clip, total);
    DW_CHECK_LAUNCH();
    return DW_OK;
}
