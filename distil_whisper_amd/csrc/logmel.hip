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

// ---------------------------------------------------------------------------------------------------------------
// MFMA version: the 400-point real DFT of 32 frames as four small fp32 GEMMs on v_mfma_f32_32x32x2_f32.
//
// fp32-input MFMA is an exact fp32 FMA chain (no bf16 rounding of the samples or of the twiddles: the parity target is
// 1e-4 absolute on the log-mel output, and a weak bin next to a strong one needs the full fp32 dynamic range), and it
// leaves the VALU free.  The work is cut four-fold by the two symmetries of a real DFT of even length N = 400:
//   * n <-> N-n:  re[k] = sum_{n=1..200} e[n] cos(2 pi k n / N),  im[k] = sum_{n=1..199} o[n] sin(2 pi k n / N)
//                 with e[n] = y[n] + y[N-n] (e[200] = y[200]), o[n] = y[n] - y[N-n], y = window * samples (y[0] = 0:
//                 the periodic Hann window starts at 0);
//   * k <-> N/2-k: cos(2 pi (200-k) n / N) = (-1)^n cos(2 pi k n / N) and sin(...) = -(-1)^n sin(...), so with
//                 Ce/Co/Se/So[k] = the even-n / odd-n parts of the two sums for k = 0..100 only:
//                 re[k] = Ce + Co, re[200-k] = Ce - Co, |im[k]| = |Se + So|, |im[200-k]| = |Se - So|.
// => four GEMMs [32 frames x 100] x [100 x 101] per tile instead of one [32 x 400] x [400 x 402]: 40.4 k MACs per
// frame instead of 160.8 k (7.8 GFLOP for a batch of 32 clips: ~70 us of matrix-pipe time at the fp32 MFMA rate).
// Workgroup = 16 waves; wave w owns sub-GEMM w/4 (Ce, Co, Se, So) and bins 32*(w%4) .. +31: its B fragments (the
// twiddles, 50 registers) stay in registers for all tiles of the workgroup, the A fragments (folded frames) are one
// ds_read_b32 per MFMA from an LDS image with an odd row stride.  Power spectrum, mel projection (sparse: a filter only
// visits the bins of its triangle), log10 and the coalesced store follow in the same workgroup.
// ---------------------------------------------------------------------------------------------------------------
#define LQ_TPB 12                       // frame tiles per workgroup (B fragments are set up once per workgroup)
#define LQ_AST 101                      // row stride (floats) of the folded-frame images: odd -> conflict-free columns
#define LQ_DST 132                      // row stride of the GEMM outputs [frame][128 bins + pad]

__global__ __launch_bounds__(1024) void logmel_mfma_kernel(const float* audio, int n_samples, const float* mel_filters,
                                                          int n_mels, const float* twiddle, const float* window,
                                                          float* out, float* clipmax, int n_frames) {
    // one static LDS array (120 KB), carved by hand
    __shared__ __attribute__((aligned(16))) float lsm[4 * 32 * LQ_AST + 4 * 32 * LQ_DST + 400 + 2 * 256 + 16];
    float* s_a = lsm;                                   // [4][32][LQ_AST] folded frames; later P [32][204]
    float* s_d = s_a + 4 * 32 * LQ_AST;                 // [4][32][LQ_DST] GEMM outputs; before: the sample span
    float* s_win = s_d + 4 * 32 * LQ_DST;               // [400]
    int* s_rng = (int*)(s_win + 400);                   // [n_mels][2]: first / last+1 bin with a non-zero filter weight
    float* s_red = (float*)(s_rng + 2 * 256);           // [16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = wave >> 2, jt = wave & 3;
    const int b = blockIdx.y;
    const float* x = audio + (long)b * n_samples;

    for (int i = tid; i < 400; i += 1024) s_win[i] = window[i];
    for (int m = tid; m < n_mels; m += 1024) {          // bin range of every mel filter (generic: any sparse bank)
        int lo = LM_NBIN, hi = 0;
        for (int k = 0; k < LM_NBIN; ++k)
            if (mel_filters[k * n_mels + m] != 0.f) { lo = min(lo, k); hi = k + 1; }
        s_rng[2 * m] = lo; s_rng[2 * m + 1] = hi;
    }
    // B fragments: lane (bin j = 32 jt + (lane & 31), kk = lane >> 5) holds basis[n(q = 2 s + kk)][j] for s = 0..49
    float bfr[50];
    {
        const int k = jt * 32 + (lane & 31), kk = lane >> 5;
#pragma unroll
        for (int s2 = 0; s2 < 50; ++s2) {
            const int q = 2 * s2 + kk;
            const int n = (sub & 1) ? 2 * q + 1 : 2 * q + 2;              // Ce, Se: even n; Co, So: odd n
            const bool dead = k > 100 || (sub == 2 && q == 99);             // (o[200] = 0: sin(pi k) anyway)
            const float t = twiddle[2 * ((k * n) % 400) + (sub >> 1)];     // cos for Ce/Co, sin for Se/So
            bfr[s2] = dead ? 0.f : t;
        }
    }
    float bmax = -INFINITY;
    const int tiles = (n_frames + 31) >> 5;
    for (int tile = blockIdx.x * LQ_TPB; tile < min(tiles, (int)(blockIdx.x + 1) * LQ_TPB); ++tile) {
        const int t0 = tile * 32;
        __syncthreads();                                 // previous tile's output staging is done
        // ---- sample span of the 32 frames (reflect padding at the clip edges), staged where the outputs go later ----
        for (int i = tid; i < LM_SPAN; i += 1024) {
            int sidx = t0 * 160 - 200 + i;
            if (sidx < 0) sidx = -sidx;
            if (sidx >= n_samples) sidx = 2 * (n_samples - 1) - sidx;
            sidx = sidx < 0 ? 0 : (sidx >= n_samples ? n_samples - 1 : sidx);
            s_d[i] = x[sidx];
        }
        __syncthreads();
        // ---- fold: e = y[n] + y[400-n], o = y[n] - y[400-n], split by the parity of n ----
        for (int i = tid; i < 32 * 200; i += 1024) {
            const int f = i / 200, n = i - f * 200 + 1;                      // n = 1 .. 200
            const float* fr = s_d + f * 160;
            const float yn = fr[n] * s_win[n];
            const float ym = n < 200 ? fr[400 - n] * s_win[400 - n] : 0.f;
            const float e = yn + ym, o = n < 200 ? yn - ym : 0.f;
            const int q = (n - 1) >> 1;                                      // n odd: (n-1)/2, n even: n/2 - 1
            if (n & 1) { s_a[(1 * 32 + f) * LQ_AST + q] = e; s_a[(3 * 32 + f) * LQ_AST + q] = o; }
            else       { s_a[(0 * 32 + f) * LQ_AST + q] = e; s_a[(2 * 32 + f) * LQ_AST + q] = o; }
        }
        __syncthreads();
        // ---- [32 frames x 100] x [100 x 32 bins] on the fp32 matrix pipe ----
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        {
            const float* ap = s_a + (sub * 32 + (lane & 31)) * LQ_AST + (lane >> 5);
#pragma unroll
            for (int s2 = 0; s2 < 50; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s2], bfr[s2], acc, 0, 0, 0);
        }
        // D[frame = (r&3) + 8(r>>2) + 4(lane>>5)][bin = 32 jt + (lane&31)]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            s_d[(sub * 32 + f) * LQ_DST + jt * 32 + (lane & 31)] = acc[r];
        }
        __syncthreads();
        // ---- power spectrum: bins k and 200-k from the same four numbers ----
        float* s_p = s_a;                                // [32][LM_PWLD]
        for (int i = tid; i < 32 * 101; i += 1024) {
            const int f = i / 101, k = i - f * 101;
            const float ce = s_d[(0 * 32 + f) * LQ_DST + k], co = s_d[(1 * 32 + f) * LQ_DST + k];
            const float se = s_d[(2 * 32 + f) * LQ_DST + k], so = s_d[(3 * 32 + f) * LQ_DST + k];
            const float rp = ce + co, ip = se + so, rm = ce - co, im = se - so;
            s_p[f * LM_PWLD + k] = rp * rp + ip * ip;
            if (k < 100) s_p[f * LM_PWLD + 200 - k] = rm * rm + im * im;
        }
        __syncthreads();
        // ---- mel projection over each filter's own bins + log10; result staged [mel][33] for the coalesced store ----
        float* s_o = s_d;
        for (int i = tid; i < n_mels * 32; i += 1024) {
            const int m = i >> 5, f = i & 31;
            float v = 0.f;
            for (int k = s_rng[2 * m]; k < s_rng[2 * m + 1]; ++k) v = fmaf(mel_filters[k * n_mels + m], s_p[f * LM_PWLD + k], v);
            v = log10f(fmaxf(v, 1e-10f));
            if (t0 + f < n_frames) bmax = fmaxf(bmax, v);
            s_o[m * 33 + f] = v;
        }
        __syncthreads();
        for (int e = tid; e < n_mels * 32; e += 1024) {
            const int m = e >> 5, f = e & 31;
            if (t0 + f < n_frames) out[((long)b * n_mels + m) * n_frames + t0 + f] = s_o[m * 33 + f];
        }
    }
    bmax = wave_max(bmax);
    __syncthreads();
    if (lane == 0) s_red[wave] = bmax;
    __syncthreads();
    if (tid == 0) {
        float t = s_red[0];
        for (int i = 1; i < 16; ++i) t = fmaxf(t, s_red[i]);
        atomic_max_f32(clipmax + b, t);
    }
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

int g_logmel_mfma = 1;   // dw_debug_set key 5: 0 = direct DFT on the VALU (round 1), 1 = folded DFT on the fp32 matrix pipe

extern "C" int dw_logmel(const float* audio, int batch, int n_samples, const float* mel_filters, int n_mels,
                         const float* twiddle, const float* window, float* out, float* clipmax, void* stream) {
    DW_CLEAR_ERR();
    if (!audio || !mel_filters || !twiddle || !window || !out || !clipmax) return DW_EINVAL;
    if (batch <= 0 || n_samples < 400 || (n_samples % 160) || n_mels <= 0 || n_mels > 256) return DW_EINVAL;
    const int n_frames = n_samples / 160;
    if (((long)n_mels * n_frames) & 3) return DW_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(logmel_init_kernel, dim3((batch + 63) / 64), dim3(64), 0, s, clipmax, batch);
    if (g_logmel_mfma) {
        const int tiles = (n_frames + 31) / 32;
        hipLaunchKernelGGL(logmel_mfma_kernel, dim3((tiles + LQ_TPB - 1) / LQ_TPB, batch), dim3(1024), 0, s, audio,
                           n_samples, mel_filters, n_mels, twiddle, window, out, clipmax, n_frames);
    } else {
        hipLaunchKernelGGL(logmel_kernel, dim3((n_frames + LM_FR - 1) / LM_FR, batch), dim3(256), 0, s, audio,
                           n_samples, mel_filters, n_mels, twiddle, window, out, clipmax, n_frames);
    }
    const long per_clip = (long)n_mels * n_frames, total = per_clip * batch;
    hipLaunchKernelGGL(logmel_norm_kernel, dim3((total / 4 + 255) / 256), dim3(256), 0, s, out, clipmax, per_clip, total);
    DW_CHECK_LAUNCH();
    return DW_OK;
}
