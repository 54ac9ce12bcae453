// bf16 MFMA GEMM for gfx950 with fused epilogues -- the QKV / out-proj / FFN / LM-head / conv(im2col) contractions
// of the Whisper distillation step, forward and backward.
//
// Replaces nn.Linear/nn.Conv1d matmuls of the reference path (TF:modeling_whisper.py:279-282, 375-376, 444-445,
// 566-567, 970) and their autograd backward (run_distillation.py:1609).
//
// Structure (CDNA4-first, not a warp-shaped port):
//   * block tile BM x BN x 64, waves laid out WM x WN, every wave owns (BM/WM) x (BN/WN) made of 32x32 accumulators
//     fed by v_mfma_f32_32x32x16_bf16 (64-lane wavefront, 16 fp32 accumulators per lane per 32x32 tile); the default
//     256x256 tile runs 16 waves (4x4, 64x64 per wave, <= 128 VGPRs): four resident waves per SIMD hide each
//     other's barrier / vmcnt / LDS latency better than the 8-wave layout with software-pipelined fragments did;
//   * operands go HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip), double buffered, one barrier per
//     K-step; the LDS image is lane-linear, so the bank-conflict swizzle is applied to the per-lane SOURCE address
//     and undone on the fragment read (same involution on both sides);
//   * operands whose contraction index is NOT contiguous in memory (the backward GEMMs: dX = dY.W, dW = dY^T.X) are
//     staged k-major and their MFMA fragments are fetched with ds_read_b64_tr_b16 (hardware 4x16 transpose read), so
//     no transposed copies of activations or weights are ever materialised in HBM;
//   * blockIdx is remapped so that each XCD (private 4 MiB L2) walks a contiguous range of output tiles.
#include "gemm_common.h"
#include <mutex>
#include <unordered_map>

bool dw_gemm_skinny_ok(const GemmP& p, int trans_a, int trans_b);   // gemm_skinny.hip
int dw_gemm_skinny_launch(const GemmP& p, hipStream_t s);
int dw_gemm_phased_launch(const GemmP& p, int ta, int tb, hipStream_t s);  // gemm_phased.hip
int dw_gemm_tile256_launch(const GemmP& p, int ta, int tb, hipStream_t s);  // gemm_tile256.hip: 16 waves, 4 x 4
int dw_gemm_tile128_launch(const GemmP& p, int ta, int tb, hipStream_t s);  // gemm_tile128.hip: 8 waves, 2 x 4
int dw_gemm_tile128w4_launch(const GemmP& p, int ta, int tb, int var, hipStream_t s);  // gemm_tile128w4.hip: 4 waves, 2 x 2
static int g_gemm_t128_w4 = 0;   // dw_debug_set key 24: 128-tile launches on the four-wave tile (1: plain K loop, 2: register double buffer)
int dw_gemm_wp8_nn_ref_launch(const GemmP& p, hipStream_t s);               // gemm_wp8_nn_ref.hip (builtin DMA; A/B only)
int dw_gemm_wp8_nn320_launch(const GemmP& p, hipStream_t s);               // gemm_wp8_m320.hip (320 x 256 block tile)
int dw_gemm_wp8_nt320_launch(const GemmP& p, hipStream_t s);
int dw_gemm_wp8_nn128_launch(const GemmP& p, hipStream_t s);               // gemm_wp8_m128.hip (128 x 256 block tile)
int dw_gemm_wp8_nt128_launch(const GemmP& p, hipStream_t s);
int dw_gemm_wp8_nn_dbg_launch(const GemmP& p, int dbg, hipStream_t s);         // gemm_wp8_dbg.hip (main-loop ablations; profiling only)
int dw_gemm_wp8_nn_launch(const GemmP& p, hipStream_t s);                   // gemm_wp8_*.hip: software-pipelined loop, 8 waves
int dw_gemm_wp8_nt_launch(const GemmP& p, hipStream_t s);
int dw_gemm_wp8_tt_launch(const GemmP& p, hipStream_t s);
int dw_gemm_wp16_nn_w4_launch(const GemmP& p, hipStream_t s);               // gemm_wp16_w4.hip: four waves, 128 x 128 per wave
int dw_gemm_wp16_nt_w4_launch(const GemmP& p, hipStream_t s);
int dw_gemm_wp16_nn_w4_dbg_launch(const GemmP& p, int dbg, hipStream_t s);
int dw_gemm_wp16_nn_launch(const GemmP& p, hipStream_t s);                  // gemm_wp16_*.hip: the same loop on v_mfma_f32_16x16x32_bf16
int dw_gemm_wp16_nn320_launch(const GemmP& p, hipStream_t s);
int dw_gemm_wp16_nt_launch(const GemmP& p, hipStream_t s);
int dw_gemm_wp16_nt320_launch(const GemmP& p, hipStream_t s);
int dw_gemm_wp16_nn_small_launch(const GemmP& p, hipStream_t s);             // gemm_wp16_small.hip (the small-M rule's 256-row launches)
int dw_gemm_wp16_nt_small_launch(const GemmP& p, hipStream_t s);
int dw_gemm_wp16_tt_launch(const GemmP& p, hipStream_t s);
int dw_gemm_wp16_nn_dbg_launch(const GemmP& p, int dbg, hipStream_t s);

extern int g_attn_bwd_stage;  // attention.hip
extern int g_attn_decode;
extern int g_attn_ablate;
extern int g_attn_fwd_waves;
extern int g_attn_fwd_pipe;
extern int g_attn_plain_order;
extern int g_attn_defer;
extern int g_attn_bwd_waves;
extern int g_logmel_mfma;     // logmel.hip
extern int g_decode_fuse_off; // decode.hip
extern int g_skinny_wide;     // gemm_skinny.hip
extern int g_ln_variant;      // norm.hip
int g_gemm_persistent = 1;
// Kernel selection for the 256x256 block tile (bit mask; dw_debug_set(0, v)):
//   bits 0-1 (3): base = 16-wave tile kernel (gemm_kernel.h) for everything, 8-wave 128x128 tile for small grids;
//   bit 2 (4):    phase-pipelined kernel (gemm_phased.hip) for dX GEMMs (k-major B) with K >= 3840: +7..10 % there in the
//                 warm micro-benchmark, a tie in the step once the 8-wave kernels existed (467.6 vs 467.1 ms): off by default;
//   bit 4 (16):   8-wave software-pipelined kernel (gemm_wp.h, 128x64 per wave) for row-major operands;
//   bit 5 (32):   ... for dX GEMMs the phased kernel does not take;   bit 6 (64): ... for dW GEMMs (both k-major);
//   bit 7 (128):  phased kernel for every dX GEMM (tests);
//   bit 8 (256):  row-major 8-wave kernel built with the operand DMA issued through the compiler builtin (A/B reference).
// (gemm_wp.h also instantiates as 4 waves x 128x128 -- one wave per SIMD, half the LDS fragment traffic -- but a lone
// wave cannot cover its own DMA issue slots: 4-10 % behind the 8-wave layout on every shape, not built.)
// Every kernel produces bit-identical results (same fp32 chain over k per output element): tests/test_kernels_gpu.py.
static int g_gemm_variant = 2163;   // 115 (8-wave software-pipelined kernels + phased dX) | 2048 (320-row tiles where they pay)
int g_gemm_strip = 0;
int g_gemm_cus = 256;
static int g_gemm_stage_next = 1;   // dw_debug_set key 11: profiling switches of the software-pipelined kernels (bit 4: skip the epilogue)
static int g_gemm_stagger = 0;   // dw_debug_set key 12: start offsets of the persistent workgroups (S | unit << 8), 0 = none
static unsigned g_gemm_trace_lo = 0, g_gemm_trace_hi = 0;   // dw_debug_set keys 13 / 14: device pointer of the phase-trace buffer
// dw_debug_set key 20, bit mask: the software-pipelined kernels run on v_mfma_f32_16x16x32_bf16 (gemm_wp16.h): 1 row-major, 2 k-major B,
// 4 both k-major.  Default 4: the weight-gradient GEMMs gain 3 % (two transposing LDS reads per fragment either way, and the
// 16-cycle instruction leaves twice the issue gaps for them: 1 127 vs 1 093 TFLOP/s, -2.0 ms per step); the row-major and dX loops
// tie without their epilogue and lose 1-6 % with it (more live registers around the epilogue walks: 152 / 332 B of scratch), so
// they keep 32x32x16 (tools/gemm_mi16_probe.py, tools/gemm_mi16_sustained.py, tools/ab_step.py field 15).  Bit-identical results.
// Bit 32 (default on): row-major GEMMs with K <= 2560, N >= 3840 and a flavour of the accumulator-side walk (QKV, the teacher's fc1)
// run the 256-row tile on 16x16x32 instead of the 320-row tile on 32x32x16: sustained 1 197 vs 1 182 (N = 3840) and 1 240 vs 1 194
// TFLOP/s (N = 5120) once the outputs are stored non-temporally (tools/gemm_w4_probe.py), -0.56 % per step; with the fp32 walk
// (student fc1: two outputs) it loses, +0.5 %.  Bits 8 / 16: the four-wave experiment (gemm_wp16_w4.hip).
static int g_gemm_mi16 = 36;
static int g_gemm_dbg = 0;       // dw_debug_set key 19: row-major 256-row GEMMs run the ablation / experiment kernel `value` of gemm_wp8_dbg.hip
static int g_gemm_dynamic = 1;   // dw_debug_set key 10: dynamic job hand-out in the persistent kernels (gemm_common.h)
static int g_gemm_small_m = 1;   // dw_debug_set key 25: kernel choice by rounds of the CUs for outputs with fewer than two rounds of 256-row tiles (0: the 128 x 128 lock-step kernel)
static int g_gemm_row_tail = 1;  // dw_debug_set key 22: the partial last row block of a wide 256-row launch goes to the 128-tile kernel when that saves a round

// Nine device counters per stream for the dynamic job hand-out of the persistent kernels (kernels of one stream never
// overlap and the last workgroup of a launch leaves them zeroed).  64 bytes are allocated the first time a stream
// launches a persistent GEMM; a stream that is being captured at that moment keeps the static hand-out.
static int* gemm_sched_slot(hipStream_t s) {
    static std::mutex mu;
    static std::unordered_map<hipStream_t, int*> slots;
    std::lock_guard<std::mutex> lock(mu);
    auto it = slots.find(s);
    if (it != slots.end()) return it->second;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return nullptr; }
    int* ptr = nullptr;
    // zeroed ON the stream that will use them: a hipMemset on the null stream is not ordered against a non-blocking
    // stream (torch's side streams), and the first persistent GEMM of a new stream could read uninitialised counters --
    // tile indices out of range, a memory access fault (seen once, under rocprofv3 --pmc with three streams active)
    if (hipMalloc((void**)&ptr, 64) != hipSuccess || hipMemsetAsync(ptr, 0, 64, s) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    slots[s] = ptr;
    return ptr;
}
int g_gemm_strip_budget = 4;   // x 512 KiB of L2 for the resident strip of B tiles (round 6: 8 -> 4, i.e. strips of 3 instead of 6 column tiles at K = 1280: -0.7 % per step on large-v3 and small.en in same-process A/Bs, tools/ab_keys.py; 3 and 5-6 lose)
extern "C" int dw_debug_set(int key, int value) {
    if (key == 0) { g_gemm_variant = value; return DW_OK; }
    if (key == 1) { g_gemm_strip = value; return DW_OK; }
    if (key == 2) { g_gemm_persistent = value; return DW_OK; }
    if (key == 3) { g_attn_bwd_stage = value; return DW_OK; }
    if (key == 4) { g_attn_decode = value; return DW_OK; }
    if (key == 5) { g_logmel_mfma = value; return DW_OK; }
    if (key == 6) { g_gemm_strip_budget = value; return DW_OK; }
    if (key == 7) { g_decode_fuse_off = value; return DW_OK; }
    if (key == 8) { g_skinny_wide = value; return DW_OK; }
    if (key == 10) { g_gemm_dynamic = value; return DW_OK; }
    if (key == 11) { g_gemm_stage_next = value; return DW_OK; }
    if (key == 12) { g_gemm_stagger = value; return DW_OK; }
    if (key == 13) { g_gemm_trace_lo = (unsigned)value; return DW_OK; }
    if (key == 14) { g_gemm_trace_hi = (unsigned)value; return DW_OK; }
    if (key == 17) { g_attn_bwd_waves = value; return DW_OK; }
    if (key == 19) { g_gemm_dbg = value; return DW_OK; }
    if (key == 20) { g_gemm_mi16 = value; return DW_OK; }
    if (key == 21) { g_ln_variant = value; return DW_OK; }
    if (key == 22) { g_gemm_row_tail = value; return DW_OK; }
    if (key == 25) { g_gemm_small_m = value; return DW_OK; }
    if (key == 26) { g_attn_fwd_pipe = value; return DW_OK; }
    if (key == 18) { g_attn_plain_order = value; return DW_OK; }
    if (key == 24) { if (value < 0 || value > 2) return DW_EINVAL; g_gemm_t128_w4 = value; return DW_OK; }
    if (key == 23) { if (value < 0 || value > 64) return DW_EINVAL; g_attn_defer = value; return DW_OK; }
    if (key == 16) { g_attn_fwd_waves = value; return DW_OK; }
    if (key == 15) { g_attn_ablate = value; return DW_OK; }
    if (key == 9) { if (value < 8 || value > 256 || (value & 7)) return DW_EINVAL; g_gemm_cus = value; return DW_OK; }
    return DW_EINVAL;
}

// out[i] (+)= sum_s part[s * stride + i]   (combination of split-K partial tiles; deterministic, no atomics)
__global__ __launch_bounds__(256) void reduce_slices_kernel(const float* part, long stride, int slices, float* out,
                                                            long n, int accumulate) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    f32x4 acc = accumulate ? *(const f32x4*)(out + i) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < slices; ++s) {
        const f32x4 v = *(const f32x4*)(part + s * stride + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += v[e];
    }
    *(f32x4*)(out + i) = acc;
}

extern "C" int dw_reduce_slices(const float* part, int64_t stride, int slices, float* out, int64_t n, int accumulate,
                                void* stream) {
    DW_CLEAR_ERR();
    if (!part || !out || slices < 1 || n <= 0 || (n & 3) || (stride & 3) || ((uintptr_t)part & 15) ||
        ((uintptr_t)out & 15))
        return DW_EINVAL;
    hipLaunchKernelGGL(reduce_slices_kernel, dim3((n / 4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, part,
                       (long)stride, slices, out, (long)n, accumulate);
    DW_CHECK_LAUNCH();
    return DW_OK;
}

// The same for slabs whose rows are ld_part elements apart (rows padded against L2-channel camping: a [1280 x 5120] fp32 slab has
// 20 480-byte rows, and the 1 KiB row pieces of a 256-column tile then all land on the same four channels); out rows ld_out apart.
__global__ __launch_bounds__(256) void reduce_slices_ld_kernel(const float* part, long slice_stride, long ld_part, int slices,
                                                               float* out, long ld_out, int rows, int cols4, int accumulate) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)rows * cols4) return;
    const int r = (int)(i / cols4), c = (int)(i - (long)r * cols4) * 4;
    float* o = out + (long)r * ld_out + c;
    f32x4 acc = accumulate ? *(const f32x4*)o : f32x4{0.f, 0.f, 0.f, 0.f};
    const float* src = part + (long)r * ld_part + c;
    for (int s = 0; s < slices; ++s) {
#ifdef DW_NT_REDUCE      // (experiment: the slabs are read once)
        const f32x4 v = __builtin_nontemporal_load((const f32x4*)(src + s * slice_stride));
#else
        const f32x4 v = *(const f32x4*)(src + s * slice_stride);
#endif
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += v[e];
    }
    *(f32x4*)o = acc;
}
extern "C" int dw_reduce_slices_ld(const float* part, int64_t slice_stride, int64_t ld_part, int slices, float* out, int64_t ld_out,
                                   int rows, int cols, int accumulate, void* stream) {
    DW_CLEAR_ERR();
    if (!part || !out || slices < 1 || rows <= 0 || cols <= 0 || (cols & 3) || (ld_part & 3) || (ld_out & 3) || (slice_stride & 3) ||
        ld_part < cols || ld_out < cols || ((uintptr_t)part & 15) || ((uintptr_t)out & 15))
        return DW_EINVAL;
    const long n = (long)rows * (cols / 4);
    hipLaunchKernelGGL(reduce_slices_ld_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, part, (long)slice_stride,
                       (long)ld_part, slices, out, (long)ld_out, rows, cols / 4, accumulate);
    DW_CHECK_LAUNCH();
    return DW_OK;
}

static inline bool q_split_ok(const GemmP& p) { return p.split_k == 1 && !p.atomic; }
extern "C" int dw_gemm_bf16(const DwGemm* g, void* stream) {
    DW_CLEAR_ERR();
    if (!g || (!g->a && !g->ln_x) || !g->b || !g->c) return DW_EINVAL;
    if (g->m <= 0 || g->n <= 0 || g->k <= 0 || (g->k & 63)) return DW_EINVAL;
    if ((g->lda & 7) || (g->ldb & 7)) return DW_EINVAL;
    if ((g->a && ((uintptr_t)g->a & 15)) || ((uintptr_t)g->b & 15)) return DW_EINVAL;
    // k-major operands are fetched in 16-byte column slots: a slot whose first column is valid is read whole, so the
    // row must be readable up to the next multiple of 8 columns (true whenever ld covers the padded width)
    if (g->trans_a && g->lda < ((g->m + 7) & ~7)) return DW_EINVAL;
    if (g->trans_b && g->ldb < ((g->n + 7) & ~7)) return DW_EINVAL;
    GemmP p;
    p.a = (const bf16*)g->a; p.b = (const bf16*)g->b; p.c = g->c; p.bias = g->bias;
    p.z_out = (bf16*)g->z_out; p.zgrad = (const bf16*)g->zgrad_in; p.r = g->r;
    p.lda = g->lda; p.ldb = g->ldb; p.ldc = g->ldc; p.ldz = g->ldz; p.ldzg = g->ldzg; p.ldr = g->ldr;
    p.m = g->m; p.n = g->n; p.k = g->k;
    p.act = g->act; p.c_dtype = g->c_dtype; p.r_dtype = g->r_dtype; p.r_row_mod = g->r_row_mod;
    p.round_res = g->round_res; p.tiles_n = 0; p.nwg = 0;
    p.split_k = g->split_k > 1 ? g->split_k : 1;
    p.atomic = g->atomic_acc ? 1 : 0;
    if (p.split_k > (g->k >> 6)) p.split_k = g->k >> 6;
    p.slice_stride = 0;
    p.sched = nullptr;
    p.zg_f16 = g->z_is_gelu_grad ? 1 : 0;
    p.colsum = g->colsum_out;
    if (g->colsum_out && (g->split_k > 1 || g->atomic_acc || ((uintptr_t)g->colsum_out & 3))) return DW_EINVAL;
    p.stage_next = g_gemm_stage_next;
    p.stagger = g_gemm_stagger;
    p.trace = (long long*)(((unsigned long long)g_gemm_trace_hi << 32) | g_gemm_trace_lo);
    if (p.zg_f16 && g->z_out && g->act != 1) return DW_EINVAL;   // gelu'(z) is a by-product of the GELU epilogue
    p.ln_x = g->ln_x; p.ln_g = g->ln_gamma; p.ln_b = g->ln_beta; p.ld_lnx = g->ld_lnx; p.ln_x_dtype = g->ln_x_dtype;
    p.ln_eps = g->ln_eps;
    p.kv_out = (bf16*)g->kv_out; p.kv_ld = g->kv_ld; p.kv_split = g->kv_split; p.kv_rpb = g->kv_rows_per_batch;
    p.kv_pitch = g->kv_batch_pitch; p.kv_row0 = g->kv_row0;
    const bool fused = g->ln_x || g->kv_out;
    if (g->ln_x && (!g->ln_gamma || !g->ln_beta || g->k > 1280 || g->m > 32 || (g->ld_lnx & 7) ||
                    ((uintptr_t)g->ln_x & 15) || ((uintptr_t)g->ln_gamma & 15) || ((uintptr_t)g->ln_beta & 15)))
        return DW_EINVAL;
    if (g->kv_out && (g->c_dtype != DW_BF16 || g->act || g->r || g->z_out || g->zgrad_in || (g->kv_split & 15) ||
                      g->kv_split <= 0 || g->kv_split >= g->n || g->kv_rows_per_batch <= 0 || (g->kv_ld & 3) ||
                      ((uintptr_t)g->kv_out & 7)))
        return DW_EINVAL;
    if (p.split_k > 1 && !p.atomic) {
        // K slices without atomics: every slice stores a plain fp32 partial at c + ks * slice_stride (the caller
        // reduces them, see dw_reduce_slices); only the bare epilogue makes sense here
        if (g->c_dtype != DW_F32 || g->bias || g->z_out || g->zgrad_in || g->r || g->act || g->slice_stride <= 0)
            return DW_EINVAL;
        p.slice_stride = g->slice_stride;
    }
    if (p.atomic && (g->c_dtype != DW_F32 || g->bias || g->z_out || g->zgrad_in || g->r || g->act)) return DW_EINVAL;
    {
        const int es = g->c_dtype == DW_F32 ? 4 : 2;
        bool v = ((uintptr_t)g->c % (4 * es)) == 0 && (g->ldc & 3) == 0;
        if (g->bias) v = v && ((uintptr_t)g->bias & 15) == 0;
        if (g->z_out) v = v && ((uintptr_t)g->z_out & 7) == 0 && (g->ldz & 3) == 0;
        if (g->zgrad_in) v = v && ((uintptr_t)g->zgrad_in & 7) == 0 && (g->ldzg & 3) == 0;
        if (g->r) v = v && ((uintptr_t)g->r % (g->r_dtype == DW_F32 ? 16 : 8)) == 0 && (g->ldr & 3) == 0;
        p.vec = v ? 1 : 0;
    }
    int tile = g->tile;
    hipStream_t s = (hipStream_t)stream;
    // decode regime (M = batch rows): weight-streaming kernel; tile = 16 requests it explicitly
    if (tile == 16 && !dw_gemm_skinny_ok(p, g->trans_a, g->trans_b)) return DW_EINVAL;
    if ((tile == 0 || tile == 16) && dw_gemm_skinny_ok(p, g->trans_a, g->trans_b)) {
        if (g->colsum_out) return DW_EINVAL;      // (column sums are a tile-kernel epilogue)
        return dw_gemm_skinny_launch(p, s);
    }
    if (fused) return DW_EINVAL;                  // the fusions exist in the skinny-M kernel only
    if (tile == 129 && (g->trans_a || p.split_k != 1 || p.atomic)) return DW_EINVAL;   // (tile 129: force the software-pipelined 128 x 256 tile)
    bool one_round_320 = false;
    bool small_m_256 = false;       // a 256-row kernel chosen by the small-M rule below: runs on the 16x16x32 loop
    if (tile != 128 && tile != 256 && !(tile == 129 && g_gemm_small_m == 2)) {     // (key 25 = 2: tile 129 forces the 128-row kernel at any size -- probing)
        const long tn = (g->n + 255) / 256;
        const long t256 = (long)((g->m + 255) / 256) * tn;
        tile = t256 >= 512 ? 256 : 128;
        if (tile == 128 && g_gemm_small_m && (g_gemm_variant & 16) && !g->trans_a && p.split_k == 1 && !p.atomic) {
            // Fewer than two rounds of 256-row tiles (the decoders: M = 32 x live positions).  Three software-pipelined kernels can
            // take the launch -- 128 x 256 tiles in a three-stage ring (gemm_wp8_m128.hip), 256 x 256 on 16x16x32, 320 x 256 -- and
            // what decides is how many rounds of the 256 CUs each needs: a 256-row tile costs ~1.65 and a 320-row tile ~2.05 of a
            // 128-row tile's time (tools/gemm_m128_probe.py: M = 4 480 / 4 258 / 7 136 at the step's N and K; e.g. N = K = 1280 at
            // M = 4 480: 21.3 us on 175 128-row tiles against 31.7 on 90 256-row tiles and 26.6 for the lock-step 128 x 128 kernel
            // this rule replaces; N = 1280, K = 5120 at M = 7 136: 108 us on 140 256-row tiles against 123 / 136).  Every kernel
            // computes the same fp32 chain over k per element: the choice never changes a result.
            const long cus = g_gemm_cus;
            const long r128 = (((long)(g->m + 127) / 128) * tn + cus - 1) / cus;
            const long r256 = (t256 + cus - 1) / cus;
            const bool ok320 = (g_gemm_variant & 2048) && g->m % 320 == 0 && g->n % 256 == 0;
            const long r320 = ok320 ? ((long)(g->m / 320) * tn + cus - 1) / cus : 0;
            const long c128 = r128 * 100, c256 = r256 * 165, c320 = ok320 ? r320 * 205 : (1L << 40);
            if (c320 <= c128 && c320 <= c256) { tile = 256; one_round_320 = true; }
            else if (c256 <= c128) { tile = 256; small_m_256 = true; }
            else tile = 129;
        }
    }
    if (tile == 129) {
        p.strip = g_gemm_strip;
        if (g_gemm_dynamic) p.sched = gemm_sched_slot(s);
        return g->trans_b ? dw_gemm_wp8_nt128_launch(p, s) : dw_gemm_wp8_nn128_launch(p, s);
    }
    if (tile == 256 && g->tile != 256 && !g->trans_a && !g->trans_b && g->r && g->r_dtype == DW_F32 && g->c_dtype == DW_F32 &&
        g->k <= 1024 && p.split_k == 1 && !p.atomic) {
        // Short K with the fp32 residual read + fp32 store (small.en's out-proj: N = K = 768): 655 KB of epilogue traffic per
        // 320 x 256 tile against 12 K tiles of main loop -- the epilogue is the kernel.  Two 128-tile workgroups per CU run
        // one's epilogue under the other's K loop: 504 vs 464 TFLOP/s at M = 48 000 (tools/gemm_small_en_probe.py).
        tile = 128;
    }
    if (tile == 256) {
        // K <= 1024 (D = 768 models): the next tile's first operand tile is requested before the epilogue (gemm_wp.h; neutral
        // at K = 1280, +1..2 % over 12 K tiles: qkv 915 -> 933, fc1 738 -> 750 TFLOP/s)
        if (g->k <= 1024) p.stage_next |= 128;
        // the software-pipelined kernels address their operand DMA with 31-bit buffer offsets
        const long spanA = g->trans_a ? (long)g->k * g->lda * 2 : 256L * g->lda * 2 + (long)g->k * 2;
        const long spanB = g->trans_b ? (long)g->k * g->ldb * 2 : 256L * g->ldb * 2 + (long)g->k * 2;
        const bool wp_ok = spanA < 0x7fffffffL && spanB < 0x7fffffffL;
        const int v = g_gemm_variant;
        p.strip = g_gemm_strip;
        if (g_gemm_dynamic) p.sched = gemm_sched_slot(s);
        // 320-row tiles (row-major A, variant bit 2048; gemm_wp.h): M % 320 == 0, N % 256 == 0, and no more rounds of the
        // CUs than 256-row tiles need, counted in 256-row-tile times (a 320-row tile costs 1.25).  Measured against the
        // 256-row kernels on M = 48000 (tools/bench_gemm_variants.py): +7..8 % for K >= 3840 on both operand layouts
        // (also against the phased kernel), +2..6 % for row-major B with N >= 2560 at K = 1280, -2 % (NN) / -7 % (NT) at
        // N = K = 1280 and -3 % for k-major B at N = 5120, K = 1280: those keep the 256-row tile.  Bit 4096 forces the
        // 320-row tile wherever it is eligible (experiments).
        bool use320 = false;
        if ((v & 2048) && wp_ok && !g->trans_a && q_split_ok(p) && g->m % 320 == 0 && g->n % 256 == 0) {
            const long tn = g->n / 256;
            const long r256 = (((g->m + 255) / 256) * tn + g_gemm_cus - 1) / g_gemm_cus * 4;
            const long r320 = ((g->m / 320) * tn + g_gemm_cus - 1) / g_gemm_cus * 5;
            const bool epi_bound = g->r && g->r_dtype == DW_F32 && g->c_dtype == DW_F32 && g->k <= 2560;
            // (round 3: with one epilogue walk per flavour the 320-row kernel also wins at N = K = 1280 -- out-proj with the
            // fp32 residual 262 -> 227 us (it used to go to the 16-wave kernel), with the bf16 residual 216 -> 195, dX of
            // out-proj 178 -> 165 -- so the K / N / epilogue conditions of round 2 are gone (bit 8192 restores them);
            // the x gelu'(z) epilogue has no flavoured walk in the 320-row kernel and keeps the 256-row tile)
            const bool r2rule = (v & 8192) != 0;
            if ((v & 4096) || one_round_320) use320 = true;
            else if (!g->trans_b) use320 = r320 <= r256 && (!r2rule || (!epi_bound && (g->k >= 2560 || g->n >= 2560)));
            else use320 = r320 < r256 && (r2rule ? g->k >= 2560 : !g->zgrad_in);
        }
        // Row-major, short K, wide N (QKV and fc1 forward): the 256-row tile on v_mfma_f32_16x16x32_bf16 (dw_debug_set key 20 bit 32)
        const bool nn16_256 = (g_gemm_mi16 & 32) && (v & 16) && wp_ok && !g->trans_a && !g->trans_b && g->k <= 2560 && g->n >= 3840 &&
                              !g->r && !g->z_out && !g->zgrad_in && g->c_dtype != DW_F32 &&    // (the flavours of the accumulator-side walk)
                              !one_round_320 && !(v & 4096) && !g_gemm_dbg;   // (a forced / single-round 320-row tile and the ablation path keep their kernel)
        auto launch256 = [&](const GemmP& q) -> int {
            if (small_m_256 && wp_ok && !g_gemm_dbg) return g->trans_b ? dw_gemm_wp16_nt_small_launch(q, s) : dw_gemm_wp16_nn_small_launch(q, s);
            if (nn16_256) {
                // Row tail (round 5; dw_debug_set key 22, default on): M = 48 000 is 187.5 row tiles -- 2 820 tiles = 11.02 rounds of the
                // CUs, and with the per-XCD job ranges four XCDs run a TWELFTH round for four tiles (8 % of the launch).  The last,
                // partial row block (<= 128 rows: 30 small tiles) goes to the 128-tile kernel when the full row blocks alone need a
                // round less: 2 805 tiles = 10.96 rounds + a ~15 us launch instead of 12 rounds.  (Splitting a whole partial ROUND off was
                // measured 8 ms slower in round 2 -- the second launch waits for the slowest workgroup of the first; this is a 30-tile
                // launch.)  Same arithmetic per output element in both kernels: bit-identical to the single launch.
                const int tm_full = q.m / 256, rem = q.m - tm_full * 256;
                const long tn = (q.n + 255) / 256;
                const long rounds_all = ((long)(tm_full + 1) * tn + g_gemm_cus - 1) / g_gemm_cus;
                const long rounds_full = ((long)tm_full * tn + g_gemm_cus - 1) / g_gemm_cus;
                if (g_gemm_row_tail && g->tile != 256 && rem > 0 && rem <= 128 && tm_full > 0 && rounds_full < rounds_all && !q.colsum) {
                    GemmP q1 = q, q2 = q;
                    q1.m = tm_full * 256;
                    const size_t es = q.c_dtype == DW_F32 ? 4 : 2;
                    q2.m = rem;
                    q2.a = q.a + (long)q1.m * q.lda;
                    q2.c = (char*)q.c + (size_t)q1.m * q.ldc * es;
                    q2.sched = nullptr;
                    const int rc1 = dw_gemm_wp16_nn_launch(q1, s);
                    if (rc1 != DW_OK) return rc1;
                    return dw_gemm_tile128_launch(q2, 0, 0, s);
                }
                return dw_gemm_wp16_nn_launch(q, s);
            }
            if (use320) {
                if (g->trans_b) return (g_gemm_mi16 & 2) ? dw_gemm_wp16_nt320_launch(q, s) : dw_gemm_wp8_nt320_launch(q, s);
                return (g_gemm_mi16 & 1) ? dw_gemm_wp16_nn320_launch(q, s) : dw_gemm_wp8_nn320_launch(q, s);
            }
            if (!g->trans_a && !g->trans_b) {
                // (short-K GEMMs with an fp32 residual and fp32 output are epilogue / HBM bound -- 615 MB per launch at
                // K = 1280 -- and the 16-wave kernel's four waves per SIMD overlap that better: 229 vs 256 us in the step)
                const bool epi_bound = g->r && g->r_dtype == DW_F32 && g->c_dtype == DW_F32 && g->k <= 2560;
                if ((v & 16) && wp_ok && !epi_bound && (g_gemm_mi16 & 8) && g_gemm_dbg) return dw_gemm_wp16_nn_w4_dbg_launch(q, g_gemm_dbg, s);
                if ((v & 16) && wp_ok && !epi_bound && (g_gemm_mi16 & 8)) return dw_gemm_wp16_nn_w4_launch(q, s);
                if ((v & 16) && wp_ok && !epi_bound && g_gemm_dbg) return dw_gemm_wp8_nn_dbg_launch(q, g_gemm_dbg, s);
                if ((v & 16) && wp_ok && !epi_bound && (v & 1536)) return dw_gemm_wp8_nn_dbg_launch(q, (v >> 9) & 3, s);
                if ((v & 16) && wp_ok && !epi_bound && (g_gemm_mi16 & 1) && g_gemm_dbg) return dw_gemm_wp16_nn_dbg_launch(q, g_gemm_dbg, s);
                if ((v & 16) && wp_ok && !epi_bound && (g_gemm_mi16 & 1)) return dw_gemm_wp16_nn_launch(q, s);
                if ((v & 16) && wp_ok && !epi_bound) return (v & 256) ? dw_gemm_wp8_nn_ref_launch(q, s) : dw_gemm_wp8_nn_launch(q, s);
            } else if (!g->trans_a && g->trans_b) {
                if (((v & 4) && g->k >= 3840 && q.split_k == 1) || (v & 128)) return dw_gemm_phased_launch(q, 0, 1, s);
                if ((v & 32) && wp_ok && (g_gemm_mi16 & 16)) return dw_gemm_wp16_nt_w4_launch(q, s);
                if ((v & 32) && wp_ok) return (g_gemm_mi16 & 2) ? dw_gemm_wp16_nt_launch(q, s) : dw_gemm_wp8_nt_launch(q, s);
            } else if (g->trans_a && g->trans_b) {
                if ((v & 64) && wp_ok) return (g_gemm_mi16 & 4) ? dw_gemm_wp16_tt_launch(q, s) : dw_gemm_wp8_tt_launch(q, s);
            }
            return dw_gemm_tile256_launch(q, g->trans_a, g->trans_b, s);
        };
        // (Measured and dropped: splitting the rows of a mostly empty last round off to the 128-tile kernel -- a launch
        // costs ceil(tiles / 256) tile times, 940 tiles pay for 1024 -- made the step 8 ms SLOWER (444 -> 452 ms): the
        // second launch cannot start before the slowest workgroup of the first one has drained.)
        return launch256(p);
    }
    if (g_gemm_t128_w4) return dw_gemm_tile128w4_launch(p, g->trans_a, g->trans_b, g_gemm_t128_w4, s);
    return dw_gemm_tile128_launch(p, g->trans_a, g->trans_b, s);
}
