// Shared device helpers for the gfx950 (CDNA4 / MI355X) kernels of the Whisper-distillation hot path.
// Wavefront = 64 lanes everywhere; MFMA tiles are v_mfma_f32_32x32x16_bf16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) void glb_void_t;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_t;

#define DW_WAVE 64

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E) -- guarantees static register indexing
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// ---- bf16 <-> f32 (round-to-nearest-even, NaN preserved) -------------------------------------------------
__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }
__device__ __forceinline__ float round_bf16(float x) { return (float)((bf16)x); }
__device__ __forceinline__ float bits2f(unsigned short u) {
    return __uint_as_float(((unsigned)u) << 16);
}

// ---- exact (erf) GELU, as transformers' GELUActivation / F.gelu(approximate='none') ------------------------
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// Fast exact-erf GELU for the GEMM epilogues (Abramowitz & Stegun 7.1.26, |erf error| <= 1.5e-7: far below the
// bf16 rounding applied to every GELU output).  Written on 2-vectors so that hipcc emits packed fp32 VALU ops
// (v_pk_fma_f32 / v_pk_mul_f32: two elements per instruction); one v_exp and one v_rcp per element, shared between
// the cdf and the pdf (exp(-x^2/2) is both the erf tail factor at x/sqrt2 and the Gaussian density).
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ void gelu_parts2(f32x2 x, f32x2& cdf, f32x2& pdf) {
    f32x2 ax; ax[0] = fabsf(x[0]); ax[1] = fabsf(x[1]);
    const f32x2 u = ax * 0.70710678118654752f;                 // |x| / sqrt(2)
    f32x2 den = u * 0.3275911f + 1.0f;
    // raw v_rcp_f32 / v_exp_f32 (1 ulp, no IEEE fix-up sequences: the result is rounded to bf16 anyway)
    f32x2 t; t[0] = __builtin_amdgcn_rcpf(den[0]); t[1] = __builtin_amdgcn_rcpf(den[1]);
    const f32x2 xx = x * x * -0.72134752044448170f;            // -0.5 * log2(e) * x^2
    f32x2 e; e[0] = __builtin_amdgcn_exp2f(xx[0]); e[1] = __builtin_amdgcn_exp2f(xx[1]);
    f32x2 y = t * 1.061405429f + (-1.453152027f);
    y = y * t + 1.421413741f;
    y = y * t + (-0.284496736f);
    y = y * t + 0.254829592f;
    const f32x2 tail = y * t * e;                               // 1 - erf(|x|/sqrt2)
    f32x2 half_tail = tail * 0.5f;                              // cdf = 1 - tail/2 (x >= 0), tail/2 (x < 0)
    cdf[0] = x[0] >= 0.f ? 1.0f - half_tail[0] : half_tail[0];
    cdf[1] = x[1] >= 0.f ? 1.0f - half_tail[1] : half_tail[1];
    pdf = e * 0.39894228040143268f;
}
// The same approximation arranged for the epilogues (round 5: 32 -> 26 vector instructions per pair in the fc1 walks): with
// s = 1 - tail = erf(|x| / sqrt2),  gelu(x) = x * cdf(x) = (x + |x| * s) / 2 for either sign of x -- no compare / select pair per
// element and no separate multiply by x; cdf itself (gelu' = cdf + x * pdf, the student's stored by-product) is
// 1/2 + copysign(s, x) / 2: one bit-field insert per element.
__device__ __forceinline__ void gelu_y_s_pdf2(f32x2 x, f32x2& y, f32x2& s, f32x2& pdf) {
    f32x2 ax; ax[0] = fabsf(x[0]); ax[1] = fabsf(x[1]);
    const f32x2 u = ax * 0.70710678118654752f;
    f32x2 den = u * 0.3275911f + 1.0f;
    f32x2 t; t[0] = __builtin_amdgcn_rcpf(den[0]); t[1] = __builtin_amdgcn_rcpf(den[1]);
    const f32x2 xx = x * x * -0.72134752044448170f;
    f32x2 e; e[0] = __builtin_amdgcn_exp2f(xx[0]); e[1] = __builtin_amdgcn_exp2f(xx[1]);
    f32x2 yy = t * 1.061405429f + (-1.453152027f);
    yy = yy * t + 1.421413741f;
    yy = yy * t + (-0.284496736f);
    yy = yy * t + 0.254829592f;
    const f32x2 tail = yy * t * e;
    s = 1.0f - tail;
    y = (ax * s + x) * 0.5f;
    pdf = e * 0.39894228040143268f;
}
__device__ __forceinline__ f32x2 gelu_cdf_from_s2(f32x2 x, f32x2 s) {
    f32x2 c;
    c[0] = __builtin_copysignf(s[0], x[0]); c[1] = __builtin_copysignf(s[1], x[1]);
    return c * 0.5f + 0.5f;
}
// two fp32 values rounded to bf16 and widened again, through ONE packed conversion (the scalar form is a conversion and a
// shift per value)
__device__ __forceinline__ f32x2 round_bf16_pair(f32x2 v) {
    const bf16x2 r = __builtin_convertvector(v, bf16x2);
    const unsigned u = __builtin_bit_cast(unsigned, r);
    f32x2 o; o[0] = __uint_as_float(u << 16); o[1] = __uint_as_float(u & 0xffff0000u);
    return o;
}
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
    f32x2 y, s, pdf;
    gelu_y_s_pdf2(x, y, s, pdf);
    return y;
}
__device__ __forceinline__ f32x2 gelu_grad_fast2(f32x2 x) {
    f32x2 y, s, pdf;
    gelu_y_s_pdf2(x, y, s, pdf);
    return gelu_cdf_from_s2(x, s) + x * pdf;
}
__device__ __forceinline__ float gelu_fast(float x) { f32x2 v; v[0] = x; v[1] = x; return gelu_fast2(v)[0]; }
__device__ __forceinline__ float gelu_grad_fast(float x) { f32x2 v; v[0] = x; v[1] = x; return gelu_grad_fast2(v)[0]; }

// ---- LDS tile swizzle for [rows][64 bf16] (=128-byte rows) tiles ------------------------------------------------
// 16-byte slot s (0..7) of row r is stored at physical slot s ^ swz7(r).  The permutation of row bits
// (bit1->bit2, bit3->bit1, bit2->bit0) makes BOTH access patterns conflict free on gfx950:
//   * ds_read_b128 of a 32x32x16 MFMA operand fragment (lane -> row lane&31, k-slot 2*kk+(lane>>5)), and
//   * ds_read_b64_tr_b16 of 4 consecutive rows x 64 bytes per 32-lane half (transposed operand fragments).
__device__ __forceinline__ int swz7(int row) {
    return (((row >> 1) & 1) << 2) | (((row >> 3) & 1) << 1) | ((row >> 2) & 1);
}

// Direct global->LDS load (global_load_lds_dwordx4 / _dword): the LDS destination is (wave-uniform base in M0) + lane * 16
// (or * 4); the global source address is per lane.
// Inline assembly ON PURPOSE: issued through the builtin, the compiler treats the instruction as an LDS store it cannot
// disambiguate from reads of the OTHER staging buffer and puts `s_waitcnt vmcnt(0)` in front of the next ds_read --
// i.e. it waits for the prefetch it has just issued, every tile.  Hidden from its memory model, the only vmcnt waits are
// the explicit wait_vm0() calls in front of the barriers (every user has one).  The compiler's own vmcnt bookkeeping
// for ordinary loads stays correct: the counter retires in order, so unseen operations in flight can only make its
// waits longer, never shorter.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lds_void_t*)lds_wave_base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(m0v), "v"(gsrc) : "memory");
}
__device__ __forceinline__ void glds4(const void* gsrc, void* lds_wave_base) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lds_void_t*)lds_wave_base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" :: "s"(m0v), "v"(gsrc) : "memory");
}

__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// The same wait as an instruction the COMPILER sees (s_waitcnt vmcnt(0), expcnt / lgkmcnt untouched).  Needed where ordinary
// global loads issued in front of a loop (the stationary operand's fragments) are first used inside it: the compiler's own
// counted waits for them (vmcnt(3) .. vmcnt(0) in front of the first MFMAs) stay in the loop body for every iteration, where
// they also wait for the tile prefetch the asm above has just issued -- the wave sat out the prefetch latency once per tile
// (round 5, found in the generated code of attn_fwd_kernel / attn_bwd_dkv_kernel).  Behind this instruction the compiler knows
// that nothing of its own is in flight and emits none.
__device__ __forceinline__ void wait_vm0_seen() {
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
}
// max / sum across the two 32-lane halves of a wave through v_permlane32_swap (one vector instruction; __shfl_xor(v, 32) is
// a ds_bpermute: an LDS round trip with an lgkmcnt wait on the softmax's serial path)
__device__ __forceinline__ float xhalf_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Fragment of a row-major [rows][64] swizzled tile for v_mfma_f32_32x32x16_bf16:
// lane holds X[rb*32 + (lane&31)][kk*16 + 8*(lane>>5) + 0..7].
__device__ __forceinline__ bf16x8 frag_rows(const char* tile, int rb, int kk, int lane) {
    const int row = rb * 32 + (lane & 31);
    const int ps = ((kk << 1) | (lane >> 5)) ^ swz7(row);
    return *(const bf16x8*)(tile + row * 128 + ps * 16);
}

// Transposed fragment of the same tile via ds_read_b64_tr_b16: lane (g = lane>>4, i = lane&15) receives
// X[rowbase(rd) + j][cb*32 + 16*(g&1) + i] for j = 0..3 in elements 4*rd + j, where
// rowbase(rd) = row0 + 8*rd_stride_sel ... (callers pass the two 4-row bases explicitly).
// Hardware semantics used: within each 16-lane group, lane p supplies the address of 4 contiguous bf16;
// output lane i, element j  =  element (i&3) of the 8 bytes supplied by lane 4*j + (i>>2).
__device__ __forceinline__ bf16x8 frag_tr(const char* tile, int cb, int rowbase0, int rowbase1, int lane) {
    const int g = lane >> 4, p = lane & 15;
    const int col = cb * 32 + ((g & 1) << 4) + ((p & 3) << 2);
    const int r0 = rowbase0 + (p >> 2);
    const int r1 = rowbase1 + (p >> 2);
    const int ls = col >> 3;
    const int inb = (p & 1) << 3;
    const char* a0 = tile + r0 * 128 + ((ls ^ swz7(r0)) << 4) + inb;
    const char* a1 = tile + r1 * 128 + ((ls ^ swz7(r1)) << 4) + inb;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)a0);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)a1);
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// Stage a [64 rows][64 bf16] tile (8 KiB) with NW waves: 8 chunks of 1 KiB (= 8 rows each).
// Rows past nrows are clamped to the last valid row (callers mask the results).
// Full tiles take the cheap path: a 32-bit per-lane element offset (its lane part is loop invariant, its row0 part
// is scalar) on top of the uniform base -- the generic 64-bit row * stride product cost ~10 VALU instructions per
// load in kernels whose VALU is the bottleneck (attention).  Callers guarantee nrows * row_stride < 2^31.
// Addressing: the tile's base pointer is wave-uniform (scalar registers), the lane part is ONE unsigned 32-bit byte offset and the
// LDS destination is an integer computed once per call -- `global_load_lds_dwordx4 voffset, s[base]` with M0 = lds + chunk * 1 KiB.
// (Through a per-lane 64-bit pointer and a generic LDS pointer every chunk cost a 64-bit multiply-add, a 64-bit shift-add and
// the generic -> LDS conversion with its null check -- ~6 vector and ~8 scalar instructions per chunk in kernels whose vector
// unit is the bottleneck -- and two 64-bit lane offsets per operand; the dK/dV kernel spilled one and reloaded it inside its loop
// behind a vmcnt(0) that also waited for the chunk just requested.)
__device__ __forceinline__ void glds16_so(const void* sbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(lds_addr), "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void glds4_so(const void* sbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" :: "s"(lds_addr), "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* lds) {       // wave-uniform LDS byte address of a __shared__ object
    return __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lds_void_t*)lds);
}
template <int NW, bool FAST = true>
__device__ __forceinline__ void stage_tile64(const bf16* base, long row_stride, int row0, int nrows, char* lds,
                                             int wave, int lane) {
    static_assert(NW == 4 || NW == 8 || NW == 12 || NW == 2, "chunk c = wave + i * NW: rows 8 c + (lane >> 3)");
    const int ld = (int)row_stride;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lds_void_t*)lds);
    if (row0 + 64 <= nrows) {
        // Full tile (every tile but the last): chunk i of a wave differs from its chunk 0 by i * NW * 8 rows -- the swizzle only
        // looks at row bits 1-3 -- so ONE loop-invariant lane offset serves all of them and the rest is scalar.
        const int row = wave * 8 + (lane >> 3);
        const unsigned voff = (unsigned)(row * ld + (((lane & 7) ^ swz7(row)) << 3)) * 2u;
#pragma unroll
        for (int c = wave; c < 8; c += NW) {
            const bf16* tb = base + (long)__builtin_amdgcn_readfirstlane((row0 + (c - wave) * 8) * ld);
            glds16_so(tb, voff, lds0 + c * 1024);
        }
    } else {
#pragma unroll
        for (int c = wave; c < 8; c += NW) {
            const int row = c * 8 + (lane >> 3);
            const int ls = (lane & 7) ^ swz7(row);
            int grow = row0 + row;
            grow = grow < nrows ? grow : nrows - 1;
            glds16_so(base, (unsigned)(grow * ld + ls * 8) * 2u, lds0 + c * 1024);
        }
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// Block-wide sum for blocks of NT threads (NT multiple of 64, <= 1024). `red` is >= NT/64 floats of LDS.
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += red[i];
    return t;
}
template <int NT>
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = red[0];
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) t = fmaxf(t, red[i]);
    return t;
}

// hipGetLastError() reports the last error of ANY earlier HIP call on this thread (including benign probes made by
// the host framework), so every launcher first clears it and then checks its own launches.
#define DW_CLEAR_ERR() (void)hipGetLastError()
#define DW_CHECK_LAUNCH()                                  \
    do {                                                   \
        hipError_t e__ = hipGetLastError();                \
        if (e__ != hipSuccess) return (int)e__;            \
    } while (0)

// Streamed-once loads of the token step (-DDW_DECODE_NT=<mask>): bit 0 the weight-streaming GEMVs' W, bit 1 the K / V rows of the
// single-query attention kernels.  Round 6, measured in one process against variant builds (tools/decode_nt_ab.py): the cross-
// attention K / V are 246 MB per token step at batch 16, read once by one workgroup each; with the non-temporal hint on them
// (mask 2, the default) they no longer push the step's RE-READ bytes -- 92 MB of layer weights + the 133 MB LM head, which fit
// the 256 MB Infinity Cache -- out of it between steps: 0.2490 -> 0.2288 ms per step of the 2-layer student (-8.1 %), 3.069 -> 3.022 ms
// for the 32-layer teacher (whose 1.7 GB of weights cannot be resident anyway).  The hint on the WEIGHT loads is the wrong way
// round for the same reason (mask 1: -1 % / +3.6 %; mask 3 cancels the gain).
#ifndef DW_DECODE_NT
#define DW_DECODE_NT 2
#endif
template <int BIT, class T>
__device__ __forceinline__ T ld_stream(const T* p) {
    if constexpr ((DW_DECODE_NT & BIT) != 0) return __builtin_nontemporal_load(p);
    else return *p;
}
