"""ctypes binding of libdwamd.so (C ABI in include/dwamd.h) over torch device tensors.

PyTorch is plumbing here: it owns device memory and the HIP stream; every compute call goes to a hand-written gfx950
kernel.  There is NO fallback: constructing `HipOps` without the shared library or without a GPU raises.
The method set is the "ops" interface the host engine (engine.py) is written against; tests inject a torch
restatement with the same interface (oracle/ref_ops.py) to check the host logic on CPU.
"""
import ctypes as C
import math
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdwamd.so")

DW_F32, DW_BF16 = 0, 1


class DwGemm(C.Structure):
    _fields_ = [
        ("a", C.c_void_p), ("b", C.c_void_p), ("c", C.c_void_p), ("bias", C.c_void_p), ("z_out", C.c_void_p),
        ("zgrad_in", C.c_void_p), ("r", C.c_void_p),
        ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64), ("ldz", C.c_int64), ("ldzg", C.c_int64),
        ("ldr", C.c_int64),
        ("m", C.c_int32), ("n", C.c_int32), ("k", C.c_int32),
        ("trans_a", C.c_int32), ("trans_b", C.c_int32), ("act", C.c_int32), ("c_dtype", C.c_int32),
        ("r_dtype", C.c_int32), ("r_row_mod", C.c_int32), ("round_res", C.c_int32), ("tile", C.c_int32),
        ("split_k", C.c_int32), ("atomic_acc", C.c_int32), ("slice_stride", C.c_int64),
        ("ln_x", C.c_void_p), ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("kv_out", C.c_void_p),
        ("ld_lnx", C.c_int64), ("kv_ld", C.c_int64),
        ("ln_x_dtype", C.c_int32), ("kv_split", C.c_int32), ("kv_rows_per_batch", C.c_int32),
        ("kv_batch_pitch", C.c_int32), ("kv_row0", C.c_int32), ("ln_eps", C.c_float),
        ("z_is_gelu_grad", C.c_int32), ("colsum_out", C.c_void_p),
    ]


class DwDecoderLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "ln1_g", "ln1_b", "wqkv", "bqkv", "wo", "bo", "ln2_g", "ln2_b", "wq", "bq", "wo2", "bo2", "ln3_g", "ln3_b",
        "w1", "b1", "w2", "b2", "self_kv", "cross_kv")]


class DwDecodeStep(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "batch", "n_new", "d_model", "heads", "ffn", "n_layers", "src_len", "max_len", "t", "vocab", "ldv",
        "stream_dtype", "cross_kv_ld")] + [(n, C.c_void_p) for n in (
            "ids", "tok_emb", "pos_emb", "lnf_g", "lnf_b", "lm_head", "layers", "x", "h", "qkv", "o", "a", "logits")]


_SIGS = {
    "dw_version": ([], C.c_int),
    "dw_decode_step": ([C.POINTER(DwDecodeStep), C.c_void_p], C.c_int),
    "dw_logmel": ([C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                   C.c_void_p], C.c_int),
    "dw_gemm_bf16": ([C.POINTER(DwGemm), C.c_void_p], C.c_int),
    "dw_layernorm_fwd": ([C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                          C.c_int, C.c_float, C.c_void_p], C.c_int),
    "dw_layernorm_bwd": ([C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p], C.c_int),
    "dw_reduce_slices_ld": ([C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p], C.c_int),
    "dw_layernorm_fwd_ld": ([C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                             C.c_int, C.c_float, C.c_int64, C.c_int64, C.c_void_p], C.c_int),
    "dw_layernorm_bwd_ld": ([C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_int64] * 4 + [C.c_void_p], C.c_int),
    "dw_attn_fwd": ([C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_int64] * 4 + [C.c_int, C.c_float, C.c_void_p], C.c_int),
    "dw_attn_fwd_ex": ([C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_int64] * 6 + [C.c_int, C.c_float, C.c_void_p], C.c_int),
    "dw_attn_fwd_varlen": ([C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_int64] * 4 + [C.c_void_p] * 4 + [C.c_int, C.c_int64, C.c_int, C.c_float,
                                                                                                           C.c_void_p], C.c_int),
    "dw_attn_bwd": ([C.c_void_p] * 10 + [C.c_int] * 4 + [C.c_int64] * 8 + [C.c_int, C.c_float, C.c_void_p], C.c_int),
    "dw_attn_bwd_ex": ([C.c_void_p] * 10 + [C.c_int] * 4 + [C.c_int64] * 8 + [C.c_int, C.c_float] + [C.c_void_p] * 3, C.c_int),
    "dw_distill_loss": ([C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_float,
                         C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                         C.c_void_p], C.c_int),
    "dw_distill_loss_w": ([C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_void_p,
                           C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p], C.c_int),
    "dw_embed_fwd": ([C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                      C.c_void_p], C.c_int),
    "dw_embed_bwd": ([C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p], C.c_int),
    "dw_im2col_mel": ([C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p], C.c_int),
    "dw_im2col_s2": ([C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p], C.c_int),
    "dw_col2im_s2_gelu_bwd": ([C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p], C.c_int),
    "dw_gelu_bwd": ([C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p], C.c_int),
    "dw_pack_conv_weight": ([C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p], C.c_int),
    "dw_unpack_conv_grad": ([C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p], C.c_int),
    "dw_cast_f32_bf16": ([C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p], C.c_int),
    "dw_cast_bf16_f32": ([C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p], C.c_int),
    "dw_colsum_bf16": ([C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p], C.c_int),
    "dw_add": ([C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_void_p], C.c_int),
    "dw_move_rows": ([C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p], C.c_int),
    "dw_sumsq_f32": ([C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p], C.c_int),
    "dw_adamw": ([C.c_void_p] * 5 + [C.c_int64, C.c_void_p] + [C.c_float] * 2 + [C.c_double] * 5 + [C.c_int, C.c_void_p], C.c_int),
    "dw_adam_tick": ([C.c_void_p, C.c_void_p, C.c_void_p], C.c_int),
    "dw_adamw_dev": ([C.c_void_p] * 5 + [C.c_int64, C.c_void_p] + [C.c_float] * 2 + [C.c_void_p] + [C.c_double] * 2 +
                     [C.c_void_p], C.c_int),
    "dw_selftest_tr16": ([C.c_void_p, C.c_void_p], C.c_int),
    "dw_debug_set": ([C.c_int, C.c_int], C.c_int),
    "dw_reduce_slices": ([C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p], C.c_int),
    "dw_greedy_select": ([C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p] + [C.c_int] * 5 +
                         [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                          C.c_void_p], C.c_int),
}

EXPORTED_SYMBOLS = tuple(_SIGS.keys())


def load_library(path: str = LIB_PATH):
    """dlopen libdwamd.so and attach the prototypes.  Raises if the library is missing (no fallback path)."""
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build it with `python -m distil_whisper_amd.build` (hipcc --offload-arch=gfx950). "
            "The MI355X path has no CPU fallback.")
    lib = C.CDLL(path)
    for name, (args, res) in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    return lib


def _dt(t):
    if t.dtype == torch.float32:
        return DW_F32
    if t.dtype == torch.bfloat16:
        return DW_BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


def _p(t):
    return None if t is None else t.data_ptr()


class HipOps:
    """MI355X kernels behind the ops interface.  All tensors must live on the same cuda (HIP) device."""

    name = "hip"
    lowp = torch.bfloat16  # dtype of GEMM/attention operands and saved activations

    def __init__(self, device="cuda:0"):
        if not torch.cuda.is_available():
            raise RuntimeError("HipOps needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU path.")
        self.lib = load_library()
        self.device = torch.device(device)
        i = torch.arange(400, dtype=torch.float64)
        ang = 2.0 * math.pi * i / 400.0
        self._twiddle = torch.stack([torch.cos(ang), torch.sin(ang)], 1).to(torch.float32).contiguous().to(self.device)
        self._window = torch.hann_window(400, periodic=True, dtype=torch.float64).to(torch.float32).to(self.device)
        self.profile = None       # set to {} to bracket every launch with HIP events on the launch stream
        self._prof_events = []
        self.profile_detail = False   # keys of the GEMM entries carry shape + epilogue flavour (tools/step_breakdown.py)

    # ------------------------------------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _chk(rc, what):
        if rc != 0:
            raise RuntimeError(f"libdwamd: {what} failed with code {rc}")

    # ---- optional per-launch timing (bench.py roofline leg): HIP events on the stream the kernels are launched on --
    def _t0(self):
        if self.profile is None:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream(self.device))
        return e

    def _t1(self, e0, key, flops=0.0, nbytes=0.0):
        if e0 is None:
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record(torch.cuda.current_stream(self.device))
        self._prof_events.append((key, flops, nbytes, e0, e1))

    def collect_profile(self):
        torch.cuda.synchronize(self.device)
        out = {}
        for key, flops, nbytes, e0, e1 in self._prof_events:
            d = out.setdefault(key, {"n": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["n"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += flops
            d["bytes"] += nbytes
        self._prof_events = []
        return out

    @staticmethod
    def pick_tile(M, N):
        """256x256 block tiles once there are at least two full rounds of them over the 256 CUs, else 128x128."""
        t256 = ((M + 255) // 256) * ((N + 255) // 256)
        return 256 if t256 >= 512 else 128

    wgrad_slab_pad = 0      # floats of row pad of the split-K partial slabs of the weight-gradient GEMMs (32 measured neutral: the slab stores are not what those launches wait for)

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def zeros(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype, device=self.device)

    # ------------------------------------------------------------------------------------------------------------
    def selftest_tr16(self):
        out = self.empty((64, 4), torch.int32)
        self._chk(self.lib.dw_selftest_tr16(_p(out), self._stream()), "selftest_tr16")
        return out

    def logmel(self, audio, filters):
        assert audio.dtype == torch.float32 and audio.dim() == 2 and audio.is_contiguous()
        assert filters.dtype == torch.float32 and filters.shape[0] == 201 and filters.is_contiguous()
        B, N = audio.shape
        M = filters.shape[1]
        out = self.empty((B, M, N // 160), torch.float32)
        clip = self.empty((B,), torch.float32)
        self._chk(self.lib.dw_logmel(_p(audio), B, N, _p(filters), M, _p(self._twiddle), _p(self._window), _p(out),
                                     _p(clip), self._stream()), "logmel")
        return out

    def gemm(self, a, b, *, trans_a=False, trans_b=False, bias=None, act=0, want_z=False, zgrad=None, residual=None,
             r_row_mod=0, round_res=True, out_dtype=None, out=None, tile=0, atomic_acc=False, split_k=0, ln=None,
             kv_append=None, colsum=None, z_row_pad=0, out_row_pad=0, overwrite=False):
        """C = op(a) @ op(b) with the fused epilogue of dw_gemm_bf16.  a: [M,K] (or [K,M] if trans_a);
        b: [N,K] (nn.Linear weight layout) or [K,N] if trans_b.  Inner strides must be 1.
        atomic_acc: out (fp32) += result via float atomics, with the K range split over several workgroups when the
        output has too few tiles to fill the 256 CUs (weight-gradient GEMMs: small M x N, K = all tokens).
        overwrite (with atomic_acc): the caller knows `out` holds zeros (first micro-batch after the gradient buffer was cleared) --
        the combination of the partial slabs STORES `out` instead of adding to it, and a single-slice launch stores its tiles
        instead of issuing one float atomic per element (the tied LM head's dE: 66 M atomics per step).
        Decode-step fusions (M <= 32 / 64, skinny kernel): ln=(gamma, beta[, eps]) makes the operand
        bf16(LayerNorm(a)) with `a` the f32 / bf16 residual stream; kv_append=(cache, split, rows_per_batch,
        batch_pitch, row0) stores output columns >= split into the K/V cache rows of their positions."""
        assert b.dtype == torch.bfloat16 and (a.dtype == torch.bfloat16 or ln is not None)
        assert a.stride(1) == 1 and b.stride(1) == 1
        if trans_a:
            K, M = a.shape
        else:
            M, K = a.shape
        if trans_b:
            Kb, N = b.shape
        else:
            N, Kb = b.shape
        assert K == Kb, (a.shape, b.shape, trans_a, trans_b)
        if out is None:
            out = self.empty((M, N + out_row_pad), self.lowp if out_dtype is None else out_dtype)
            if out_row_pad:                     # (row pitch of an output this call allocates: engine.row_pad)
                out = out[:, :N]
        assert out.shape == (M, N) and out.stride(1) == 1
        g = DwGemm()
        g.a, g.b, g.c = a.data_ptr(), b.data_ptr(), out.data_ptr()
        g.lda, g.ldb, g.ldc = a.stride(0), b.stride(0), out.stride(0)
        g.m, g.n, g.k = M, N, K
        g.trans_a, g.trans_b = int(trans_a), int(trans_b)
        g.act = int(act)
        g.c_dtype = _dt(out)
        g.tile = int(tile) if tile else 0      # 0: the library's choice (skinny-M kernel, tile rule, tail split)
        tile_key = g.tile if tile or M <= 64 else self.pick_tile(M, N)
        ws = None
        if atomic_acc:
            # out (fp32, contiguous) += A^T.B : weight-gradient form.  Few output tiles and a very long K: cut K into
            # slices so that ~all 256 CUs get one 256x256 tile job (or two rounds of them), each slice stores its fp32
            # partial into a workspace and dw_reduce_slices adds them to `out` (deterministic; float atomics measured
            # slower beyond 2 slices).  A single slice accumulates directly with atomics.
            assert out.dtype == torch.float32 and bias is None and residual is None and not want_z and zgrad is None
            if not tile:
                g.tile = tile_key = 256
            tiles = ((M + g.tile - 1) // g.tile) * ((N + g.tile - 1) // g.tile)
            nt = K // 64
            if split_k:
                sk = int(split_k)
            else:
                # fill whole rounds of the 256 CUs: maximise (tiles*sk) / (ceil(tiles*sk/256)*256), mild penalty per slice
                best, sk = -1.0, 1
                for cand in range(1, max(1, min(16, nt // 16)) + 1):
                    jobs = tiles * cand
                    eff = jobs / (-(-jobs // 256) * 256) - 0.004 * cand
                    if eff > best + 1e-9:
                        best, sk = eff, cand
            sk = max(1, min(sk, nt // 16))
            if sk > 1 and out.stride(1) == 1 and N % 4 == 0 and out.stride(0) % 4 == 0:
                # (optional row pad of the slabs, self.wgrad_slab_pad floats: [M x 5120] fp32 slabs have 20 480-byte rows and the
                # 1 KiB pieces a 256-column tile stores into 256 consecutive rows all land on the same four L2 channels -- measured neutral)
                ldw = N + self.wgrad_slab_pad
                ws = self.empty((sk, M, ldw), torch.float32)
                g.c, g.ldc = ws.data_ptr(), ldw
                g.split_k, g.slice_stride = sk, M * ldw
            elif overwrite:
                g.atomic_acc, g.split_k = 0, 1         # one slice into a buffer that holds zeros: plain fp32 stores, no float atomics
            else:
                g.atomic_acc, g.split_k = 1, 1
        if ln is not None:
            gamma, beta = ln[0], ln[1]
            assert a.dtype in (torch.float32, torch.bfloat16) and not trans_a      # (m <= 32, k <= 1280: the library checks)
            assert gamma.dtype == torch.float32 and beta.dtype == torch.float32 and gamma.numel() == K == beta.numel()
            g.a, g.ln_x, g.ld_lnx, g.ln_x_dtype = None, a.data_ptr(), a.stride(0), _dt(a)
            g.ln_gamma, g.ln_beta, g.ln_eps = gamma.data_ptr(), beta.data_ptr(), float(ln[2]) if len(ln) > 2 else 1e-5
        if kv_append is not None:
            cache, split, rpb, pitch, row0 = kv_append
            assert cache.dtype == torch.bfloat16 and cache.stride(-1) == 1
            g.kv_out, g.kv_ld, g.kv_split = cache.data_ptr(), N - split, int(split)
            g.kv_rows_per_batch, g.kv_batch_pitch, g.kv_row0 = int(rpb), int(pitch), int(row0)
        z = None
        if bias is not None:
            assert bias.dtype == torch.float32 and bias.numel() == N and bias.is_contiguous()
            g.bias = bias.data_ptr()
        if want_z:
            # want_z="grad": the epilogue stores gelu'(z) as fp16 (what the backward multiplies by) instead of z as bf16
            as_grad = want_z == "grad"
            assert not as_grad or act == 1
            z = self.empty((M, N + z_row_pad), torch.float16 if as_grad else torch.bfloat16)
            if z_row_pad:                       # (row pitch of the stored by-product: engine.ffn_row_pad)
                z = z[:, :N]
            g.z_out, g.ldz, g.z_is_gelu_grad = z.data_ptr(), z.stride(0), int(as_grad)
        if zgrad is not None:
            assert zgrad.dtype in (torch.bfloat16, torch.float16) and zgrad.shape == (M, N) and zgrad.stride(1) == 1
            assert not (want_z and (zgrad.dtype == torch.float16) != (want_z == "grad"))
            g.zgrad_in, g.ldzg = zgrad.data_ptr(), zgrad.stride(0)
            g.z_is_gelu_grad = int(zgrad.dtype == torch.float16)
        if colsum is not None:      # f32 [N] += column sums of the stored output (bias gradient of the producing Linear)
            assert colsum.dtype == torch.float32 and colsum.numel() == N and colsum.is_contiguous() and not atomic_acc and M > 64
            g.colsum_out = colsum.data_ptr()
        if residual is not None:
            assert residual.stride(1) == 1 and residual.shape[1] == N
            g.r, g.ldr, g.r_dtype = residual.data_ptr(), residual.stride(0), _dt(residual)
            g.r_row_mod = int(r_row_mod)
            g.round_res = int(bool(round_res))
        e0 = self._t0()
        self._chk(self.lib.dw_gemm_bf16(C.byref(g), self._stream()), f"gemm m={M} n={N} k={K} ta={trans_a} tb={trans_b}")
        if ws is not None:
            self._chk(self.lib.dw_reduce_slices_ld(ws.data_ptr(), g.slice_stride, g.ldc, g.split_k, out.data_ptr(), out.stride(0),
                                                   M, N, 0 if overwrite else 1, self._stream()), "reduce_slices")
        key = f"gemm_t{tile_key}_{'T' if trans_a else 'N'}{'T' if trans_b else 'N'}"
        if self.profile_detail:
            key += (f" m{M} n{N} k{K} c{'f32' if out.dtype == torch.float32 else 'bf16'}"
                    f"{' bias' if bias is not None else ''}{' act%d' % act if act else ''}{' z' if want_z else ''}"
                    f"{' zg' if zgrad is not None else ''}"
                    f"{' r' + ('f32' if residual.dtype == torch.float32 else 'bf16') if residual is not None else ''}"
                    f"{' sk%d' % g.split_k if g.split_k else ''}")
        self._t1(e0, key, 2.0 * M * N * K)
        return (out, z) if want_z else out

    def decode_pass(self, desc):
        """One decoder pass of cached greedy decoding through the single C entry point dw_decode_step (every launch of
        the pass enqueued by the library; `desc` is a DwDecodeStep whose buffers the caller keeps alive)."""
        e0 = self._t0()
        self._chk(self.lib.dw_decode_step(C.byref(desc), self._stream()), "decode_step")
        self._t1(e0, "decode_step", 0.0)

    def layernorm_fwd(self, x, gamma, beta, eps=1e-5, save_stats=True, out=None):
        """x, out: 2-D with unit column stride; their row pitches may exceed the row length (padded activation buffers)."""
        rows, cols = x.shape
        assert x.stride(1) == 1 and gamma.dtype == torch.float32 and beta.dtype == torch.float32
        y = self.empty((rows, cols), torch.bfloat16) if out is None else out
        assert y.stride(1) == 1 and y.shape == (rows, cols) and y.dtype == torch.bfloat16
        mean = self.empty((rows,), torch.float32) if save_stats else None
        rstd = self.empty((rows,), torch.float32) if save_stats else None
        self._chk(self.lib.dw_layernorm_fwd_ld(_p(x), _dt(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, cols,
                                               float(eps), x.stride(0), y.stride(0), self._stream()), "layernorm_fwd")
        return y, mean, rstd

    def layernorm_bwd(self, dy, x, mean, rstd, gamma, dres, dgamma, dbeta, out_lowp=None, colsum=None):
        """dres (f32 [rows,cols]) += LN'(dy) if dres is given, else a new tensor is returned.  dgamma/dbeta += ...
        out_lowp (bf16 [rows,cols]) receives bf16(dres) and colsum (f32 [cols]) += its column sums (fused).
        Every 2-D argument may have a row pitch larger than cols (unit column stride)."""
        rows, cols = x.shape
        assert dy.dtype == torch.bfloat16 and dy.stride(1) == 1 and x.stride(1) == 1 and dy.shape == (rows, cols)
        acc = dres is not None
        if dres is None:
            dres = self.empty((rows, cols), torch.float32)
        assert dres.dtype == torch.float32 and dres.stride(1) == 1 and dres.shape == (rows, cols)
        ldl = cols
        if out_lowp is not None:
            assert out_lowp.dtype == torch.bfloat16 and out_lowp.stride(1) == 1 and out_lowp.shape == (rows, cols)
            ldl = out_lowp.stride(0)
        assert colsum is None or out_lowp is not None
        self._chk(self.lib.dw_layernorm_bwd_ld(_p(dy), _p(x), _dt(x), _p(mean), _p(rstd), _p(gamma), _p(dres), int(acc),
                                               _p(dgamma), _p(dbeta), _p(out_lowp), _p(colsum), rows, cols,
                                               dy.stride(0), x.stride(0), dres.stride(0), ldl, self._stream()), "layernorm_bwd")
        return dres

    def attn_fwd_varlen(self, q, k, v, H, max_q, q_start, q_len, causal, scale, out, Lk=0, kv_batches=0, self_attention=True,
                        flops=0.0):
        """Ragged batches over packed rows (dw_attn_fwd_varlen).  q_start / q_len: int32 device tensors, one entry per sequence.
        self_attention: k / v are rows of the same packed buffers; otherwise k / v are [kv_batches * Lk] rectangular rows."""
        for t in (q, k, v, out):
            assert t.dtype == torch.bfloat16 and t.stride(1) == 1
        assert q_start.dtype == torch.int32 and q_len.dtype == torch.int32 and q_start.numel() == q_len.numel()
        n = q_start.numel()
        e0 = self._t0()
        self._chk(self.lib.dw_attn_fwd_varlen(_p(q), _p(k), _p(v), _p(out), n, H, int(max_q), int(Lk), q.stride(0), k.stride(0),
                                              v.stride(0), out.stride(0), _p(q_start), _p(q_len),
                                              _p(q_start) if self_attention else None, _p(q_len) if self_attention else None,
                                              int(kv_batches), int(Lk), int(causal), float(scale), self._stream()), "attn_fwd_varlen")
        self._t1(e0, "attn_fwd", flops)
        return out

    def attn_fwd(self, q, k, v, B, H, Lq, Lk, causal, scale, out=None, kv_batch_rows=None):
        """kv_batch_rows: rows between consecutive batches of k/v in memory (a padded KV cache), default Lk."""
        for t in (q, k, v):
            assert t.dtype == torch.bfloat16 and t.stride(1) == 1
        o = self.empty((B * Lq, H * 64), torch.bfloat16) if out is None else out
        assert o.dtype == torch.bfloat16 and o.stride(1) == 1 and o.shape == (B * Lq, H * 64)
        lse = self.empty((B, H, Lq), torch.float32)
        e0 = self._t0()
        self._chk(self.lib.dw_attn_fwd_ex(_p(q), _p(k), _p(v), _p(o), _p(lse), B, H, Lq, Lk, q.stride(0), k.stride(0),
                                          v.stride(0), o.stride(0), Lq, Lk if kv_batch_rows is None else kv_batch_rows,
                                          int(causal), float(scale), self._stream()), "attn_fwd")
        self._t1(e0, "attn_fwd", 4.0 * B * H * Lq * Lk * 64)
        return o, lse

    def attn_bwd(self, q, k, v, o, do, lse, B, H, Lq, Lk, causal, scale, dq=None, dk=None, dv=None, dq_colsum=None,
                 dv_colsum=None):
        """dq_colsum / dv_colsum: f32 [H*64] += column sums of dq / dv (the q_proj / v_proj bias gradients)."""
        if dq is None:
            dq = self.empty((B * Lq, H * 64), torch.bfloat16)
        if dk is None:
            dk = self.empty((B * Lk, H * 64), torch.bfloat16)
        if dv is None:
            dv = self.empty((B * Lk, H * 64), torch.bfloat16)
        delta = self.empty((2, B, H, Lq), torch.float32)     # scratch: [-delta | -lse / scale]
        for t in (q, k, v, o, do, dq, dk, dv):
            assert t.dtype == torch.bfloat16 and t.stride(1) == 1
        e0 = self._t0()
        for t in (dq_colsum, dv_colsum):
            assert t is None or (t.dtype == torch.float32 and t.numel() == H * 64 and t.is_contiguous())
        part = None
        if dq_colsum is not None or dv_colsum is not None:
            # per-batch-row partial sums [2][B][H*64] (zeroed here, filled by the kernels' atomics, added up below)
            key = (B, H)
            if not hasattr(self, "_attn_cs"):
                self._attn_cs = {}
            if key not in self._attn_cs:
                self._attn_cs[key] = self.zeros((2, B, H * 64), torch.float32)
            part = self._attn_cs[key]
            part.zero_()
        self._chk(self.lib.dw_attn_bwd_ex(_p(q), _p(k), _p(v), _p(o), _p(do), _p(lse), _p(delta), _p(dq), _p(dk), _p(dv),
                                          B, H, Lq, Lk, q.stride(0), k.stride(0), v.stride(0), o.stride(0), do.stride(0),
                                          dq.stride(0), dk.stride(0), dv.stride(0), int(causal), float(scale),
                                          _p(part[0]) if dq_colsum is not None else None,
                                          _p(part[1]) if dv_colsum is not None else None, self._stream()), "attn_bwd")
        for i, dst in enumerate((dq_colsum, dv_colsum)):
            if dst is not None:
                self._chk(self.lib.dw_reduce_slices(_p(part[i]), H * 64, B, _p(dst), H * 64, 1, self._stream()),
                          "reduce_slices")
        self._t1(e0, "attn_bwd", 10.0 * B * H * Lq * Lk * 64)
        return dq, dk, dv

    def distill_loss(self, s_logits, t_logits, labels, V, temperature, ce_weight, kl_weight, grad_scale, want_grad,
                     grad_out=None, weights_dev=None):
        """Returns losses f32[4] = (ce, kl, total, n_valid).  With want_grad the gradient w.r.t. the student logits
        overwrites s_logits in place (bf16), or goes to `grad_out` (same shape and dtype) when that is given.
        `weights_dev` (f32[2] on the device) replaces (ce_weight, kl_weight): dw_distill_loss_w."""
        rows, ld = s_logits.shape
        if grad_out is not None:
            assert want_grad and grad_out.shape == s_logits.shape and grad_out.dtype == torch.bfloat16 and grad_out.is_contiguous()
        assert s_logits.dtype == torch.bfloat16 and t_logits.dtype == torch.bfloat16
        assert s_logits.is_contiguous() and t_logits.is_contiguous() and t_logits.shape == s_logits.shape
        assert labels.dtype == torch.int64 and labels.numel() == rows and labels.is_contiguous()
        losses = self.empty((4,), torch.float32)
        row_ce = self.empty((rows,), torch.float32)
        row_kl = self.empty((rows,), torch.float32)
        counts = self.empty((2,), torch.int32)
        dl = (_p(s_logits) if grad_out is None else _p(grad_out)) if want_grad else None
        if weights_dev is not None:
            assert weights_dev.dtype == torch.float32 and weights_dev.numel() == 2 and weights_dev.is_contiguous()
            self._chk(self.lib.dw_distill_loss_w(_p(s_logits), _p(t_logits), _p(labels), rows, V, ld, float(temperature),
                                                 _p(weights_dev), float(grad_scale), _p(losses), dl, _p(row_ce), _p(row_kl),
                                                 _p(counts), self._stream()), "distill_loss_w")
            return losses
        self._chk(self.lib.dw_distill_loss(_p(s_logits), _p(t_logits), _p(labels), rows, V, ld, float(temperature),
                                           float(ce_weight), float(kl_weight), float(grad_scale), _p(losses), dl,
                                           _p(row_ce), _p(row_kl), _p(counts),
                                           self._stream()), "distill_loss")
        return losses

    def embed_fwd(self, ids, tok, pos, out_dtype, rows_alloc=0):
        """rows_alloc > B*T: the output buffer gets that many rows (the extra rows are zero; callers whose GEMMs run
        over a padded row count)."""
        B, T = ids.shape
        D = tok.shape[1]
        assert ids.dtype == torch.int64 and ids.is_contiguous() and tok.is_contiguous() and pos.is_contiguous()
        assert tok.dtype == pos.dtype
        if rows_alloc > B * T:
            out = self.empty((rows_alloc, D), out_dtype)
            out[B * T:].zero_()
        else:
            out = self.empty((B * T, D), out_dtype)
        self._chk(self.lib.dw_embed_fwd(_p(ids), _p(tok), _p(pos), _dt(tok), _p(out), _dt(out), B, T, D,
                                        self._stream()), "embed_fwd")
        return out

    def embed_bwd(self, dx, ids, dtok, dpos):
        B, T = ids.shape
        D = dx.shape[1]
        assert dx.dtype == torch.float32 and dx.is_contiguous() and dtok.dtype == torch.float32
        self._chk(self.lib.dw_embed_bwd(_p(dx), _p(ids), _p(dtok), _p(dpos), B, T, D, self._stream()), "embed_bwd")

    def im2col_mel(self, mel, kpad, out=None):
        B, Cc, T = mel.shape
        assert mel.dtype == torch.float32 and mel.is_contiguous()
        if out is None:
            out = self.empty((B * T, kpad), torch.bfloat16)
        assert out.is_contiguous() and out.shape == (B * T, kpad)
        self._chk(self.lib.dw_im2col_mel(_p(mel), _p(out), B, Cc, T, kpad, self._stream()), "im2col_mel")
        return out

    def im2col_s2(self, a, B, T, out=None):
        Cc = a.shape[1]
        assert a.dtype == torch.bfloat16 and a.is_contiguous() and a.shape[0] == B * T
        if out is None:
            out = self.empty((B * T // 2, 3 * Cc), torch.bfloat16)
        assert out.is_contiguous() and out.shape == (B * T // 2, 3 * Cc)
        self._chk(self.lib.dw_im2col_s2(_p(a), _p(out), B, T, Cc, self._stream()), "im2col_s2")
        return out

    def col2im_s2_gelu_bwd(self, dxcol, z, B, T, out=None):
        Cc = z.shape[1]
        assert dxcol.is_contiguous() and z.is_contiguous() and dxcol.dtype == torch.bfloat16
        dz = self.empty((B * T, Cc), torch.bfloat16) if out is None else out
        assert dz.is_contiguous() and dz.shape == (B * T, Cc)
        self._chk(self.lib.dw_col2im_s2_gelu_bwd(_p(dxcol), _p(z), _p(dz), B, T, Cc, self._stream()), "col2im")
        return dz

    def gelu_bwd(self, dy, z, out=None):
        assert dy.is_contiguous() and z.is_contiguous() and z.dtype == torch.bfloat16 and dy.shape == z.shape
        dz = self.empty(z.shape, torch.bfloat16) if out is None else out
        assert dz.is_contiguous() and dz.shape == z.shape
        self._chk(self.lib.dw_gelu_bwd(_p(dy), _dt(dy), _p(z), _p(dz), z.numel(), self._stream()), "gelu_bwd")
        return dz

    def pack_conv_weight(self, w, kpad, out=None):
        D, Cc, _ = w.shape
        assert w.dtype == torch.float32 and w.is_contiguous()
        if out is None:
            out = self.empty((D, kpad), torch.bfloat16)
        self._chk(self.lib.dw_pack_conv_weight(_p(w), _p(out), D, Cc, kpad, self._stream()), "pack_conv_weight")
        return out

    def unpack_conv_grad(self, gwp, gw, accumulate):
        D, Cc, _ = gw.shape
        assert gwp.dtype == torch.float32 and gwp.is_contiguous() and gw.is_contiguous()
        self._chk(self.lib.dw_unpack_conv_grad(_p(gwp), _p(gw), D, Cc, gwp.shape[1], int(accumulate), self._stream()),
                  "unpack_conv_grad")

    def cast_bf16(self, x, out=None):
        assert x.dtype == torch.float32 and x.is_contiguous()
        if out is None:
            out = self.empty(x.shape, torch.bfloat16)
        self._chk(self.lib.dw_cast_f32_bf16(_p(x), _p(out), x.numel(), self._stream()), "cast_f32_bf16")
        return out

    def cast_f32(self, x, out=None):
        assert x.dtype == torch.bfloat16 and x.is_contiguous()
        if out is None:
            out = self.empty(x.shape, torch.float32)
        self._chk(self.lib.dw_cast_bf16_f32(_p(x), _p(out), x.numel(), self._stream()), "cast_bf16_f32")
        return out

    def colsum(self, x, out, accumulate):
        rows, cols = x.shape
        assert x.dtype == torch.bfloat16 and x.stride(1) == 1 and out.dtype == torch.float32
        self._chk(self.lib.dw_colsum_bf16(_p(x), x.stride(0), rows, cols, _p(out), int(accumulate), self._stream()),
                  "colsum")
        return out

    def gather_rows(self, src, idx, out):
        """out[i] = src[idx[i]] for i < len(idx) (rows of 2-D tensors with unit column stride; idx int32 on the device)."""
        n, rb = idx.numel(), src.shape[1] * src.element_size()
        assert src.stride(1) == 1 and out.stride(1) == 1 and out.shape[1] == src.shape[1] and out.dtype == src.dtype
        assert idx.dtype == torch.int32 and idx.is_contiguous() and out.shape[0] >= n
        self._chk(self.lib.dw_move_rows(_p(src), src.stride(0) * src.element_size(), _p(out),
                                        out.stride(0) * out.element_size(), _p(idx), n, rb, 0, self._stream()), "move_rows")
        return out

    def scatter_rows(self, src, idx, out):
        """out[idx[i]] = src[i] for i < len(idx); the other rows of `out` are left as they are."""
        n, rb = idx.numel(), src.shape[1] * src.element_size()
        assert src.stride(1) == 1 and out.stride(1) == 1 and out.shape[1] == src.shape[1] and out.dtype == src.dtype
        assert idx.dtype == torch.int32 and idx.is_contiguous() and src.shape[0] >= n
        self._chk(self.lib.dw_move_rows(_p(src), src.stride(0) * src.element_size(), _p(out),
                                        out.stride(0) * out.element_size(), _p(idx), n, rb, 1, self._stream()), "move_rows")
        return out

    def add(self, a, b, out_dtype):
        assert a.is_contiguous() and b.is_contiguous() and a.shape == b.shape
        y = self.empty(a.shape, out_dtype)
        self._chk(self.lib.dw_add(_p(a), _dt(a), _p(b), _dt(b), _p(y), _dt(y), a.numel(), self._stream()), "add")
        return y

    def sumsq(self, g, out):
        assert g.dtype == torch.float32 and g.is_contiguous()
        if getattr(self, "_sumsq_ws", None) is None:
            self._sumsq_ws = self.empty((2048,), torch.float32)        # DW_SUMSQ_PARTIALS
        self._chk(self.lib.dw_sumsq_f32(_p(g), g.numel(), _p(out), _p(self._sumsq_ws), self._stream()), "sumsq")
        return out

    def greedy_select(self, logits, V, tokens, n, cur, *, suppress=None, begin_suppress=None, first=False, no_eos=False,
                      forced=False, ts_begin=-1, max_initial=-1, begin_index=1, eos=-1, fill=-1, done=None):
        """One decoding step's token selection for the whole batch (csrc/decode.hip): logits bf16 [B, ld] -> next token
        written to tokens[:, n] and cur[:, 0]; masks are uint8 [V] (1 = never sampled); done bool [B] in/out."""
        B = tokens.shape[0]
        assert tokens.dtype == torch.int64 and tokens.is_contiguous() and cur.dtype == torch.int64 and cur.is_contiguous()
        if not forced:
            assert logits.dtype == torch.bfloat16 and logits.stride(1) == 1 and logits.shape[0] >= B
        for m in (suppress, begin_suppress):
            assert m is None or (m.dtype == torch.uint8 and m.numel() >= V and m.is_contiguous())
        assert done is None or (done.dtype == torch.bool and done.is_contiguous())
        self._chk(self.lib.dw_greedy_select(_p(logits), B, int(V), logits.stride(0) if logits is not None else 0,
                                            _p(suppress), _p(begin_suppress), int(first), int(no_eos), int(forced),
                                            int(ts_begin), int(max_initial), _p(tokens), tokens.stride(0), int(n),
                                            int(begin_index), int(eos), int(fill), _p(done), _p(cur), self._stream()),
                  "greedy_select")

    def adamw(self, p, g, m, v, shadow, sumsq, max_norm, grad_mul, lr, beta1, beta2, eps, weight_decay, step):
        assert p.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous()
        self._chk(self.lib.dw_adamw(_p(p), _p(g), _p(m), _p(v), _p(shadow), p.numel(), _p(sumsq), float(max_norm),
                                    float(grad_mul), float(lr), float(beta1), float(beta2), float(eps),
                                    float(weight_decay), int(step), self._stream()), "adamw")

    def adam_state(self, lr, beta1, beta2, step=0):
        """Device-resident optimizer scalars (8 doubles, include/dwamd.h dw_adam_tick)."""
        return torch.tensor([lr, step, beta1, beta2, 0, 0, 0, 0], dtype=torch.float64, device=self.device)

    def adam_tick(self, state, gate=None):
        assert state.dtype == torch.float64 and state.numel() == 8 and (gate is None or gate.dtype == torch.float32)
        self._chk(self.lib.dw_adam_tick(_p(state), _p(gate), self._stream()), "adam_tick")

    def adamw_dev(self, p, g, m, v, shadow, sumsq, max_norm, grad_mul, state, eps, weight_decay):
        assert p.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous()
        self._chk(self.lib.dw_adamw_dev(_p(p), _p(g), _p(m), _p(v), _p(shadow), p.numel(), _p(sumsq), float(max_norm),
                                        float(grad_mul), _p(state), float(eps), float(weight_decay), self._stream()),
                  "adamw_dev")


def _timed(key):
    def deco(fn):
        def wrapper(self, *a, **k):
            e0 = self._t0()
            r = fn(self, *a, **k)
            self._t1(e0, key)
            return r
        wrapper.__name__, wrapper.__doc__ = fn.__name__, fn.__doc__
        return wrapper
    return deco


for _name, _key in (("layernorm_fwd", "ln_fwd"), ("layernorm_bwd", "ln_bwd"), ("distill_loss", "loss"),
                    ("logmel", "logmel"), ("adamw", "adamw"), ("adamw_dev", "adamw"), ("cast_bf16", "cast"), ("colsum", "colsum"),
                    ("sumsq", "sumsq"), ("embed_fwd", "embed"), ("embed_bwd", "embed"), ("im2col_mel", "conv_aux"),
                    ("im2col_s2", "conv_aux"), ("col2im_s2_gelu_bwd", "conv_aux"), ("gelu_bwd", "conv_aux"),
                    ("pack_conv_weight", "conv_aux"), ("unpack_conv_grad", "conv_aux"), ("greedy_select", "select")):
    setattr(HipOps, _name, _timed(_key)(getattr(HipOps, _name)))
