"""Four waves x 128x128 (gemm_wp16_w4.hip) against the eight-wave kernels on the step's row-major shapes: sustained legs, interleaved."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
from distil_whisper_amd import ops_hip as _oh
libs = {0: ops.lib}
_base = os.path.join(os.path.dirname(_oh.LIB_PATH), "libdwamd_base.so")      # a build of tools/build_variant_lib.sh / build_base_lib.sh
if os.path.exists(_base): libs[1] = _oh.load_library(_base)                   # optional 6th config field = 1: run on it
if os.environ.get("DW_BASELIB"): libs[0] = libs[1]
M = 48000
def rnd(shape, s=1.0): return (torch.randn(shape, device="cuda") * s).bfloat16()
bias = torch.randn(5120, device="cuda")
cases = [c for c in [("NN qkv N=3840 K=1280 bias", 3840, 1280, False, dict(bias=bias[:3840].contiguous())),
         ("NN fc1 N=5120 K=1280 bias", 5120, 1280, False, dict(bias=bias.contiguous())),
         ("NN fc2 N=1280 K=5120 plain", 1280, 5120, False, {}),
         ("NT dX fc2 N=5120 K=1280 plain", 5120, 1280, True, {}),
         ("NN out N=1280 K=1280 bias+res bf16", 1280, 1280, False, dict(bias=bias[:1280].contiguous(), residual=rnd((M, 1280)))),
         ("NN out32 N=1280 K=1280 bias+res fp32", 1280, 1280, False, dict(bias=bias[:1280].contiguous(), residual=torch.randn(M, 1280, device="cuda"), out_dtype=torch.float32)),
         ("NN fc2-32 N=1280 K=5120 bias+res fp32", 1280, 5120, False, dict(bias=bias[:1280].contiguous(), residual=torch.randn(M, 1280, device="cuda"), out_dtype=torch.float32)),
         ("NT dX out N=1280 K=1280 plain", 1280, 1280, True, {}),
         ("NT dX qkv N=1280 K=3840 plain", 1280, 3840, True, {}),
         ("NT dX fc1 N=1280 K=5120 plain", 1280, 5120, True, {})] if os.environ.get("DW_CASE", "") in c[0]]
# (label, GEMM variant key 0, key 20 mask, key 11, key 19)
configs = [("default", 2163, 4, 1, 0), ("8w-256", 115, 4, 1, 0), ("4w-256", 115, 4 | 8 | 16, 1, 0),
           ("8w-256 no-epi", 115, 4, 17, 0), ("4w-256 no-epi", 115, 4 | 8 | 16, 17, 0), ("8w16-256 no-epi", 115, 7, 17, 0),
           ("4w no-epi no-frag", 115, 12, 17, 1), ("4w no-epi no-dma", 115, 12, 17, 2), ("4w no-epi mfma-only", 115, 12, 17, 3),
           ("8w16 no-epi no-frag", 115, 7, 17, 1), ("8w16 no-epi no-dma", 115, 7, 17, 2), ("8w16 no-epi mfma-only", 115, 7, 17, 3)]
if os.environ.get('DW_CFG'): configs = eval(os.environ['DW_CFG'])
SEC = float(os.environ.get("DW_SEC", "1.5"))
for name, N, K, tb, kw in cases:
    a = rnd((M, K)); b = rnd((K, N) if tb else (N, K), 0.05)
    out = torch.empty(M, N, device="cuda", dtype=kw.pop("out_dtype", torch.bfloat16))
    ref = None
    res = {c[0]: [] for c in configs}
    for r in range(2):
        for label, v, mi, sn, dbg, *rest in configs:
            ops.lib = libs[rest[0] if rest else 0]
            ops.lib.dw_debug_set(0, v); ops.lib.dw_debug_set(20, mi); ops.lib.dw_debug_set(11, sn); ops.lib.dw_debug_set(19, dbg)
            ops.lib.dw_debug_set(1, rest[1] if len(rest) > 1 else 0)          # optional 7th config field: strip width override
            out.zero_()
            for _ in range(3): ops.gemm(a, b, trans_b=tb, out=out, tile=256, **kw)
            torch.cuda.synchronize()
            if sn == 1 and dbg == 0:
                if ref is None: ref = out.clone()
                elif not torch.equal(ref, out): print(f"   MISMATCH {label}: max|d| {(ref.float() - out.float()).abs().max().item():.4g}")
            n = 0
            t0 = time.perf_counter()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            while time.perf_counter() - t0 < SEC:
                for _ in range(50): ops.gemm(a, b, trans_b=tb, out=out, tile=256, **kw)
                n += 50
                torch.cuda.synchronize()
            e.record(); torch.cuda.synchronize()
            res[label].append(round(2.0 * M * N * K * n / (s.elapsed_time(e) * 1e-3) / 1e12))
    print(name); [print(f"   {k:24s} {v}", flush=True) for k, v in res.items()]
ops.lib.dw_debug_set(0, 2163); ops.lib.dw_debug_set(20, 36); ops.lib.dw_debug_set(11, 1); ops.lib.dw_debug_set(19, 0); ops.lib.dw_debug_set(1, 0)
