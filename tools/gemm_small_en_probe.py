"""Kernel choice for the D = 768 shapes of BASELINE config 2 (small.en, M = 48 000): TFLOP/s of the selectable main loops /
tiles per shape and epilogue flavour, interleaved rounds in one process."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
M, D, F = 32 * 1500, 768, 3072
def rnd(shape, s=1.0): return (torch.randn(shape, device="cuda") * s).bfloat16()
res32 = torch.randn(M, D, device="cuda")
bias = {n: torch.randn(n, device="cuda") for n in (D, 3 * D, F)}
cases = [("qkv bias N=2304 K=768", 3 * D, D, False, dict(bias=bias[3 * D])),
         ("out bias+res32 N=768 K=768", D, D, False, dict(bias=bias[D], residual=res32, out_dtype=torch.float32)),
         ("fc1 bias+gelu N=3072 K=768", F, D, False, dict(bias=bias[F], act=1)),
         ("fc2 bias+res32 N=768 K=3072", D, F, False, dict(bias=bias[D], residual=res32, out_dtype=torch.float32)),
         ("dX qkv plain N=768 K=2304", D, 3 * D, True, {}), ("dX fc1 plain N=768 K=3072", D, F, True, {}),
         ("dX out plain N=768 K=768", D, D, True, {})]
# (label, key-0 variant, tile argument, key-11 value)
variants = [("default", 2163, 0, 1), ("256-row 8-wave", 115, 256, 1), ("16-wave", 3, 256, 1), ("tile128", 2163, 128, 1),
            ("320 forced", 2163 | 4096, 256, 1), ("default+next-tile prefetch", 2163, 0, 129)]
for name, N, K, tb, kw in cases:
    a = rnd((M, K)); b = rnd((K, N) if tb else (N, K), 0.05)
    res = {v[0]: [] for v in variants}
    for r in range(4):
        for label, var, tile, k11 in variants:
            ops.lib.dw_debug_set(0, var); ops.lib.dw_debug_set(11, k11)
            for _ in range(2): ops.gemm(a, b, trans_b=tb, tile=tile, **kw)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): ops.gemm(a, b, trans_b=tb, tile=tile, **kw)
            e.record(); torch.cuda.synchronize()
            res[label].append(2.0 * M * N * K / (s.elapsed_time(e) / 10 * 1e-3) / 1e12)
    print(f"{name:30s}", {k: f"{sorted(v)[len(v)//2]:.0f}" for k, v in res.items()}, flush=True)
ops.lib.dw_debug_set(0, 2163); ops.lib.dw_debug_set(11, 1)
