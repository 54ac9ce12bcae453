"""Sustained (power-limited) rate of the two main loops: each variant runs alone for ~3 s per leg, legs interleaved."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
M = 48000
def rnd(shape, s=1.0): return (torch.randn(shape, device="cuda") * s).bfloat16()
bias = torch.randn(5120, device="cuda")
cases = [("NN qkv N=3840 K=1280 bias (320-row)", 3840, 1280, False, dict(bias=bias[:3840].contiguous())),
         ("NN fc2 N=1280 K=5120 plain (320-row)", 1280, 5120, False, {}),
         ("NT dX fc2 N=5120 K=1280 plain", 5120, 1280, True, {})]
SEC = float(os.environ.get("DW_SEC", "3"))
for name, N, K, tb, kw in cases:
    a = rnd((M, K)); b = rnd((K, N) if tb else (N, K), 0.05)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = {0: [], 7: []}
    for r in range(2):
        for v in (0, 7):
            ops.lib.dw_debug_set(20, v)
            for _ in range(3): ops.gemm(a, b, trans_b=tb, out=out, **kw)
            torch.cuda.synchronize()
            n = 0
            t0 = time.perf_counter()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            while time.perf_counter() - t0 < SEC:
                for _ in range(50): ops.gemm(a, b, trans_b=tb, out=out, **kw)
                n += 50
                torch.cuda.synchronize()
            e.record(); torch.cuda.synchronize()
            res[v].append(2.0 * M * N * K * n / (s.elapsed_time(e) * 1e-3) / 1e12)
    print(f"{name:40s} 32x32x16 {res[0]}  16x16x32 {res[7]}", flush=True)
ops.lib.dw_debug_set(20, 0)
