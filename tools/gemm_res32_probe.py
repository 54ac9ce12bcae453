"""fp32-residual epilogue (student out-proj / fc2: HBM-bound epilogue) at D = 1280: 320-row persistent tiles against two
128-tile workgroups per CU (one's epilogue under the other's K loop) and the 16-wave kernel.  TFLOP/s medians."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
M = 32 * 1500
def rnd(shape, s=1.0): return (torch.randn(shape, device="cuda") * s).bfloat16()
bias = torch.randn(1280, device="cuda")
res32 = torch.randn(M, 1280, device="cuda")
resbf = res32.bfloat16()
cases = [("out bias+res32 N=1280 K=1280", 1280, 1280, dict(bias=bias, residual=res32, out_dtype=torch.float32)),
         ("fc2 bias+res32 N=1280 K=5120", 1280, 5120, dict(bias=bias, residual=res32, out_dtype=torch.float32)),
         ("out bias+res bf16 N=1280 K=1280 (teacher)", 1280, 1280, dict(bias=bias, residual=resbf))]
variants = [("default", 2163, 0), ("tile128", 2163, 128), ("16-wave", 3, 256), ("256-row 8-wave", 115, 256)]
for name, N, K, kw in cases:
    a = rnd((M, K)); b = rnd((N, K), 0.05)
    res = {v[0]: [] for v in variants}
    for r in range(4):
        for label, var, tile in variants:
            ops.lib.dw_debug_set(0, var)
            for _ in range(2): ops.gemm(a, b, tile=tile, **kw)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): ops.gemm(a, b, tile=tile, **kw)
            e.record(); torch.cuda.synchronize()
            res[label].append(2.0 * M * N * K / (s.elapsed_time(e) / 10 * 1e-3) / 1e12)
    print(f"{name:44s}", {k: f"{sorted(v)[len(v)//2]:.0f}" for k, v in res.items()}, flush=True)
ops.lib.dw_debug_set(0, 2163)
