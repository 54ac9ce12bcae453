"""Start offsets between the persistent workgroups (dw_debug_set key 12 = S | unit << 8: workgroup `local % S` sleeps
`unit` x ~4.5 us first) on the launches whose epilogue is an HBM burst (fp32 residual read + fp32 store): do the bursts of a
tile round, spread in time, overlap with the other CUs' K loops?  TFLOP/s medians, interleaved rounds."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
M = 32 * 1500
def rnd(shape, s=1.0): return (torch.randn(shape, device="cuda") * s).bfloat16()
bias = {n: torch.randn(n, device="cuda") for n in (1280, 3840, 5120)}
res32 = torch.randn(M, 1280, device="cuda")
cases = [("out bias+res32 N=1280 K=1280", 1280, 1280, dict(bias=bias[1280], residual=res32, out_dtype=torch.float32)),
         ("fc2 bias+res32 N=1280 K=5120", 1280, 5120, dict(bias=bias[1280], residual=res32, out_dtype=torch.float32)),
         ("qkv bias N=3840 K=1280", 3840, 1280, dict(bias=bias[3840])), ("fc1 bias+gelu N=5120 K=1280", 5120, 1280, dict(bias=bias[5120], act=1))]
staggers = [0, 2 | (4 << 8), 4 | (2 << 8), 8 | (1 << 8), 4 | (4 << 8), 3 | (3 << 8)]
for name, N, K, kw in cases:
    a = rnd((M, K)); b = rnd((N, K), 0.05)
    res = {s: [] for s in staggers}
    for r in range(4):
        for st in staggers:
            ops.lib.dw_debug_set(12, st)
            for _ in range(2): ops.gemm(a, b, **kw)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): ops.gemm(a, b, **kw)
            e.record(); torch.cuda.synchronize()
            res[st].append(2.0 * M * N * K / (s.elapsed_time(e) / 10 * 1e-3) / 1e12)
    print(f"{name:30s}", {f"S={s & 255} unit={s >> 8}": f"{sorted(v)[len(v)//2]:.0f}" for s, v in res.items()}, flush=True)
ops.lib.dw_debug_set(12, 0)
