"""GEMM rate against N at fixed M = 48000, K = 1280 (row-major operands), and against the rasterisation strip width at
N = 5120: looks for the cause of the N = 5120 plateau (same kernel, same K, ~12 % below N = 3840)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
M, K = 48000, int(os.environ.get("K", 1280))
a = torch.randn(M, K, device="cuda").bfloat16()
def rate(N, strip=0, reps=20):
    b = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.lib.dw_debug_set(1, strip)
    for _ in range(3): ops.gemm(a, b, out=out, tile=256)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): ops.gemm(a, b, out=out, tile=256)
    e.record(); torch.cuda.synchronize()
    ops.lib.dw_debug_set(1, 0)
    return 2.0 * M * N * K * reps / (s.elapsed_time(e) * 1e-3) / 1e12
for N in (1280, 2560, 3840, 4096, 4352, 4608, 4864, 5120, 5376, 6400, 7680, 10240):
    tiles = 188 * ((N + 255) // 256)
    print(f"N={N:6d} tiles {tiles:5d} rounds {tiles / 256:6.2f}: {rate(N):6.0f} TF/s", flush=True)
for strip in (2, 3, 4, 5, 7, 10, 20):
    print(f"N=5120 strip {strip:2d}: {rate(5120, strip):6.0f} TF/s", flush=True)
