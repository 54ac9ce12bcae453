"""Sustained GEMM rate: the same launch repeated for seconds, TFLOP/s per chunk of launches (power / clock management
acts on a longer time scale than a 10-launch micro-benchmark)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
M, N, K = 48000, int(os.environ.get("N", 3840)), int(os.environ.get("K", 1280))
a = torch.randn(M, K, device="cuda").bfloat16(); b = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
variants = [int(v) for v in os.environ.get("DW_VARIANTS", "3,19").split(",")]
chunks, per = int(os.environ.get("CHUNKS", 20)), int(os.environ.get("PER", 150))
for v in variants + variants:
    ops.lib.dw_debug_set(0, v)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(chunks + 1)]
    torch.cuda.synchronize()
    ev[0].record()
    for c in range(chunks):
        for _ in range(per): ops.gemm(a, b, out=out, tile=256)
        ev[c + 1].record()
    torch.cuda.synchronize()
    tf = [2.0 * M * N * K * per / (ev[c].elapsed_time(ev[c + 1]) * 1e-3) / 1e12 for c in range(chunks)]
    print(f"variant {v} N={N} K={K}: " + " ".join(f"{x:.0f}" for x in tf), flush=True)
    if os.environ.get("VENDOR"):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(chunks + 1)]
        out2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ev[0].record()
        for c in range(chunks):
            for _ in range(per): torch.matmul(a, b.t(), out=out2)
            ev[c + 1].record()
        torch.cuda.synchronize()
        tf = [2.0 * M * N * K * per / (ev[c].elapsed_time(ev[c + 1]) * 1e-3) / 1e12 for c in range(chunks)]
        print(f"vendor    N={N} K={K}: " + " ".join(f"{x:.0f}" for x in tf), flush=True)
ops.lib.dw_debug_set(0, 2163)
