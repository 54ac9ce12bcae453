"""Rasterisation strip width (dw_debug_set key 1) against GEMM rate for the step's big shapes, interleaved rounds."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
M = 48000
shapes = [("NN", 1280, 1280), ("NN", 3840, 1280), ("NN", 5120, 1280), ("NN", 1280, 5120), ("NT", 5120, 1280), ("NT", 1280, 1280),
          ("NT", 1280, 5120), ("NN", 2560, 1280)]
strips = [0, 2, 3, 4, 5, 6, 8]
for lay, N, K in shapes:
    tb = lay == "NT"
    a = torch.randn(M, K, device="cuda").bfloat16()
    b = (torch.randn((K, N) if tb else (N, K), device="cuda") * 0.05).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = {s: [] for s in strips}
    for rnd in range(4):
        for st in strips:
            ops.lib.dw_debug_set(1, st)
            for _ in range(2): ops.gemm(a, b, trans_b=tb, out=out, tile=256)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): ops.gemm(a, b, trans_b=tb, out=out, tile=256)
            e.record(); torch.cuda.synchronize()
            res[st].append(2.0 * M * N * K * 10 / (s.elapsed_time(e) * 1e-3) / 1e12)
    ops.lib.dw_debug_set(1, 0)
    print(f"{lay} N={N} K={K}: " + "  ".join(f"s{st}:{sorted(r)[len(r) // 2]:.0f}" for st, r in res.items()), flush=True)
