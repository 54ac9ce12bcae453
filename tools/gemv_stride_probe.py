"""Token-step GEMVs (M = 16 rows) with the weight rows padded by P elements: does the row pitch of W matter for the weight-streaming
kernel?  Four weight copies in rotation so that a launch does not find its stream in the caches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
M = 16
def timed(fn, n=60):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for name, N, K in (("qkv", 3840, 1280), ("out", 1280, 1280), ("fc1", 5120, 1280), ("fc2", 1280, 5120), ("lm head", 51904, 1280)):
    a = torch.randn(M, K, device="cuda").bfloat16()
    for P in (0, 64, 128):
        nb = 4 if N < 20000 else 3
        Ws = [(torch.randn(N, K + P, device="cuda") * 0.03).bfloat16() for _ in range(nb)]
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        i = [0]
        def fn():
            w = Ws[i[0] % nb]; i[0] += 1
            ops.gemm(a, w[:, :K], out=out)
        t = sorted(timed(fn) for _ in range(3))[1]
        print(f"{name:8s} N={N:5d} K={K:4d} pad {P:3d}: {t:6.2f} us  {N * K * 2 / t / 1e6:5.2f} TB/s", flush=True)
