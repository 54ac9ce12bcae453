"""dw_debug_set key 22: the partial last row block of the wide 256-row launches on the 128-tile kernel (on) against one launch (off)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
M, D = 48000, 1280
for name, N, kw in (("qkv", 3840, {}), ("fc1 gelu", 5120, dict(act=1))):
    a = torch.randn(M, D, device="cuda").bfloat16(); b = (torch.randn(N, D, device="cuda") * 0.03).bfloat16()
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = {0: [], 1: []}
    for r in range(5):
        for v in (0, 1):
            ops.lib.dw_debug_set(22, v)
            for _ in range(2): ops.gemm(a, b, bias=bias, out=out, **kw)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): ops.gemm(a, b, bias=bias, out=out, **kw)
            e.record(); torch.cuda.synchronize()
            res[v].append(s.elapsed_time(e) / 10 * 1e3)
    print(name, {v: f"{sorted(t)[2]:.1f} us" for v, t in res.items()}, flush=True)
ops.lib.dw_debug_set(22, 1)
