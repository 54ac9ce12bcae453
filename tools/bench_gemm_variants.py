"""A/B of GEMM main-loop variants (dw_debug_set key 0), interleaved rounds in one process."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
B = 32
def rnd(shape, s=1.0): return (torch.randn(shape, device="cuda") * s).bfloat16()
shapes = [("fwd N=1280 K=1280", B*1500, 1280, 1280, False, False), ("fwd N=5120 K=1280", B*1500, 5120, 1280, False, False),
          ("fwd N=1280 K=5120", B*1500, 1280, 5120, False, False), ("dX N=5120 K=1280", B*1500, 5120, 1280, False, True),
          ("dW 5120x1280 K=48000", 5120, 1280, B*1500, True, True)]
if os.environ.get("DW_MORE"):
    shapes = [("qkv fwd", B*1500, 3840, 1280, False, False), ("conv2 fwd K=3840", B*1500, 1280, 3840, False, False),
              ("dX qkv K=3840", B*1500, 1280, 3840, False, True), ("dX out K=1280 N=1280", B*1500, 1280, 1280, False, True),
              ("dX fc1 K=5120 N=1280", B*1500, 1280, 5120, False, True),
              ("dec qkv M=14304", B*447, 3840, 1280, False, False), ("dec fc1 M=14304", B*447, 5120, 1280, False, False),
              ("lm head", B*447, 51904, 1280, False, False), ("lm head dX", B*447, 1280, 51904, False, True),
              ("dW qkv 3840x1280", 3840, 1280, B*1500, True, True), ("dW out 1280x1280", 1280, 1280, B*1500, True, True)]
variants = [int(v) for v in os.environ.get("DW_VARIANTS", "3,5,7").split(",")]
KEY = int(os.environ.get("DW_KEY", "0"))
for name, M, N, K, ta, tb in shapes:
    a = rnd((K, M) if ta else (M, K)); b = rnd((K, N) if tb else (N, K), 0.05)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = {v: [] for v in variants}
    ops.lib.dw_debug_set(0, 3)
    ref = ops.gemm(a, b, trans_a=ta, trans_b=tb, tile=256).clone()
    for v in variants:
        ops.lib.dw_debug_set(KEY, v)
        for rep in range(3):
            o = ops.gemm(a, b, trans_a=ta, trans_b=tb, tile=256)
            d = (o.float() - ref.float()).abs().max().item()
            if d != 0.0:
                print("MISMATCH variant", v, "rep", rep, "max abs diff", d, flush=True)
    for rnd_i in range(int(os.environ.get("DW_ROUNDS", "5"))):
        for v in variants:
            ops.lib.dw_debug_set(KEY, v)
            for _ in range(2): ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out, tile=256)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out, tile=256)
            e.record(); torch.cuda.synchronize()
            res[v].append(2.0*M*N*K/(s.elapsed_time(e)/10*1e-3)/1e12)
    print(name, {v: f"med {sorted(r)[len(r)//2]:.0f} max {max(r):.0f}" for v, r in res.items()}, flush=True)
ops.lib.dw_debug_set(0, 2163)
