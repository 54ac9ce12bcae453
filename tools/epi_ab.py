import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
M = 48000
def rnd(shape, s=1.0, dt=torch.bfloat16): return (torch.randn(shape, device="cuda") * s).to(dt)
x = rnd((M, 1280)); w1 = rnd((5120, 1280), 0.05); b1 = rnd((5120,), 0.1, torch.float32)
out = torch.empty(M, 5120, device="cuda", dtype=torch.bfloat16)
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for name, kw in (("plain", {}), ("bias", dict(bias=b1)), ("bias+z", dict(bias=b1, want_z=True)), ("bias+gelu", dict(bias=b1, act=1)),
                 ("bias+gelu+z", dict(bias=b1, act=1, want_z=True)), ("f32 out", dict(out_dtype=torch.float32))):
    o = out if "out_dtype" not in kw else torch.empty(M, 5120, device="cuda", dtype=torch.float32)
    kw2 = {k: v for k, v in kw.items() if k != "out_dtype"}
    t = timeit(lambda: ops.gemm(x, w1, out=o, **kw2))
    print(f"{name:14s} {t:.3f} ms  {2.0*M*5120*1280/t/1e9:.0f} TF", flush=True)
