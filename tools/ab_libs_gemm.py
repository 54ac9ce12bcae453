"""Same-process A/B of two BUILDS of the library (libdwamd.so against libdwamd_base.so, tools/build_variant_lib.sh) on the
step's GEMM launches with their epilogue flavours: TFLOP/s medians of interleaved rounds, and bit-identity of the results."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd import ops_hip as oh
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
libs = {"new": ops.lib, "base": oh.load_library(os.path.join(os.path.dirname(oh.LIB_PATH), "libdwamd_base.so"))}
B = 32
def rnd(shape, s=1.0): return (torch.randn(shape, device="cuda") * s).bfloat16()
M = B * 1500
bias = {n: torch.randn(n, device="cuda") for n in (1280, 3840, 5120)}
# (name, N, K, trans_b, kwargs)
cases = [("qkv bias", 3840, 1280, False, dict(bias=bias[3840])), ("fc1 bias+gelu (teacher)", 5120, 1280, False, dict(bias=bias[5120], act=1)),
         ("fc1 bias+gelu+gelu' (student)", 5120, 1280, False, dict(bias=bias[5120], act=1, want_z="grad")),
         ("dX qkv plain", 1280, 3840, True, {}), ("dX out plain", 1280, 1280, True, {}), ("kv proj bias", 2560, 1280, False, dict(bias=bias[3840][:2560].contiguous())),
         ("out plain NN", 1280, 1280, False, {})]
for name, N, K, tb, kw in cases:
    a = rnd((M, K)); b = rnd((K, N) if tb else (N, K), 0.05)
    outs = {}
    res = {k: [] for k in libs}
    for k, lib in libs.items():
        ops.lib = lib
        r = ops.gemm(a, b, trans_b=tb, **kw)
        outs[k] = [t.clone() for t in (r if isinstance(r, tuple) else (r,))]
    same = all(torch.equal(x, y) for x, y in zip(outs["new"], outs["base"]))
    for rnd_i in range(5):
        for k, lib in libs.items():
            ops.lib = lib
            for _ in range(2): ops.gemm(a, b, trans_b=tb, **kw)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): ops.gemm(a, b, trans_b=tb, **kw)
            e.record(); torch.cuda.synchronize()
            res[k].append(2.0 * M * N * K / (s.elapsed_time(e) / 10 * 1e-3) / 1e12)
    print(f"{name:32s}", {k: f"{sorted(v)[len(v)//2]:.0f}" for k, v in res.items()}, "bit-identical" if same else "DIFFERENT", flush=True)
