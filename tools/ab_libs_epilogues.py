"""Same-process A/B of two BUILDS of the library (libdwamd.so against libdwamd_base.so) on the step's GEMM launches with their
real epilogues (the cases of tools/bench_gemm_epilogues.py): us per launch, medians of interleaved rounds, bit-identity."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd import ops_hip as oh
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
libs = {"base": oh.load_library(os.path.join(os.path.dirname(oh.LIB_PATH), "libdwamd_base.so"))}
_b2 = os.path.join(os.path.dirname(oh.LIB_PATH), "libdwamd_base2.so")      # (optional third build: tools/build_variant_lib.sh "<flags>" libdwamd_base2.so)
if os.path.exists(_b2) and not os.environ.get("DW_NO_BASE2"):
    libs["base2"] = oh.load_library(_b2)
libs["new"] = ops.lib
def rnd(shape, s=1.0, dt=torch.bfloat16): return (torch.randn(shape, device="cuda") * s).to(dt)
M = 48000
cases = [
    ("qkv bias", M, 3840, 1280, False, lambda N: dict(bias=rnd((N,), 0.1, torch.float32))),
    ("out-proj student bias+res f32", M, 1280, 1280, False, lambda N: dict(bias=rnd((N,), 0.1, torch.float32), residual=rnd((M, N), 1.0, torch.float32), out_dtype=torch.float32)),
    ("out-proj teacher bias+res bf16", M, 1280, 1280, False, lambda N: dict(bias=rnd((N,), 0.1, torch.float32), residual=rnd((M, N)))),
    ("fc1 student bias+gelu+g", M, 5120, 1280, False, lambda N: dict(bias=rnd((N,), 0.1, torch.float32), act=1, want_z="grad")),
    ("fc1 teacher bias+gelu", M, 5120, 1280, False, lambda N: dict(bias=rnd((N,), 0.1, torch.float32), act=1)),
    ("fc2 student bias+res f32", M, 1280, 5120, False, lambda N: dict(bias=rnd((N,), 0.1, torch.float32), residual=rnd((M, N), 1.0, torch.float32), out_dtype=torch.float32)),
    ("fc2 teacher bias+res bf16", M, 1280, 5120, False, lambda N: dict(bias=rnd((N,), 0.1, torch.float32), residual=rnd((M, N)))),
    ("dX fc2 zg", M, 5120, 1280, True, lambda N: dict(zgrad=rnd((M, N), 0.5, torch.float16))),
    ("dX fc1", M, 1280, 5120, True, lambda N: {}),
    ("dX qkv", M, 1280, 3840, True, lambda N: {}),
    ("dX out", M, 1280, 1280, True, lambda N: {}),
    ("student dec fc1 M=14304", 14304, 5120, 1280, False, lambda N: dict(bias=rnd((N,), 0.1, torch.float32), act=1, want_z="grad")),
    ("lm head M=14304", 14304, 51904, 1280, False, lambda N: {}),
]
only = os.environ.get("DW_CASES")
if only:
    cases = [c for c in cases if any(k in c[0] for k in only.split(","))]
rounds = int(os.environ.get("DW_ROUNDS", "3"))
def run(a, b, out, tb, kw):
    kw = dict(kw); kw.pop("out_dtype", None)
    return ops.gemm(a, b, trans_b=tb, out=out, **kw)
tot = {k: 0.0 for k in libs}
for name, m, N, K, tb, mk in cases:
    As = [rnd((m, K)) for _ in range(3)]
    b = rnd((K, N) if tb else (N, K), 0.05)
    kw = mk(N)
    odt = kw.get("out_dtype", torch.bfloat16)
    outs = [torch.empty(m, N, device="cuda", dtype=odt) for _ in range(3)]
    got = {}
    for k, lib in libs.items():
        ops.lib = lib
        r = run(As[0], b, outs[0], tb, kw)
        got[k] = [t.clone() for t in (r if isinstance(r, tuple) else (r,))]
    same = all(torch.equal(x, y) for x, y in zip(got["new"], got["base"]))
    res = {k: [] for k in libs}
    for _ in range(rounds):
        for k, lib in libs.items():
            ops.lib = lib
            for i in range(3): run(As[i % 3], b, outs[i % 3], tb, kw)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for i in range(12): run(As[i % 3], b, outs[i % 3], tb, kw)
            e.record(); torch.cuda.synchronize()
            res[k].append(s.elapsed_time(e) / 12 * 1e3)
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    for k in med: tot[k] += med[k]
    print(f"{name:34s} us/launch " + "  ".join(f"{k} {v:7.1f}" for k, v in med.items()) +
          f"  ({med['base'] / med['new']:.3f}x, {2.0 * m * N * K / med['new'] / 1e6:.0f} TF/s)  {'bit-identical' if same else 'DIFFERENT'}", flush=True)
    del As, outs, kw, b, got
    torch.cuda.empty_cache()
print("sum of medians: " + ", ".join(f"{k} {v:.0f} us" for k, v in tot.items()))
