"""The step's GEMM launches with their REAL epilogues, per kernel variant (DW_VARIANTS = JSON list of [dw_debug_set key 0
value, key 11 value]; key 11: 1 = default, 65 = one run-time epilogue walk for every flavour, 17 = no epilogue).  Operand A and the output rotate over three buffers (369 MB of A: past the Infinity Cache, as in
the step).  Every variant is checked bit for bit against the 16-wave reference kernel (variant 3).  Round 3 used it for
the two-workgroups-per-CU experiment (profiles/r3_gemm_two_workgroups_per_cu.md; that kernel is not in the build)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
def rnd(shape, s=1.0, dt=torch.bfloat16): return (torch.randn(shape, device="cuda") * s).to(dt)
M = 48000
cases = [  # name, M, N, K, trans_b, kwargs-builder
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
    ("teacher dec qkv M=14400", 14400, 3840, 1280, False, lambda N: dict(bias=rnd((N,), 0.1, torch.float32))),
    ("teacher dec out M=14400", 14400, 1280, 1280, False, lambda N: dict(bias=rnd((N,), 0.1, torch.float32), residual=rnd((14400, N)))),
    ("student dec fc1 M=14304", 14304, 5120, 1280, False, lambda N: dict(bias=rnd((N,), 0.1, torch.float32), act=1, want_z="grad")),
    ("lm head M=14304", 14304, 51904, 1280, False, lambda N: {}),
]
only = os.environ.get("DW_CASES")
if only:
    cases = [c for c in cases if any(k in c[0] for k in only.split(","))]
variants = json.loads(os.environ.get("DW_VARIANTS", "[[2163,1],[2163,65]]"))
rounds = int(os.environ.get("DW_ROUNDS", "3"))
def run(a, b, out, tb, kw):
    kw = dict(kw)
    od = kw.pop("out_dtype", None)
    return ops.gemm(a, b, trans_b=tb, out=out, **kw)
for name, m, N, K, tb, mk in cases:
    As = [rnd((m, K)) for _ in range(3)]
    b = rnd((K, N) if tb else (N, K), 0.05)
    kw = mk(N)
    if "residual" in kw and kw["residual"].shape[0] != m:
        kw["residual"] = kw["residual"][:m]
    if "zgrad" in kw:
        kw["zgrad"] = kw["zgrad"][:m]
    odt = kw.get("out_dtype", torch.bfloat16)
    outs = [torch.empty(m, N, device="cuda", dtype=odt) for _ in range(3)]
    ops.lib.dw_debug_set(0, 3); ops.lib.dw_debug_set(11, 1)
    r = run(As[0], b, outs[0], tb, kw)
    ref = (r[0] if isinstance(r, tuple) else r).clone()
    refz = r[1].clone() if isinstance(r, tuple) else None
    res = {}
    for v, st in variants:
        ops.lib.dw_debug_set(0, v); ops.lib.dw_debug_set(11, st)
        r = run(As[0], b, outs[1], tb, kw)
        o = r[0] if isinstance(r, tuple) else r
        ok = torch.equal(o, ref) and (refz is None or torch.equal(r[1], refz))
        if not ok:
            d = (o.float() - ref.float()).abs().max().item()
            print(f"MISMATCH {name} variant {v} stagger {st}: max abs diff {d}", flush=True)
        res[(v, st)] = []
    for _ in range(rounds):
        for v, st in variants:
            ops.lib.dw_debug_set(0, v); ops.lib.dw_debug_set(11, st)
            for i in range(3): run(As[i % 3], b, outs[i % 3], tb, kw)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for i in range(12): run(As[i % 3], b, outs[i % 3], tb, kw)
            e.record(); torch.cuda.synchronize()
            res[(v, st)].append(s.elapsed_time(e) / 12 * 1e3)
    line = {f"{v}/{st}": round(sorted(t)[len(t) // 2], 1) for (v, st), t in res.items()}
    base = sorted(res[tuple(variants[0])])[rounds // 2]
    best = min(line, key=lambda k: line[k])
    print(f"{name:34s} us/launch {line}  best {best} ({base / line[best]:.3f}x, {2.0 * m * N * K / line[best] / 1e6:.0f} TF/s)", flush=True)
    del As, outs, kw, b
    torch.cuda.empty_cache()
ops.lib.dw_debug_set(0, 2163); ops.lib.dw_debug_set(11, 1)
