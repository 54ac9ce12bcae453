"""Same-process A/B of attention kernel variants selected by dw_debug_set keys (3: backward staging / PIPE, 22: forward
variant) at the step's shapes; results compared with the default kernels'."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
D, H = 1280, 20
VARIANTS = {"default": {3: 5, 22: 0}}
for spec in sys.argv[1:]:             # name=key:value,key:value
    name, kv = spec.split("=")
    VARIANTS[name] = dict(VARIANTS["default"])
    VARIANTS[name].update({int(a.split(":")[0]): int(a.split(":")[1]) for a in kv.split(",")})
def timed(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [("encoder self", 32, 1500, 1500, False), ("cross", 32, 448, 1500, False)]
for name, B, Lq, Lk, causal in shapes:
    q = torch.randn(B * Lq, D, device="cuda").bfloat16()
    kv = torch.randn(B * Lk, 2 * D, device="cuda").bfloat16()
    k, v = kv[:, :D], kv[:, D:]
    do = torch.randn(B * Lq, D, device="cuda").bfloat16()
    res, outs = {}, {}
    for rnd in range(3):
        for tag, keys in VARIANTS.items():
            for kk, vv in keys.items():
                assert ops.lib.dw_debug_set(kk, vv) == 0
            o, lse = ops.attn_fwd(q, k, v, B, H, Lq, Lk, causal, 0.125)
            g = ops.attn_bwd(q, k, v, o, do, lse, B, H, Lq, Lk, causal, 0.125)
            outs[tag] = (o, lse) + tuple(g)
            res.setdefault(tag + " fwd", []).append(timed(lambda: ops.attn_fwd(q, k, v, B, H, Lq, Lk, causal, 0.125)))
            res.setdefault(tag + " bwd", []).append(timed(lambda: ops.attn_bwd(q, k, v, o, do, lse, B, H, Lq, Lk, causal, 0.125)))
    fl = 4.0 * B * H * Lq * Lk * 64 * (0.5 if causal else 1.0)
    for kk, t in res.items():
        m = sorted(t)[len(t) // 2]
        f = fl * (2.5 if "bwd" in kk else 1.0)
        print(f"{name:13s} {kk:24s} us: " + " ".join(f"{x:.0f}" for x in t) + f"   median {f / m / 1e6:.0f} TFLOP/s", flush=True)
    for tag in VARIANTS:
        if tag == "default": continue
        d = [(a.float() - b.float()).abs().max().item() / (b.float().abs().max().item() + 1e-30) for a, b in zip(outs[tag], outs["default"])]
        print(f"{name:13s} max |{tag} - default| / max|default| for o, lse, dq, dk, dv: " + " ".join(f"{x:.2e}" for x in d), flush=True)
for kk, vv in VARIANTS["default"].items():
    ops.lib.dw_debug_set(kk, vv)
