"""small.en-sized GEMMs (D = 768, M = 48 000): the 128 x 256 three-stage kernel forced (tile 129 with key 25 = 2) against the
library's choice."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
def rnd(shape, s=1.0): return (torch.randn(shape, device="cuda") * s).bfloat16()
M = int(os.environ.get("M", 48000)); D = int(os.environ.get("D", 768)); F = 4 * D
bias = {n: torch.randn(n, device="cuda") for n in (D, 3 * D, F)}
res32 = torch.randn(M, D, device="cuda"); resb = res32.bfloat16()
zin = torch.randn(M, F, device="cuda").to(torch.float16)
cases = [("qkv bias", 3 * D, D, False, dict(bias=bias[3 * D])), ("out bias+res32", D, D, False, dict(bias=bias[D], residual=res32, out_dtype=torch.float32)),
         ("out bias+resbf16", D, D, False, dict(bias=bias[D], residual=resb)),
         ("fc1 gelu+g", F, D, False, dict(bias=bias[F], act=1, want_z="grad")), ("fc1 gelu", F, D, False, dict(bias=bias[F], act=1)),
         ("fc2 bias+res32", D, F, False, dict(bias=bias[D], residual=res32, out_dtype=torch.float32)),
         ("dX qkv", D, 3 * D, True, {}), ("dX out", D, D, True, {}), ("dX fc1", D, F, True, {}), ("dX fc2 zg", F, D, True, dict(zgrad=zin))]
for name, N, K, tb, kw in cases:
    a = rnd((M, K)); b = rnd((K, N) if tb else (N, K), 0.05)
    res = {"auto": [], "wp128": []}
    ops.lib.dw_debug_set(25, 1); ref = ops.gemm(a, b, trans_b=tb, **kw)
    ops.lib.dw_debug_set(25, 2); got = ops.gemm(a, b, trans_b=tb, tile=129, **kw)
    same = all(torch.equal(x, y) for x, y in zip(ref if isinstance(ref, tuple) else (ref,), got if isinstance(got, tuple) else (got,)))
    for r in range(4):
        for label, key, tile in (("auto", 1, 0), ("wp128", 2, 129)):
            ops.lib.dw_debug_set(25, key)
            for _ in range(2): ops.gemm(a, b, trans_b=tb, tile=tile, **kw)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): ops.gemm(a, b, trans_b=tb, tile=tile, **kw)
            e.record(); torch.cuda.synchronize()
            res[label].append(s.elapsed_time(e) / 10 * 1e3)
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    print(f"M={M} D={D} {name:18s} N={N} K={K} identical={same} " + " ".join(f"{k}: {v:.1f} us ({2.0 * M * N * K / v / 1e6:.0f} TF/s)" for k, v in med.items()), flush=True)
ops.lib.dw_debug_set(25, 1)
