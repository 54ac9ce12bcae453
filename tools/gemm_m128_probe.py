"""Decoder-side GEMMs (M = 32 x live positions): the software-pipelined 128 x 256 tile (tile = 129) against the library's
choice, the lock-step 128-tile kernel and the 256-row kernels; correctness against the 128-tile kernel (bit-identical sums)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
def rnd(shape, s=1.0): return (torch.randn(shape, device="cuda") * s).bfloat16()
bias = {n: torch.randn(n, device="cuda") for n in (1280, 3840, 5120)}
cases = []
for M in [int(x) for x in os.environ.get("MS", "4480,4258,7136,14304").split(",")]:
    resb = torch.randn(M, 1280, device="cuda").bfloat16()
    res32 = torch.randn(M, 1280, device="cuda")
    cases += [(f"M={M} qkv N=3840 K=1280 bias", M, 3840, 1280, False, dict(bias=bias[3840])),
              (f"M={M} out N=1280 K=1280 bias+resbf16", M, 1280, 1280, False, dict(bias=bias[1280], residual=resb)),
              (f"M={M} out N=1280 K=1280 bias+res32", M, 1280, 1280, False, dict(bias=bias[1280], residual=res32, out_dtype=torch.float32)),
              (f"M={M} fc1 N=5120 K=1280 gelu", M, 5120, 1280, False, dict(bias=bias[5120], act=1)),
              (f"M={M} fc2 N=1280 K=5120 bias+resbf16", M, 1280, 5120, False, dict(bias=bias[1280], residual=resb)),
              (f"M={M} dX N=1280 K=3840 plain", M, 1280, 3840, True, {}),
              (f"M={M} dX N=1280 K=5120 plain", M, 1280, 5120, True, {})]
variants = [("auto", 0, 0), ("t128", 128, 0), ("wp128", 129, 0), ("t256 32x32", 256, 0), ("t256 16x16", 256, 7)]
for name, M, N, K, tb, kw in cases:
    a = rnd((M, K)); b = rnd((K, N) if tb else (N, K), 0.05)
    ref = ops.gemm(a, b, trans_b=tb, tile=128, **kw)
    got = ops.gemm(a, b, trans_b=tb, tile=129, **kw)
    same = bool(torch.equal(ref, got))
    res = {v[0]: [] for v in variants}
    for r in range(4):
        for label, tile, mi in variants:
            ops.lib.dw_debug_set(20, mi if mi else 36)
            for _ in range(2): ops.gemm(a, b, trans_b=tb, tile=tile, **kw)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20): ops.gemm(a, b, trans_b=tb, tile=tile, **kw)
            e.record(); torch.cuda.synchronize()
            res[label].append(s.elapsed_time(e) / 20 * 1e3)
    print(f"{name:44s} identical={same} ", {k: f"{sorted(v)[len(v)//2]:.1f}" for k, v in res.items()}, flush=True)
ops.lib.dw_debug_set(20, 36)
