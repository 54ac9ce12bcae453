"""Teacher-decoder GEMMs (M = packed live rows of the batch, ~4 100; bf16 residual stream): the 128-tile kernel with eight waves of
64 x 32 (default) against four waves of 64 x 64 (dw_debug_set key 24 = 1 plain K loop, 2 register double buffer).  us per launch
(median of interleaved rounds), TFLOP/s, and bit-identity with the default."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
def rnd(shape, s=1.0): return (torch.randn(shape, device="cuda") * s).bfloat16()
bias = {n: torch.randn(n, device="cuda") for n in (1280, 3840, 5120)}
cases = []
for M in (4107, 4224, 3500, 7136):
    res = rnd((M, 1280))
    cases += [(f"M={M} qkv N=3840 K=1280 bias", M, 3840, 1280, dict(bias=bias[3840])),
              (f"M={M} out N=1280 K=1280 bias+rbf16", M, 1280, 1280, dict(bias=bias[1280], residual=res)),
              (f"M={M} fc1 N=5120 K=1280 bias gelu", M, 5120, 1280, dict(bias=bias[5120], act=1)),
              (f"M={M} fc2 N=1280 K=5120 bias+rbf16", M, 1280, 5120, dict(bias=bias[1280], residual=res))]
for name, M, N, K, kw in cases:
    a = rnd((M, K)); b = rnd((N, K), 0.05)
    t = {0: [], 1: [], 2: []}
    outs = {}
    for r in range(5):
        for v in (0, 1, 2):
            ops.lib.dw_debug_set(24, v)
            for _ in range(2): o = ops.gemm(a, b, tile=128, **kw)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20): o = ops.gemm(a, b, tile=128, **kw)
            e.record(); torch.cuda.synchronize()
            t[v].append(s.elapsed_time(e) / 20 * 1e3)
            outs[v] = o
    med = {v: sorted(x)[len(x) // 2] for v, x in t.items()}
    fl = 2.0 * M * N * K
    print(f"{name:40s}", "  ".join(f"key24={v}: {med[v]:6.1f} us {fl / med[v] / 1e6:6.0f} TF/s" for v in (0, 1, 2)),
          " identical:", torch.equal(outs[0], outs[1]), torch.equal(outs[0], outs[2]), flush=True)
ops.lib.dw_debug_set(24, 0)
