"""Where a tile's time goes inside the persistent GEMM workgroups: timestamps (s_memrealtime, 100 MHz) written by wave 0
of every workgroup for its first 8 tiles (dw_debug_set keys 13 / 14 = device pointer of the buffer): tile start, first
operand tile landed, K loop done, epilogue returned, behind the last barrier.  Medians over workgroups, tiles 1..6."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
M = 48000
def rnd(shape, s=1.0, dt=torch.bfloat16): return (torch.randn(shape, device="cuda") * s).to(dt)
cases = [("qkv bias (320-row)", 3840, 1280, False, lambda N: dict(bias=rnd((N,), 0.1, torch.float32))),
         ("out-proj teacher bias+res bf16 (256-row)", 1280, 1280, False, lambda N: dict(bias=rnd((N,), 0.1, torch.float32), residual=rnd((M, N)))),
         ("dX out plain (256-row)", 1280, 1280, True, lambda N: {}),
         ("fc1 teacher bias+gelu (320-row)", 5120, 1280, False, lambda N: dict(bias=rnd((N,), 0.1, torch.float32), act=1)),
         ("dX fc2 zg (256-row)", 5120, 1280, True, lambda N: dict(zgrad=rnd((M, N), 0.5, torch.float16)))]
trace = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda")
modes = [int(x) for x in os.environ.get("DW_K11", "1").split(",")]
cases = [c + (m,) for c in cases for m in modes]
for name, N, K, tb, mk, k11 in cases:
    ops.lib.dw_debug_set(11, k11)
    name = f"{name} [key11={k11}]"
    a = rnd((M, K)); b = rnd((K, N) if tb else (N, K), 0.05); kw = mk(N)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): ops.gemm(a, b, trans_b=tb, out=out, **kw)
    torch.cuda.synchronize()
    trace.zero_()
    ptr = trace.data_ptr()
    ops.lib.dw_debug_set(13, ptr & 0xffffffff if (ptr & 0xffffffff) < 2**31 else (ptr & 0xffffffff) - 2**32)
    ops.lib.dw_debug_set(14, ptr >> 32)
    ops.gemm(a, b, trans_b=tb, out=out, **kw)
    torch.cuda.synchronize()
    ops.lib.dw_debug_set(13, 0); ops.lib.dw_debug_set(14, 0)
    t = trace.view(256, 8, 8).double() * 0.01          # microseconds
    ok = t[:, 1:, 4] > 0                               # tiles 1.. of every workgroup that ran them (tile 0 starts cold)
    t = t[:, 1:][ok]
    seg = {"prologue (start -> first operand tile)": t[:, 1] - t[:, 0], "K loop": t[:, 2] - t[:, 1],
           "epilogue (wave 0)": t[:, 3] - t[:, 2], "  of it: to the leading barrier": t[:, 5] - t[:, 2],
           "  first slab": t[:, 6] - t[:, 5], "  other slabs": t[:, 3] - t[:, 6],
           "last barrier": t[:, 4] - t[:, 3], "tile": t[:, 4] - t[:, 0]}
    print(f"{name}: {int(ok.sum())} tiles; us per tile (median / p90): " +
          "; ".join(f"{k} {v.median():.2f} / {v.quantile(0.9):.2f}" for k, v in seg.items()), flush=True)
ops.lib.dw_debug_set(11, 1)
