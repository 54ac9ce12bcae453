"""One GEMM shape, a few launches: target of rocprofv3 --pmc passes (tools/pmc_dump.py prints the per-kernel averages)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
if os.environ.get("DW_V"): ops.lib.dw_debug_set(0, int(os.environ["DW_V"]))
M, N, K = 48000, int(os.environ.get("N", 5120)), int(os.environ.get("K", 1280))
ta = tb = bool(int(os.environ.get("TT", "0")))
a = (torch.randn((K, M) if ta else (M, K), device="cuda")).bfloat16()
b = (torch.randn((K, N) if tb else (N, K), device="cuda") * 0.05).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(6):
    ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out, tile=int(os.environ.get("TILE", "256")))
torch.cuda.synchronize()
