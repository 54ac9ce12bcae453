"""One weight-gradient GEMM (dW = dY^T . X: both operands k-major, K = 48 000 tokens) and, for comparison, the forward
GEMM of the same layer: targets of rocprofv3 --pmc passes (tools/pmc_dump.py prints the per-kernel averages)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
R, F, D = 48000, 5120, 1280
dy = (torch.randn(R, F, device="cuda") * 0.05).bfloat16()
x = torch.randn(R, D, device="cuda").bfloat16()
w = (torch.randn(F, D, device="cuda") * 0.03).bfloat16()
g = torch.zeros(F, D, device="cuda")
out = torch.empty(R, F, device="cuda", dtype=torch.bfloat16)
for _ in range(4):
    ops.gemm(dy, x, trans_a=True, trans_b=True, out_dtype=torch.float32, out=g, atomic_acc=True)
    ops.gemm(x, w, out=out)
torch.cuda.synchronize()
