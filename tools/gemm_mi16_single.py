import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
M, N, K = 48000, 3840, 1280
a = torch.randn(M, K, device="cuda").bfloat16(); b = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
ops.lib.dw_debug_set(0, 115)          # 256-row kernels
for v in (0, 7):
    ops.lib.dw_debug_set(20, v)
    for _ in range(4): ops.gemm(a, b, out=out, tile=256)
torch.cuda.synchronize()
