import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
B, H, L = 32, 20, 1500
q, k, v = [(torch.randn(B * L, H * 64, device="cuda")).bfloat16() for _ in range(3)]
for _ in range(3):
    o, lse = ops.attn_fwd(q, k, v, B, H, L, L, False, 0.125)
do = torch.randn_like(o)
for _ in range(3):
    ops.attn_bwd(q, k, v, o, do, lse, B, H, L, L, False, 0.125)
torch.cuda.synchronize()
