"""In-process A/B of the attention backward variants (dw_debug_set key 3: bit 1 = dK/dV fast staging, bit 2 = dK/dV at
3 waves per SIMD) at the encoder shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
B, H, L, D = 32, 20, 1500, 1280
qkv = torch.randn(B * L, 3 * D, device="cuda").bfloat16()
q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
o, lse = ops.attn_fwd(q, k, v, B, H, L, L, False, 0.125)
do = torch.randn(B * L, D, device="cuda").bfloat16()
res = {}
for rnd in range(3):
    for mode in (1, 3, 5, 7):
        ops.lib.dw_debug_set(3, mode)
        for _ in range(2): ops.attn_bwd(q, k, v, o, do, lse, B, H, L, L, False, 0.125)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.attn_bwd(q, k, v, o, do, lse, B, H, L, L, False, 0.125)
        e1.record(); torch.cuda.synchronize()
        res.setdefault(mode, []).append(e0.elapsed_time(e1) / 10)
for m, t in res.items():
    print("stage mode", m, "ms per attn_bwd:", " ".join(f"{x:.3f}" for x in t))
