"""Where the attention forward kernel's time goes: variants of the kernel that each leave out one resource user (and
compute garbage), timed at the encoder shape.  Needs a library built with DW_ABLATE=1 (python -m distil_whisper_amd.build
--force under that environment variable); dw_debug_set key 15 selects the variant."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
B, H, L, D = 32, 20, 1500, 1280
qkv = torch.randn(B * L, 3 * D, device="cuda").bfloat16()
q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
NAMES = {0: "full kernel", 1: "no exp", 2: "K fragments from registers", 4: "V^T fragments from registers", 6: "K and V^T from registers",
         8: "no operand staging in the loop", 24: "no staging, no barrier", 32: "no QK MFMAs", 64: "no PV MFMAs",
         7: "no exp, no fragment reads", 31: "no exp / reads / staging / barrier (MFMA + softmax arithmetic only)",
         30: "no reads / staging / barrier (MFMA + softmax with exp)"}
def timed(n=10):
    for _ in range(2): ops.attn_fwd(q, k, v, B, H, L, L, False, 0.125)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): ops.attn_fwd(q, k, v, B, H, L, L, False, 0.125)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
res = {}
for rnd in range(3):
    for a in NAMES:
        assert ops.lib.dw_debug_set(15, a) == 0
        res.setdefault(a, []).append(timed())
ops.lib.dw_debug_set(15, 0)
for a, t in res.items():
    print(f"{a:3d} {NAMES[a]:70s} us: " + " ".join(f"{x:.0f}" for x in t))
