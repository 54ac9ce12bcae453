"""Where the four-wave kernel's output differs from the eight-wave kernel's (pattern over the 256 x 256 block tile)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
M, N, K = 1024, 512, int(os.environ.get("DW_K", "256"))
a = (torch.randn(M, K, device="cuda")).bfloat16(); b = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
outs = {}
for label, mi in (("8w", 4), ("4w", 12)):
    ops.lib.dw_debug_set(0, 115); ops.lib.dw_debug_set(20, mi)
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm(a, b, out=out, tile=256)
    torch.cuda.synchronize()
    outs[label] = out.float()
ref = a.float() @ b.float().t()
for k, v in outs.items(): print(k, "max|out - fp32 ref|", (v - ref).abs().max().item())
bad = (outs["8w"] != outs["4w"])
print("mismatching elements", bad.sum().item(), "of", bad.numel())
t = bad.view(M // 256, 256, N // 256, 256).any(0).any(1)          # [256 rows of the tile, 256 cols]
blk = t.view(16, 16, 16, 16).any(1).any(2)                         # 16 x 16 blocks
for r in range(16): print("".join("X" if blk[r, c] else "." for c in range(16)))
ops.lib.dw_debug_set(0, 2163); ops.lib.dw_debug_set(20, 36)
