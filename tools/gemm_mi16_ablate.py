import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
M, N, K = 48000, 3840, 1280
a = torch.randn(M, K, device="cuda").bfloat16(); b = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
ops.lib.dw_debug_set(0, 115); ops.lib.dw_debug_set(11, 17)      # 256-row kernels, no epilogue
res = {}
for r in range(4):
    for mi in (0, 1):
        for dbg in (0, 1, 2, 3):
            ops.lib.dw_debug_set(20, mi); ops.lib.dw_debug_set(19, dbg)
            for _ in range(2): ops.gemm(a, b, out=out, tile=256)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): ops.gemm(a, b, out=out, tile=256)
            e.record(); torch.cuda.synchronize()
            res.setdefault((mi, dbg), []).append(2.0*M*N*K/(s.elapsed_time(e)/10*1e-3)/1e12)
names = {0: "full", 1: "no frag reads", 2: "no DMA", 3: "MFMA only"}
for mi in (0, 1):
    print("16x16x32" if mi else "32x32x16", {names[d]: f"{sorted(res[(mi, d)])[2]:.0f}" for d in (0, 1, 2, 3)})
ops.lib.dw_debug_set(19, 0); ops.lib.dw_debug_set(20, 0); ops.lib.dw_debug_set(11, 1); ops.lib.dw_debug_set(0, 2163)
