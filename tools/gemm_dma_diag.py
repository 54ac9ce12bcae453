"""What the operand DMA of the K loop costs, split into issue + LDS write (DMA that always hits L2: dw_debug_set(19, 4)) and
latency of L2 misses (the rest).  256-row NN kernels, with and without epilogue
(key 11 = 1 / 17).  Interleaved rounds in one process; TFLOP/s medians."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
B = 32
def rnd(shape, s=1.0): return (torch.randn(shape, device="cuda") * s).bfloat16()
shapes = [("N=1280 K=1280", B*1500, 1280, 1280), ("N=3840 K=1280", B*1500, 3840, 1280), ("N=1280 K=5120", B*1500, 1280, 5120),
          ("N=5120 K=1280", B*1500, 5120, 1280)]
ops.lib.dw_debug_set(0, 115)          # 256-row kernels only (the experiment kernels are 256-row)
variants = [("base", 0), ("no-dma", 2), ("l2-warm-dma", 4)]
for name, M, N, K in shapes:
    a = rnd((M, K)); b = rnd((N, K), 0.05)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.lib.dw_debug_set(19, 0)
    ref = ops.gemm(a, b, tile=256).clone()
    res = {}
    for r in range(5):
        for epi in (1, 17):
            ops.lib.dw_debug_set(11, epi)
            for vn, v in variants:
                ops.lib.dw_debug_set(19, v)
                for _ in range(2): ops.gemm(a, b, out=out, tile=256)
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(10): ops.gemm(a, b, out=out, tile=256)
                e.record(); torch.cuda.synchronize()
                res.setdefault((vn, epi), []).append(2.0*M*N*K/(s.elapsed_time(e)/10*1e-3)/1e12)
    print(name, {f"{vn}{'' if epi == 1 else ' noepi'}": f"{sorted(r)[len(r)//2]:.0f}" for (vn, epi), r in res.items()}, flush=True)
ops.lib.dw_debug_set(19, 0); ops.lib.dw_debug_set(11, 1); ops.lib.dw_debug_set(0, 2163)
