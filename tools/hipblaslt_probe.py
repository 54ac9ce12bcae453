"""Calibration: this library's bf16 GEMM against the vendor library (torch.matmul -> hipBLASLt) on the step's shapes,
interleaved in one process on the same buffers.  Not a product path: it only says how far the kernels are from what
the vendor's tuned assembly reaches on this box (same power cap, same clocks)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps

ops = HipOps("cuda:0")
B = 32
def rnd(shape, s=1.0): return (torch.randn(shape, device="cuda") * s).bfloat16()
# (name, M, N, K, trans_a, trans_b): C[M,N] = op(A) . op(B)^T in this library's convention (B stored [N,K] unless tb)
shapes = [("fwd qkv N=3840 K=1280", B*1500, 3840, 1280, False, False), ("fwd fc1 N=5120 K=1280", B*1500, 5120, 1280, False, False),
          ("fwd fc2 N=1280 K=5120", B*1500, 1280, 5120, False, False), ("fwd out N=1280 K=1280", B*1500, 1280, 1280, False, False),
          ("dX fc2 N=5120 K=1280", B*1500, 5120, 1280, False, True), ("dX fc1 N=1280 K=5120", B*1500, 1280, 5120, False, True),
          ("dW fc1 5120x1280 K=48000", 5120, 1280, B*1500, True, True), ("dW qkv 3840x1280 K=48000", 3840, 1280, B*1500, True, True),
          ("dec fc1 M=14304", B*447, 5120, 1280, False, False), ("lm head", B*447, 51904, 1280, False, False)]

def timed(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

records = {}
for name, M, N, K, ta, tb in shapes:
    a = rnd((K, M) if ta else (M, K)); b = rnd((K, N) if tb else (N, K), 0.05)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    out2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    A = a.t() if ta else a                  # [M, K] view
    Bm = b if tb else b.t()                 # [K, N] view
    if ta and tb:
        # weight-gradient form as the step runs it: fp32 accumulation into the flat gradient buffer, K cut into slices that
        # fill the 256 CUs + dw_reduce_slices (round-3 probe called the bf16-output form without slices: 100 tiles on 256 CUs,
        # which is where its 439-550 TFLOP/s against the step's 1 029 came from)
        out32 = torch.zeros(M, N, device="cuda", dtype=torch.float32)
        mine = lambda: ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out32, atomic_acc=True)
    else:
        mine = lambda: ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out, tile=256)
    vend = lambda: torch.matmul(A, Bm, out=out2)
    mine(); vend()
    d = ((out32 if ta and tb else out.float()) - out2.float()).abs().max().item()
    r = {"mine": [], "vendor": []}
    for _ in range(5):
        r["mine"].append(2.0 * M * N * K / (timed(mine) * 1e-3) / 1e12)
        r["vendor"].append(2.0 * M * N * K / (timed(vend) * 1e-3) / 1e12)
    med = {k: sorted(v)[len(v) // 2] for k, v in r.items()}
    print(f"{name:28s} mine {med['mine']:6.0f} TF/s   vendor {med['vendor']:6.0f} TF/s   max|diff| {d:.3g}", flush=True)
    records[name] = {"this_library_tflops": round(med["mine"]), "vendor_library_tflops": round(med["vendor"])}
if len(sys.argv) > 1:        # -> profiles/vendor_gemm_ceiling.json (bench.py reads row_major_forward_median_tflops)
    import json
    from distil_whisper_amd.build import kernels_sha16
    fwd = sorted(v["vendor_library_tflops"] for k, v in records.items() if k.startswith("fwd"))
    json.dump({"source": "tools/hipblaslt_probe.py on one MI355X box of the pool (torch.matmul -> hipBLASLt, bf16, interleaved with "
                         "this library on the same buffers; weight gradients through the step's split-K path)",
               "kernels_sha16": kernels_sha16(), "shapes": records,
               "row_major_forward_median_tflops": (fwd[len(fwd) // 2 - 1] + fwd[len(fwd) // 2]) / 2 if len(fwd) % 2 == 0 else fwd[len(fwd) // 2]},
              open(sys.argv[1], "w"), indent=1)
