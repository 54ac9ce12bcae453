"""v_mfma_f32_16x16x32_bf16 main loop (gemm_wp16.h, dw_debug_set(20, mask)) against the 32x32x16 one: agreement of the
results (fp32 rounding of a different summation tree, checked against an fp32 torch matmul) and TFLOP/s, interleaved rounds."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
B = 32
M = B * 1500
def rnd(shape, s=1.0): return (torch.randn(shape, device="cuda") * s).bfloat16()
bias = {n: torch.randn(n, device="cuda") for n in (1280, 3840, 5120)}
res32 = torch.randn(M, 1280, device="cuda")
cases = [("NN qkv bias N=3840 K=1280", M, 3840, 1280, False, False, dict(bias=bias[3840])),
         ("NN out bias+res32 N=1280 K=1280", M, 1280, 1280, False, False, dict(bias=bias[1280], residual=res32, out_dtype=torch.float32)),
         ("NN fc1 bias+gelu N=5120 K=1280", M, 5120, 1280, False, False, dict(bias=bias[5120], act=1)),
         ("NN fc2 plain N=1280 K=5120", M, 1280, 5120, False, False, {}),
         ("NN ragged M=14304 N=1280 K=1280 (256-row)", 14304 + 7, 1280, 1280, False, False, dict(tile=256)),
         ("NT dX qkv N=1280 K=3840", M, 1280, 3840, False, True, {}), ("NT dX fc2 N=5120 K=1280", M, 5120, 1280, False, True, {}),
         ("TT dW fc1 5120x1280 K=48000", 5120, 1280, M, True, True, dict(atomic_acc=True, out_dtype=torch.float32))]
only = os.environ.get("DW_ONLY")
ops.lib.dw_debug_set(11, int(os.environ.get("DW_K11", "1")))
for name, Mm, N, K, ta, tb, kw in cases:
    if only and only not in name: continue
    a = rnd((K, Mm) if ta else (Mm, K)); b = rnd((K, N) if tb else (N, K), 0.05)
    kw = dict(kw)
    acc = kw.pop("atomic_acc", False)
    def run(out=None):
        if acc:
            o = torch.zeros(Mm, N, device="cuda", dtype=torch.float32) if out is None else out
            return ops.gemm(a, b, trans_a=ta, trans_b=tb, out=o, atomic_acc=True)
        return ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out, **kw)
    ops.lib.dw_debug_set(20, 0); ref = run().float().clone()
    ops.lib.dw_debug_set(20, 7); new = run().float().clone()
    torch.cuda.synchronize()
    d = (new - ref).abs().max().item()
    scale = ref.abs().max().item()
    bad = int((~torch.isfinite(new)).sum().item())
    res = {0: [], 7: []}
    out = torch.empty_like(run()) if not acc else torch.zeros(Mm, N, device="cuda", dtype=torch.float32)
    for r in range(5):
        for v in (0, 7):
            ops.lib.dw_debug_set(20, v)
            for _ in range(2): run(out)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): run(out)
            e.record(); torch.cuda.synchronize()
            res[v].append(2.0 * Mm * N * K / (s.elapsed_time(e) / 10 * 1e-3) / 1e12)
    print(f"{name:44s} 32x32x16 {sorted(res[0])[2]:6.0f}   16x16x32 {sorted(res[7])[2]:6.0f} TFLOP/s   max|diff| {d:.3g} of {scale:.3g}  nonfinite {bad}", flush=True)
ops.lib.dw_debug_set(20, 0)
