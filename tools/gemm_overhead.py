"""Per-tile fixed cost (prologue + epilogue) and per-K-tile cost of the persistent 256x256 GEMM: a K sweep at fixed M, N.
time per launch / tile rounds = a + b * (K / 64); a is what the matrix pipe idles through on every tile."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
M = int(os.environ.get("DW_M", 48000))
def rnd(shape, s=1.0): return (torch.randn(shape, device="cuda") * s).bfloat16()
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n * 1e3)
    return best
Ks = [64, 128, 256, 640, 1280, 2560, 5120]
for N in (1280, 3840):
    tiles = math.ceil(M / 256) * math.ceil(N / 256)
    rounds = math.ceil(tiles / 256)
    bias = torch.randn(N, device="cuda")
    res32 = torch.randn(M, N, device="cuda")
    for flavour in os.environ.get("DW_FL", "plain,bias,bias+gelu,bias+res_f32").split(","):
        for key11 in [int(x) for x in os.environ.get("DW_K11", "1,0").split(",")]:
            ops.lib.dw_debug_set(11, key11 & 255)
            ops.lib.dw_debug_set(0, 115 | (((key11 >> 8) & 3) << 9))     # key11 bits 8-9: main-loop ablation (1 = no fragment reads, 2 = no operand DMA)
            row = []
            for K in Ks:
                a = rnd((M, K)); b = rnd((N, K), 0.05)
                if flavour == "plain":
                    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
                    fn = lambda: ops.gemm(a, b, out=out, tile=256)
                elif flavour == "bias":
                    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
                    fn = lambda: ops.gemm(a, b, bias=bias, out=out, tile=256)
                elif flavour == "bias+gelu":
                    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
                    fn = lambda: ops.gemm(a, b, bias=bias, act=1, out=out, tile=256)
                else:
                    out = torch.empty(M, N, device="cuda", dtype=torch.float32)
                    fn = lambda: ops.gemm(a, b, bias=bias, residual=res32, out_dtype=torch.float32, out=out, tile=256)
                row.append(timeit(fn))
            per_round = [t / rounds for t in row]
            # least squares on the last four points (K >= 640) and on all
            xs = [k / 64 for k in Ks]
            n = len(xs); sx = sum(xs); sy = sum(per_round); sxx = sum(x * x for x in xs); sxy = sum(x * y for x, y in zip(xs, per_round))
            bb = (n * sxy - sx * sy) / (n * sxx - sx * sx); aa = (sy - bb * sx) / n
            print(f"N={N} {flavour:13s} stage_next={key11} rounds={rounds} us/launch " + " ".join(f"K{k}:{t:.0f}" for k, t in zip(Ks, row)) +
                  f" | per tile: a={aa:.2f} us  b={bb:.3f} us/Ktile (ideal 0.853 at 2.4 GHz)  K=1280 TF/s={2.0*M*N*1280/row[4]/1e6:.0f}", flush=True)
ops.lib.dw_debug_set(11, 1); ops.lib.dw_debug_set(0, 2163)
