"""HBM-bound kernels of the step at their big shapes: GB/s of algorithmic bytes (each timed over buffers larger than the
256 MiB Infinity Cache in rotation, so that reads come from HBM as they do in the step)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
R, D, F = 48000, 1280, 5120
NB = 4   # rotating buffer sets
def timed(fn, n=12):
    for i in range(NB): fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn(i % NB)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
g, b = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
xs = [torch.randn(R, D, device="cuda") for _ in range(NB)]
ys = [torch.empty(R, D, device="cuda", dtype=torch.bfloat16) for _ in range(NB)]
t = timed(lambda i: ops.layernorm_fwd(xs[i], g, b, 1e-5, save_stats=False, out=ys[i]))
print(f"ln_fwd f32 [{R}x{D}]: {t * 1e6:.0f} us, {(R * D * 6) / t / 1e12:.2f} TB/s")
xb = [x.bfloat16() for x in xs]
t = timed(lambda i: ops.layernorm_fwd(xb[i], g, b, 1e-5, save_stats=False, out=ys[i]))
print(f"ln_fwd bf16 [{R}x{D}]: {t * 1e6:.0f} us, {(R * D * 4) / t / 1e12:.2f} TB/s")
zs = [torch.randn(R, F, device="cuda").bfloat16() for _ in range(NB)]
out = torch.zeros(F, device="cuda")
t = timed(lambda i: ops.colsum(zs[i], out, accumulate=True))
print(f"colsum bf16 [{R}x{F}]: {t * 1e6:.0f} us, {(R * F * 2) / t / 1e12:.2f} TB/s")
t = timed(lambda i: ops.colsum(ys[i], out[:D], accumulate=True))
print(f"colsum bf16 [{R}x{D}]: {t * 1e6:.0f} us, {(R * D * 2) / t / 1e12:.2f} TB/s")
mean = [x.mean(1) for x in xs]; rstd = [torch.rsqrt(x.var(1, unbiased=False) + 1e-5) for x in xs]
dres = [torch.randn(R, D, device="cuda") for _ in range(NB)]
dg, db, cs = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
t = timed(lambda i: ops.layernorm_bwd(ys[i], xs[i], mean[i], rstd[i], g, dres[i], dg, db, out_lowp=ys[(i + 1) % NB], colsum=cs))
# bytes: dy bf16 2 + x f32 4 + dres read 4 + dres write 4 + bf16 copy 2 = 16 per element
print(f"ln_bwd (accumulate + bf16 copy + colsum) [{R}x{D}]: {t * 1e6:.0f} us, {(R * D * 16) / t / 1e12:.2f} TB/s")
