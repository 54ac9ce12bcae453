"""A/B of the LayerNorm variants (dw_debug_set key 21) at the encoder shape [48000 x 1280]: us per launch over buffers rotated
past the Infinity Cache, results compared with the default kernels'."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
R, D = 48000, 1280
NB = 4
def timed(fn, n=16):
    for i in range(NB): fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn(i % NB)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
g, b = torch.rand(D, device="cuda") + 0.5, torch.randn(D, device="cuda")
xs = [torch.randn(R, D, device="cuda") * 2 + 0.3 for _ in range(NB)]
ys = [torch.empty(R, D, device="cuda", dtype=torch.bfloat16) for _ in range(NB)]
dys = [torch.randn(R, D, device="cuda").bfloat16() for _ in range(NB)]
lows = [torch.empty(R, D, device="cuda", dtype=torch.bfloat16) for _ in range(NB)]
mean = [x.mean(1) for x in xs]; rstd = [torch.rsqrt(x.var(1, unbiased=False) + 1e-5) for x in xs]
dres = [torch.randn(R, D, device="cuda") for _ in range(NB)]
dres0 = dres[0].clone()
fwd_variants = [0, 1, 1 | (4 << 8), 1 | (6 << 8), 1 | (8 << 8)]
bwd_variants = [0, 2]
ref = None
for rnd in range(2):
    for v in fwd_variants:
        ops.lib.dw_debug_set(21, v)
        y, m, r = ops.layernorm_fwd(xs[0], g, b, 1e-5, save_stats=True)
        if v == 0: ref = (y.clone(), m.clone(), r.clone())
        ok = torch.equal(y, ref[0]) and torch.equal(m, ref[1]) and torch.equal(r, ref[2])
        t = timed(lambda i: ops.layernorm_fwd(xs[i], g, b, 1e-5, save_stats=False, out=ys[i]))
        print(f"ln_fwd f32 variant {v:5d}: {t * 1e6:6.1f} us  {(R * D * 6) / t / 1e12:.2f} TB/s  identical to default: {ok}", flush=True)
for rnd in range(2):
    for v in bwd_variants:
        ops.lib.dw_debug_set(21, v)
        dg, db, cs = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
        d0 = dres0.clone(); lo = torch.empty_like(lows[0])
        ops.layernorm_bwd(dys[0], xs[0], mean[0], rstd[0], g, d0, dg, db, out_lowp=lo, colsum=cs)
        if v == 0: refb = (d0.clone(), lo.clone(), dg.clone(), db.clone(), cs.clone())
        errs = [((a - b_).abs().max() / (b_.abs().max() + 1e-30)).item() for a, b_ in zip((d0, lo.float(), dg, db, cs), (refb[0], refb[1].float(), refb[2], refb[3], refb[4]))]
        dg2, db2, cs2 = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
        t = timed(lambda i: ops.layernorm_bwd(dys[i], xs[i], mean[i], rstd[i], g, dres[i], dg2, db2, out_lowp=lows[i], colsum=cs2))
        print(f"ln_bwd variant {v}: {t * 1e6:6.1f} us  {(R * D * 16) / t / 1e12:.2f} TB/s  max rel diff vs default (dres, lowp, dgamma, dbeta, colsum): "
              + " ".join(f"{e:.1e}" for e in errs), flush=True)
ops.lib.dw_debug_set(21, 3)
