"""Single-query attention over 1500 cached keys (the token step's cross-attention) with the K | V rows padded by P elements."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
B, H, L, D = 16, 20, 1500, 1280
def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
q = torch.randn(B, D, device="cuda").bfloat16()
for P in (0, 64, 128, 192, 32):
    # two layers' worth of K/V so that consecutive launches do not find their stream in the caches
    kvs = [torch.randn(B * L, 2 * D + P, device="cuda").bfloat16() for _ in range(4)]
    i = [0]
    def fn():
        kv = kvs[i[0] % 4]; i[0] += 1
        ops.attn_fwd(q, kv[:, :D], kv[:, D:2 * D], B, H, 1, L, False, 0.125)
    t = sorted(timed(fn) for _ in range(3))[1]
    print(f"pad {P:3d}: {t:6.1f} us  {B * L * 2 * D * 2 / t / 1e6:5.2f} TB/s", flush=True)
