"""Ceiling only: torch's scaled_dot_product_attention (whatever flash backend this ROCm build picks) against this
library's attention kernels on the three shape classes of the step, forward and backward, interleaved in one process on
the same random data.  Not a product path -- it says how far the hand-written kernels are from what the vendor stack
reaches on this box under the same power cap.  Output: one JSON line per shape class (TFLOP/s, algorithmic flops:
fwd 4 B H Lq Lk 64, bwd 2.5x; causal counted full like SURVEY 8d)."""
import json, os, sys, time, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps

ops = HipOps("cuda:0")
B, H, D = 32, 20, 64
shapes = [("enc self 1500x1500", 1500, 1500, False), ("dec self 447x447 causal", 447, 447, True), ("cross 447x1500", 447, 1500, False)]

def timed(fn, n):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

out = []
for name, Lq, Lk, causal in shapes:
    n = 10 if Lq > 1000 else 30
    q2 = torch.randn(B * Lq, H * D, device="cuda").bfloat16()
    k2 = torch.randn(B * Lk, H * D, device="cuda").bfloat16()
    v2 = torch.randn(B * Lk, H * D, device="cuda").bfloat16()
    o, lse = ops.attn_fwd(q2, k2, v2, B, H, Lq, Lk, causal, 0.125)
    do2 = torch.randn_like(o)
    mine_f = lambda: ops.attn_fwd(q2, k2, v2, B, H, Lq, Lk, causal, 0.125, out=o)
    dq, dk, dv = torch.empty_like(q2), torch.empty_like(k2), torch.empty_like(v2)
    mine_b = lambda: ops.attn_bwd(q2, k2, v2, o, do2, lse, B, H, Lq, Lk, causal, 0.125, dq=dq, dk=dk, dv=dv)
    # torch layout [B, H, L, D] (contiguous copies: the vendor kernels get their preferred layout)
    q4 = q2.view(B, Lq, H, D).transpose(1, 2).contiguous().requires_grad_(True)
    k4 = k2.view(B, Lk, H, D).transpose(1, 2).contiguous().requires_grad_(True)
    v4 = v2.view(B, Lk, H, D).transpose(1, 2).contiguous().requires_grad_(True)
    do4 = do2.view(B, Lq, H, D).transpose(1, 2).contiguous()
    rec = {"shape": name, "B": B, "H": H, "Lq": Lq, "Lk": Lk, "causal": causal}
    flops_f = 4.0 * B * H * Lq * Lk * D
    backends = {}
    try:
        from torch.nn.attention import sdpa_kernel, SDPBackend
        cand = [("flash", SDPBackend.FLASH_ATTENTION), ("efficient", SDPBackend.EFFICIENT_ATTENTION)]
    except Exception as ex:          # noqa
        cand = []
    for bname, be in cand:
        try:
            with sdpa_kernel([be]):
                with torch.no_grad():
                    ref = F.scaled_dot_product_attention(q4, k4, v4, is_causal=causal, scale=0.125)
                    tf = timed(lambda: F.scaled_dot_product_attention(q4, k4, v4, is_causal=causal, scale=0.125), n)
                o4 = F.scaled_dot_product_attention(q4, k4, v4, is_causal=causal, scale=0.125)
                def bwd():
                    q4.grad = k4.grad = v4.grad = None
                    o4.backward(do4, retain_graph=True)
                tb = timed(bwd, n)
            diff = (ref.transpose(1, 2).reshape(B * Lq, H * D).float() - o.float()).abs().max().item()
            backends[bname] = {"fwd_tflops": round(flops_f / tf / 1e9, 1), "bwd_tflops": round(2.5 * flops_f / tb / 1e9, 1),
                               "fwd_ms": round(tf, 4), "bwd_ms": round(tb, 4), "max_abs_diff_vs_mine": diff}
        except Exception as ex:
            backends[bname] = {"error": str(ex)[:200]}
    tf = timed(mine_f, n); tb = timed(mine_b, n)
    rec["mine"] = {"fwd_tflops": round(flops_f / tf / 1e9, 1), "bwd_tflops": round(2.5 * flops_f / tb / 1e9, 1),
                   "fwd_ms": round(tf, 4), "bwd_ms": round(tb, 4)}
    rec["vendor"] = backends
    print(json.dumps(rec), flush=True)
    out.append(rec)
    del q4, k4, v4, do4
if len(sys.argv) > 1:        # -> profiles/attn_vendor_ceiling.json (bench.py quotes it in `roofline`)
    from distil_whisper_amd.build import kernels_sha16
    json.dump({"note": "ceiling only: torch.nn.functional.scaled_dot_product_attention, torch " + torch.__version__ +
               ", same box, same random data, interleaved with this library's kernels", "kernels_sha16": kernels_sha16(),
               "classes": out}, open(sys.argv[1], "w"), indent=1)
