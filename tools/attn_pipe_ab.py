"""Forward attention: attn_fwd_kernel against attn_fwd_pipe_kernel (dw_debug_set key 26), same process, interleaved rounds;
outputs and LSE compared bitwise.  Shapes: encoder self (1500 x 1500), cross (live decoder rows x 1500), a ragged one."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
H = 20
for B, Lq, Lk in ((32, 1500, 1500), (32, 223, 1500), (7, 333, 777)):
    q = torch.randn(B * Lq, H * 64, device="cuda").bfloat16()
    k, v = [torch.randn(B * Lk, H * 64, device="cuda").bfloat16() for _ in range(2)]
    outs = {}
    res = {0: [], 1: [], 2: []}
    for r in range(5):
        for key in (0, 1, 2):
            ops.lib.dw_debug_set(26, key)
            for _ in range(2): o, lse = ops.attn_fwd(q, k, v, B, H, Lq, Lk, False, 0.125)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): o, lse = ops.attn_fwd(q, k, v, B, H, Lq, Lk, False, 0.125)
            e.record(); torch.cuda.synchronize()
            res[key].append(s.elapsed_time(e) / 10 * 1e3)
            outs[key] = (o.clone(), lse.clone())
    same = all(torch.equal(outs[0][0], outs[k_][0]) and torch.equal(outs[0][1], outs[k_][1]) for k_ in (1, 2))
    fl = 4.0 * B * H * Lq * Lk * 64
    med = {k_: sorted(v_)[len(v_) // 2] for k_, v_ in res.items()}
    print(f"B={B} Lq={Lq} Lk={Lk} identical={same} max|do|={(outs[0][0].float() - outs[1][0].float()).abs().max().item():.3g} " +
          " ".join(f"key26={k_}: {v_:.1f} us ({fl / v_ / 1e6:.0f} TF/s)" for k_, v_ in med.items()), flush=True)
ops.lib.dw_debug_set(26, 0)
