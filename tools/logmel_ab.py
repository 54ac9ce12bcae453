"""In-process A/B of the log-mel front end: direct DFT on the VALU (dw_debug_set(5, 0)) vs the folded DFT on the fp32
matrix pipe (dw_debug_set(5, 1)) on a batch of 30 s clips; prints ms per batch, GB/s of algorithmic traffic and the
largest difference between the two outputs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
from distil_whisper_amd import student_init as si
ops = HipOps("cuda:0")
B = int(os.environ.get("B", 32))
audio = 0.1 * torch.randn(B, 480000, device="cuda")
for M in (128, 80):
    filt = torch.tensor(si.mel_filter_bank(M), dtype=torch.float32, device="cuda").contiguous()
    outs, res = {}, {}
    for rnd in range(3):
        for mode in (0, 1):
            ops.lib.dw_debug_set(5, mode)
            outs[mode] = ops.logmel(audio, filt)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.logmel(audio, filt)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(mode, []).append(e0.elapsed_time(e1) / 10)
    nbytes = B * (480000 * 4 + M * 3000 * 4)
    for mode in (0, 1):
        t = sorted(res[mode])[1]
        print(f"mels {M} mode {mode}: {t:.3f} ms per batch of {B}  ->  {nbytes / t / 1e6:.0f} GB/s algorithmic")
    print("max |mfma - direct| =", (outs[0] - outs[1]).abs().max().item())
ops.lib.dw_debug_set(5, 1)
