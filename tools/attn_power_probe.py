"""Clock / socket power under a sustained stream of the attention kernels (encoder shape), and their rates."""
import os, subprocess, sys, threading, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
B, H, L, D = 32, 20, 1500, 1280
qkv = torch.randn(B * L, 3 * D, device="cuda").bfloat16()
q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
o, lse = ops.attn_fwd(q, k, v, B, H, L, L, False, 0.125)
do = torch.randn(B * L, D, device="cuda").bfloat16()
FL = 4.0 * B * H * L * L * 64
def poll(samples, stop):
    while not stop[0]:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            pw = [l.split(":")[-1].strip() for l in out.splitlines() if "Power (W)" in l][:1]
            sc = [l.split("(")[-1].split(")")[0] for l in out.splitlines() if "sclk" in l][:1]
            samples.append((pw, sc))
        except Exception as e:  # noqa
            samples.append((str(e)[:40], ""))
        time.sleep(0.25)
def run(name, fn, flops, secs=4.0):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    samples, stop = [], [False]
    th = threading.Thread(target=poll, args=(samples, stop)); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(20): fn()
        torch.cuda.synchronize(); n += 20
    dt = time.time() - t0
    stop[0] = True; th.join()
    print(f"{name}: {dt/n*1e6:.0f} us per call, {flops*n/dt/1e12:.0f} TFLOP/s; (power W / sclk): " +
          " ".join(f"{p[0] if p else '?'}/{c[0] if c else '?'}" for p, c in samples[2:10]), flush=True)
run("attn fwd (1500x1500, B=32, H=20)", lambda: ops.attn_fwd(q, k, v, B, H, L, L, False, 0.125), FL)
run("attn bwd (delta + dQ + dK/dV)", lambda: ops.attn_bwd(q, k, v, o, do, lse, B, H, L, L, False, 0.125), 2.5 * FL)
