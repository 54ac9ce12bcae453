"""Clock and socket power under a sustained GEMM stream: the normal kernel, the main-loop ablations of gemm_wp8_dbg.hip
(MFMA only / no fragment reads / no operand DMA) and the vendor library.  Is the matrix pipe clock-throttled at full duty?"""
import os, subprocess, sys, threading, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
ops = HipOps("cuda:0")
M, N, K = 48000, 3840, int(os.environ.get("K", 1280))
a = torch.randn(M, K, device="cuda").bfloat16(); b = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
def poll(samples, stop):
    while not stop[0]:
        try:
            o = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            pw = [l.split(":")[-1].strip() for l in o.splitlines() if "Power (W)" in l][:1]
            sc = [l.split("(")[-1].split(")")[0] for l in o.splitlines() if "sclk" in l][:1]
            samples.append((pw, sc))
        except Exception as e:  # noqa
            samples.append((str(e)[:40], ""))
        time.sleep(0.25)
def run(name, fn, secs=5.0):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    samples, stop = [], [False]
    th = threading.Thread(target=poll, args=(samples, stop)); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(50): fn()
        torch.cuda.synchronize(); n += 50
    dt = time.time() - t0
    stop[0] = True; th.join()
    print(f"{name}: {2.0*M*N*K*n/dt/1e12:.0f} TFLOP/s sustained; (power W, sclk): " + " ".join(f"{p[0] if p else '?'}/{c[0] if c else '?'}" for p, c in samples[2:14]), flush=True)
def mine(v, k11):
    def f():
        ops.gemm(a, b, out=out, tile=256)
    ops.lib.dw_debug_set(0, v); ops.lib.dw_debug_set(11, k11)
    return f
for name, v, k11 in (("full kernel", 115, 1), ("no epilogue", 115, 17), ("MFMA only, no epilogue", 115 | 1536, 17),
                     ("no DMA, no epilogue", 115 | 1024, 17), ("no fragment reads, no epilogue", 115 | 512, 17)):
    run(name, mine(v, k11))
ops.lib.dw_debug_set(0, 2163); ops.lib.dw_debug_set(11, 1)
out2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
run("vendor (torch.matmul)", lambda: torch.matmul(a, b.t(), out=out2))
z = torch.zeros_like(a); zb = torch.zeros_like(b)
a.copy_(z); b.copy_(zb)
run("full kernel, zero operands", mine(115, 1))
run("MFMA only, zero operands", mine(115 | 1536, 17))
ops.lib.dw_debug_set(0, 2163); ops.lib.dw_debug_set(11, 1)
