"""Sample socket power / sclk with rocm-smi while the distillation step runs (is the step power/clock limited?)."""
import os, subprocess, sys, threading, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
from distil_whisper_amd.distill import DistillationTrainer
from distil_whisper_amd import student_init as si
dev = "cuda:0"
ops = HipOps(dev)
tdims = si.PRESETS["large-v3"]
t_sd = si.random_state_dict(tdims, 0, dev)
s_sd, sdims = si.student_from_teacher(t_sd, tdims, 32, 2)
filt = torch.tensor(si.mel_filter_bank(128), dtype=torch.float32, device=dev).contiguous()
tr = DistillationTrainer(ops, s_sd, sdims, t_sd, tdims, mel_filters=filt)
del t_sd, s_sd
B, T = 32, 447
audio = 0.1 * torch.randn(B, 480000, device=dev)
ids = torch.randint(0, 50257, (B, T + 1), device=dev); ids[:, 0] = 50258
dec_in = ids[:, :-1].contiguous(); labels = ids[:, 1:].clone(); labels[:, 200:] = -100
stop = False
samples = []
def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            pw = [l.split(":")[-1].strip() for l in out.splitlines() if "Power (W)" in l]
            sc = [l.split("(")[-1].split(")")[0] for l in out.splitlines() if "sclk" in l]
            samples.append((time.time(), pw[:1], sc[:1]))
        except Exception as e:  # noqa
            samples.append((time.time(), str(e)[:40], ""))
        time.sleep(0.3)
tr.train_step(tr.features(audio), dec_in, labels); torch.cuda.synchronize()
th = threading.Thread(target=poll); th.start()
t0 = time.time()
n = 0
while time.time() - t0 < 12:
    tr.train_step(tr.features(audio), dec_in, labels); n += 1
torch.cuda.synchronize()
dt = time.time() - t0
stop = True; th.join()
print("steps", n, "ms/step", dt / n * 1e3)
for s in samples: print(round(s[0] - t0, 1), s[1], s[2])
