"""Per-shape / per-epilogue breakdown of the GEMM launches of one full distillation step (HIP events per launch).
Writes a markdown table: which GEMM flavours of the step run furthest below the kernel's best rate."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
from distil_whisper_amd.distill import DistillationTrainer
from distil_whisper_amd import student_init as si
dev = "cuda:0"
ops = HipOps(dev)
MODEL = os.environ.get("MODEL", "large-v3")            # large-v3 | small.en | tiny.en (BASELINE configs 3 / 2 / 1)
tdims = si.PRESETS[MODEL]
t_sd = si.random_state_dict(tdims, 0, dev)
s_sd, sdims = si.student_from_teacher(t_sd, tdims, *si.STUDENT_LAYERS[MODEL])
filt = torch.tensor(si.mel_filter_bank(tdims.n_mels), dtype=torch.float32, device=dev).contiguous()
tr = DistillationTrainer(ops, s_sd, sdims, t_sd, tdims, mel_filters=filt)
del t_sd, s_sd
B, T = int(os.environ.get("B", 32)), 447
audio = 0.1 * torch.randn(B, 480000, device=dev)
ids = torch.randint(0, 50257, (B, T + 1), device=dev); ids[:, 0] = tdims.decoder_start_token_id
dec_in = ids[:, :-1].contiguous(); labels = ids[:, 1:].clone()
# DW_LENS=1: bench.py's batch (label lengths U{32..224}, the step over the live decoder rows only)
if os.environ.get("DW_LENS"):
    lens = torch.randint(32, 225, (B,), generator=torch.Generator().manual_seed(1234)).tolist()
    labels[torch.arange(T, device=dev)[None, :] >= torch.tensor(lens, device=dev)[:, None]] = -100
else:
    lens = None
    labels[:, 200:] = -100
def step():
    return tr.train_step(tr.features(audio), dec_in, labels, valid_len=lens)
ops.lib.dw_debug_set(0, int(os.environ.get('DW_VARIANT', 2163)))  # 2163 = the library default (dw_debug_set key 0)
for _ in range(2): step()
torch.cuda.synchronize()
ops.profile_detail = True
acc = {}
for _ in range(2):
    ops.profile = {}
    step()
    for k, d in ops.collect_profile().items():
        a = acc.setdefault(k, {"n": 0, "ms": 0.0, "flops": 0.0})
        a["n"] += d["n"]; a["ms"] += d["ms"]; a["flops"] += d["flops"]
ops.profile = None
rows = sorted(acc.items(), key=lambda kv: -kv[1]["ms"])
tot = sum(d["ms"] for _, d in rows) / 2
print(f"# per-flavour launch times of one {MODEL} step (B={B}); instrumented step = {tot:.1f} ms\n")
print("| launch | calls/step | ms/step | avg us | TFLOP/s |\n|---|---|---|---|---|")
for k, d in rows:
    tf = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["flops"] else 0.0
    print(f"| {k} | {d['n'] // 2} | {d['ms'] / 2:.2f} | {d['ms'] / d['n'] * 1e3:.1f} | {tf:.0f} |" if tf else
          f"| {k} | {d['n'] // 2} | {d['ms'] / 2:.2f} | {d['ms'] / d['n'] * 1e3:.1f} | |")
