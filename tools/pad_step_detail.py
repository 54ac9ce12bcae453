"""Per-shape GEMM times of one instrumented step with and without the row pitch pad of the FFN-wide buffers."""
import os, sys, torch, json
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
dec_in = ids[:, :-1].contiguous(); labels = ids[:, 1:].clone()
lens = torch.randint(32, 225, (B,), generator=torch.Generator().manual_seed(1234)).tolist()
labels[torch.arange(T, device=dev)[None, :] >= torch.tensor(lens, device=dev)[:, None]] = -100
def step(): return tr.train_step(tr.features(audio), dec_in, labels, valid_len=lens)
res = {}
pads = eval(os.environ.get("PADS", "[(0, 0), (64, 0)]"))
for pad in pads + pads:
    type(tr.student).ffn_row_pad = type(tr.teacher).ffn_row_pad = pad[0]
    type(tr.student).row_pad = type(tr.teacher).row_pad = pad[1]
    type(tr.student).stream_row_pad = type(tr.teacher).stream_row_pad = pad[2] if len(pad) > 2 else 0
    type(tr.student).dx_row_pad = type(tr.teacher).dx_row_pad = pad[3] if len(pad) > 3 else 0
    ops.wgrad_slab_pad = pad[4] if len(pad) > 4 else 0
    step(); step(); torch.cuda.synchronize()
    ops.profile, ops.profile_detail = {}, True
    step(); torch.cuda.synchronize()
    prof = ops.collect_profile(); ops.profile = None; ops.profile_detail = False
    for k, v in prof.items():
        res.setdefault(k, {}).setdefault(pad, []).append(v["ms"])
tot = {p: 0.0 for p in pads}
rows = []
for k, d in res.items():
    ms = {p: min(d.get(p, [0.0])) for p in pads}
    for p in pads: tot[p] += ms[p]
    rows.append((max(ms.values()), k, ms))
for _, k, ms in sorted(rows, reverse=True)[:45]:
    print(f"{k:70s} " + "  ".join(f"{p}: {ms[p]:7.2f}" for p in pads))
print("total", tot)
