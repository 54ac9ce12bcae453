"""Forward-only encoder pass (long-form / pseudo-labelling: B windows of 30 s) per launch flavour: where the 41 ms of a batch of 16 go."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
from distil_whisper_amd.modeling import WhisperForConditionalGeneration
from distil_whisper_amd import student_init as si
dev = "cuda:0"
ops = HipOps(dev)
tdims = si.PRESETS["large-v3"]
t_sd = si.random_state_dict(tdims, 0, dev)
s_sd, sdims = si.student_from_teacher(t_sd, tdims, 32, 2)
del t_sd
model = WhisperForConditionalGeneration(sdims, ops=ops, state_dict=s_sd, dtype=torch.bfloat16)
for B in [int(x) for x in os.environ.get("BS", "16,32").split(",")]:
    feats = torch.randn(B, 128, 3000, device=dev) * 0.5
    for _ in range(2): model.engine.encode(feats, save=False)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3): model.engine.encode(feats, save=False)
    e.record(); torch.cuda.synchronize()
    print(f"# B={B}: encoder {s.elapsed_time(e) / 3:.2f} ms per batch = {s.elapsed_time(e) / 3 / B:.3f} ms per window")
    ops.profile_detail = True
    ops.profile = {}
    model.engine.encode(feats, save=False)
    rows = sorted(ops.collect_profile().items(), key=lambda kv: -kv[1]["ms"])
    ops.profile = None
    for k, d in rows[:14]:
        tf = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["flops"] else 0.0
        print(f"| {k} | {d['n']} | {d['ms']:.2f} ms | {d['ms'] / d['n'] * 1e3:.1f} us | {tf:.0f} TF/s |")
