"""Token-step loop only (distil-large-v3 student, batch 16, graphs on): target of rocprofv3 --kernel-trace --stats.
Prints the wall time per step; the per-kernel durations come from the profile (their sum against the wall time tells how
much of a step is kernel time and how much is the boundaries between dependent kernels)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps
from distil_whisper_amd.modeling import WhisperFeatureExtractor, WhisperForConditionalGeneration
from distil_whisper_amd.longform import LongFormTranscriber
from distil_whisper_amd import student_init as si
dev = "cuda:0"
ops = HipOps(dev)
tdims = si.PRESETS["large-v3"]
t_sd = si.random_state_dict(tdims, 0, dev)
s_sd, sdims = si.student_from_teacher(t_sd, tdims, 32, 2)
del t_sd
model = WhisperForConditionalGeneration(sdims, ops=ops, state_dict=s_sd)
fe = WhisperFeatureExtractor(feature_size=128, ops=ops)
B, NEW = int(os.environ.get("B", 16)), int(os.environ.get("NEW", 64))
ops.lib.dw_debug_set(7, int(os.environ.get("FUSE_OFF", 4)))
ops.lib.dw_debug_set(8, int(os.environ.get("DW_KEY8", 5)))
tr = LongFormTranscriber(model, fe, batch_size=B, max_new_tokens=NEW, use_graphs=bool(int(os.environ.get("GRAPHS", 1))))
feats = torch.randn(B, 128, 3000, device=dev) * 0.5
enc, _ = model.engine.encode(feats, save=False)
prompt = tr.prompt[None, :].expand(B, -1).contiguous()
for _ in range(2): tr.decoder.run(enc, prompt, NEW)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4): tr.decoder.run(enc, prompt, NEW)
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 4 / NEW * 1e3:.4f} ms per decode step", flush=True)
