"""Baseline of the (round-2) decode path: distil-large-v3 student, greedy KV-cache decode of chunk batches on the
engine's existing kernels (no skinny-M kernel yet).  Mirrors run_eval.py's benchmark_gen set-up (806-844): random
encoder input, fixed number of new tokens."""
import json, os, sys, time, torch
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
model = WhisperForConditionalGeneration(sdims, ops=ops, state_dict=s_sd)
B, NEW = int(os.environ.get("B", 16)), int(os.environ.get("NEW", 64))
feats = torch.randn(B, 128, 3000, device=dev) * 0.5
model.generate(feats, max_new_tokens=4, return_dict_in_generate=True)
torch.cuda.synchronize()
t0 = time.perf_counter(); enc, _ = model.engine.encode(feats, save=False); torch.cuda.synchronize(); t_enc = time.perf_counter() - t0
t0 = time.perf_counter(); ids = model.generate(feats, max_new_tokens=NEW, return_dict_in_generate=True).sequences; torch.cuda.synchronize(); t_all = time.perf_counter() - t0
out = {"B": B, "new_tokens": NEW, "encode_ms": t_enc * 1e3, "generate_ms": t_all * 1e3,
       "ms_per_decode_step": (t_all - t_enc) / NEW * 1e3, "tokens_per_s": B * NEW / (t_all - t_enc),
       "rtfx_30s_chunks": B * 30.0 / t_all}
print(json.dumps(out))
