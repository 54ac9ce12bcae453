"""Pseudo-labelling throughput (round-5 review item 8): the TEACHER's generate -- whisper-large-v3
shape, 32 decoder layers, bf16 weights -- over 30 s packs, as run_pseudo_labelling.py:861-996 drives it per GPU.

A different regime from the long-form bench (2-layer student): a token step streams 32 layers of decoder weights
(32 x 52.5 MB + the 133 MB LM head = 1.8 GB) + 32 x 15.4 MB x B of cross-attention K/V, so it is bound by HBM bytes per
step, not by launch gaps; and the encoder is the same 32 layers either way.

Reports audio-seconds labelled per second end to end (gather packs -> log-mel -> encoder -> cross K/V -> greedy decode of
NEW tokens, random weights never emit EOS: mirrors benchmark_gen of run_eval.py:806-844), ms per decode step eager and
from HIP graphs, and the achieved fraction of the HBM rate on the step's algorithmic bytes."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps                                                   # noqa: E402
from distil_whisper_amd.modeling import WhisperFeatureExtractor, WhisperForConditionalGeneration  # noqa: E402
from distil_whisper_amd.pseudo_label import PseudoLabeller                                       # noqa: E402
from distil_whisper_amd import student_init as si                                                # noqa: E402

dev = "cuda:0"
ops = HipOps(dev)
B, NEW, PACKS = int(os.environ.get("B", 16)), int(os.environ.get("NEW", 128)), int(os.environ.get("PACKS", 64))
tdims = si.PRESETS["large-v3"]
t_sd = si.random_state_dict(tdims, 0, dev)
model = WhisperForConditionalGeneration(tdims, ops=ops, state_dict=t_sd, dtype=torch.bfloat16)
del t_sd
fe = WhisperFeatureExtractor(feature_size=128, ops=ops)
g = torch.Generator(device=dev).manual_seed(0)
# utterances of 5-14 s of one speaker each: the packing rule joins consecutive ones up to 30 s
audios, spk = [], []
for i in range(PACKS * 3):
    n = int(torch.randint(80000, 224000, (1,), generator=g, device=dev).item())
    audios.append(0.1 * torch.randn(n, generator=g, device=dev))
    spk.append(i // 6)
D, F, L, V = tdims.d_model, tdims.ffn, tdims.dec_layers, tdims.vocab
w_layer = (4 * D * D + 4 * D * D + 2 * D * F) * 2          # self q/k/v/o + cross q/o (+ k/v unused at the step) + fc1/fc2, bf16
step_bytes = L * w_layer + V * D * 2 + B * L * 2 * 1500 * D * 2
res = {"model": "whisper-large-v3 shape, 32/32, bf16", "batch": B, "new_tokens": NEW,
       "algorithmic_bytes_per_decode_step": step_bytes}
for graphs in (False, True):
    pl = PseudoLabeller(model, fe, batch_size=B, max_new_tokens=NEW, use_graphs=graphs)
    pl(audios[:B * 2], spk[:B * 2])                                   # warm-up (captures the graphs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rows, packs, cond = pl(audios, spk)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    secs = sum(a.numel() for a in audios) / 16000.0
    # decode-only time of one batch
    feats = torch.randn(B, 128, 3000, device=dev) * 0.5
    enc, _ = model.engine.encode(feats, save=False)
    prompt = pl.prompt[None, :].expand(B, -1).contiguous()
    pl.decoder.run(enc, prompt, NEW)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pl.decoder.run(enc, prompt, NEW)
    torch.cuda.synchronize()
    td = time.perf_counter() - t0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        model.engine.encode(feats, save=False)
    e1.record()
    torch.cuda.synchronize()
    res["graphs" if graphs else "eager"] = {
        "utterances": len(audios), "packs": len(packs), "audio_s": secs, "wall_s": dt, "audio_s_per_s": secs / dt,
        "ms_per_decode_step": td / NEW * 1e3, "decode_tokens_per_s": B * NEW / td,
        "decode_step_TBps_on_algorithmic_bytes": step_bytes / (td / NEW) / 1e12,
        "decode_step_frac_of_8TBps": step_bytes / (td / NEW) / 8e12,
        "encoder_ms_per_batch": e0.elapsed_time(e1) / 3}
# the encoder of the next batch of packs beside the token loop of the current one (PseudoLabeller(overlap=True)); same labels
ref = PseudoLabeller(model, fe, batch_size=B, max_new_tokens=NEW, use_graphs=True)
ref(audios[:B * 2], spk[:B * 2]); torch.cuda.synchronize()
t0 = time.perf_counter(); rows_ref, _, _ = ref(audios, spk); torch.cuda.synchronize(); dt_ref = time.perf_counter() - t0
secs = sum(a.numel() for a in audios) / 16000.0
res["overlap"] = {"sequential": {"wall_s": dt_ref, "audio_s_per_s": secs / dt_ref}}
for dc in (0, 32, 64):
    pl = PseudoLabeller(model, fe, batch_size=B, max_new_tokens=NEW, use_graphs=True, overlap=True, decode_cus=dc)
    pl(audios[:B * 2], spk[:B * 2]); torch.cuda.synchronize()
    t0 = time.perf_counter(); rows, _, _ = pl(audios, spk); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res["overlap"][f"decode_cus_{dc}"] = {"wall_s": dt, "audio_s_per_s": secs / dt, "same_labels": rows == rows_ref}
print(json.dumps(res))
