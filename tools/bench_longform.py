"""BASELINE config 5 shape: distil-large-v3 student, 5-minute synthetic clips cut into 30 s windows with 5 s strides
(15 windows per clip), batches of 16 windows, greedy decode of a fixed number of new tokens (mirrors run_eval.py's
benchmark_gen, 806-844: random weights never emit EOS, so every window decodes max_new_tokens).  Reports audio-seconds
transcribed per second and decoded tokens/s, with the token steps launched eagerly vs replayed from HIP graphs."""
import json, os, sys, time, torch
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
CLIPS, NEW, B = int(os.environ.get("CLIPS", 4)), int(os.environ.get("NEW", 128)), int(os.environ.get("B", 16))
audio = [0.1 * torch.randn(4_800_000, device=dev) for _ in range(CLIPS)]
res = {"clips": CLIPS, "clip_s": 300, "batch": B, "new_tokens": NEW}
if os.environ.get("DW_FUSE_AB"):              # decode-pass fusions OFF bits (key 7): 1 LN on load, 2 K/V append, 4 self-attention with its QKV projection inside, 8 cross-attention with its q projection inside
    for mode in eval(os.environ.get("DW_FUSE_MODES", "(3, 2, 1, 0, 3, 0)")):
        ops.lib.dw_debug_set(7, mode)
        tr = LongFormTranscriber(model, fe, batch_size=B, max_new_tokens=NEW, use_graphs=True)
        feats = torch.randn(B, 128, 3000, device=dev) * 0.5
        enc, _ = model.engine.encode(feats, save=False)
        prompt = tr.prompt[None, :].expand(B, -1).contiguous()
        for _ in range(2): tr.decoder.run(enc, prompt, NEW)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): tr.decoder.run(enc, prompt, NEW)
        torch.cuda.synchronize()
        print(f"fuse_off={mode}: {(time.perf_counter() - t0) / 3 / NEW * 1e3:.4f} ms per decode step (graphs)", file=sys.stderr)
    ops.lib.dw_debug_set(7, 4)
if os.environ.get("DW_DECODE_AB"):            # streaming single-query attention kernel off (tile kernel) vs on
    for mode in (0, 1):
        ops.lib.dw_debug_set(4, mode)
        tr = LongFormTranscriber(model, fe, batch_size=B, max_new_tokens=NEW, use_graphs=True)
        feats = torch.randn(B, 128, 3000, device=dev) * 0.5
        enc, _ = model.engine.encode(feats, save=False)
        prompt = tr.prompt[None, :].expand(B, -1).contiguous()
        tr.decoder.run(enc, prompt, NEW); torch.cuda.synchronize()
        t0 = time.perf_counter(); tr.decoder.run(enc, prompt, NEW); torch.cuda.synchronize(); td = time.perf_counter() - t0
        res[f"decode_attn_kernel_{mode}"] = {"ms_per_decode_step": td / NEW * 1e3}
    ops.lib.dw_debug_set(4, 1)
for graphs in (False, True):
    tr = LongFormTranscriber(model, fe, batch_size=B, max_new_tokens=NEW, use_graphs=graphs)
    windows = len(tr.plan([a.numel() for a in audio]))
    tr(audio[:1])                                  # warm-up (captures the graphs)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); out = tr(audio); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    # decode-only time of one batch
    feats = torch.randn(B, 128, 3000, device=dev) * 0.5
    enc, _ = model.engine.encode(feats, save=False)
    prompt = tr.prompt[None, :].expand(B, -1).contiguous()
    tr.decoder.run(enc, prompt, NEW); torch.cuda.synchronize()
    t0 = time.perf_counter(); tr.decoder.run(enc, prompt, NEW); torch.cuda.synchronize(); td = time.perf_counter() - t0
    res["graphs" if graphs else "eager"] = {"windows": windows, "wall_s": dt, "audio_s_per_s": CLIPS * 300 / dt,
                                            "ms_per_decode_step": td / NEW * 1e3, "decode_tokens_per_s": B * NEW / td}
# the encoder of batch i+1 on one stream beside the token loop of batch i on another (LongFormTranscriber(overlap=True)), with
# `decode_cus` CUs kept out of the persistent GEMM grids; the transcripts must equal the sequential run's
CLIPS_OV = max(CLIPS, 8)
audio_ov = audio + [0.1 * torch.randn(4_800_000, device=dev) for _ in range(CLIPS_OV - CLIPS)]
seq = LongFormTranscriber(model, fe, batch_size=B, max_new_tokens=NEW, use_graphs=True)
seq(audio_ov[:1]); torch.cuda.synchronize()
t0 = time.perf_counter(); ref_out = seq(audio_ov); torch.cuda.synchronize(); dt_seq = time.perf_counter() - t0
res["overlap"] = {"clips": CLIPS_OV, "sequential": {"wall_s": dt_seq, "audio_s_per_s": CLIPS_OV * 300 / dt_seq}}
for dc in eval(os.environ.get("DECODE_CUS", "(0, 32, 64, 96)")):
    tr = LongFormTranscriber(model, fe, batch_size=B, max_new_tokens=NEW, use_graphs=True, overlap=True, decode_cus=dc)
    tr(audio_ov[:1]); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = tr(audio_ov); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res["overlap"][f"decode_cus_{dc}"] = {"wall_s": dt, "audio_s_per_s": CLIPS_OV * 300 / dt, "same_transcripts": out == ref_out}
del seq
# per-launch-class breakdown of the decode step (eager, HIP events around every launch; 32 steps)
tr = LongFormTranscriber(model, fe, batch_size=B, max_new_tokens=33, use_graphs=False)
feats = torch.randn(B, 128, 3000, device=dev) * 0.5
enc, _ = model.engine.encode(feats, save=False)
prompt = tr.prompt[None, :].expand(B, -1).contiguous()
tr.decoder.run(enc, prompt, 33); torch.cuda.synchronize()
ops.profile, ops.profile_detail = {}, True
tr.decoder.run(enc, prompt, 33); torch.cuda.synchronize()
prof = ops.collect_profile(); ops.profile = None
res["decode_step_breakdown_us"] = {k: {"n_per_step": round(v["n"] / 32, 2), "us_per_step": round(v["ms"] / 32 * 1e3, 2)}
                                   for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
print(json.dumps(res))
