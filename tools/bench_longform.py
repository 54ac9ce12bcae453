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
print(json.dumps(res))
