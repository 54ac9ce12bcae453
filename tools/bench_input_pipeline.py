"""Input side of the training step, timed against what the step consumes (round-5 review item 8).

The step eats 32 clips per ~340 ms on one MI355X = ~94 clips/s/GPU.  What has to keep up with that, per batch of 32 x 30 s:

  host   prepare_train_labels  (run_distillation.py:1186-1229: timestamp filtering + prompt assembly on token-id lists)
  host   collator              (405-478: pad to 448, shift, -100 masks, prompt mask) -> three small device tensors
  H2D    32 x 480 000 fp32 samples (61.4 MB) from a pinned staging buffer
  GPU    log-mel               (dw_logmel: the audio half of prepare_train_dataset, 1176-1177)

and, for comparison, what the reference does instead of the last two: `transformers.WhisperFeatureExtractor` on the host
cores (its numpy path, and its torch path on CPU) followed by the H2D of the features (49 MB).

Prints one JSON line; clips/s per stage and for the whole chain, plus the chain run from a prefetch thread WHILE the GPU
executes training-step-sized work (a GEMM loop of the step's duration) -- does the input side hide behind the step?
"""
import json
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distil_whisper_amd.ops_hip import HipOps                                   # noqa: E402
from distil_whisper_amd import student_init as si                                # noqa: E402
from distil_whisper_amd.collator import DataCollatorSpeechSeq2SeqWithPadding     # noqa: E402
from distil_whisper_amd.labels import prepare_train_labels                       # noqa: E402

B, REPS = int(os.environ.get("B", 32)), int(os.environ.get("REPS", 20))
STEP_MS = float(os.environ.get("STEP_MS", 340.0))
dev = "cuda:0"
ops = HipOps(dev)
filt = torch.tensor(si.mel_filter_bank(128), dtype=torch.float32, device=dev).contiguous()
rng = np.random.default_rng(0)
SOT, PREV, NOTS = 50258, 50362, 50364           # large-v3 vocabulary: <|startoftranscript|>, <|startofprev|>, <|notimestamps|>


def make_token_lists():
    """Pseudo-label id lists as the tokenizer hands them over: SOT, language, task, then text with timestamp pairs."""
    out = []
    for _ in range(B):
        n = int(rng.integers(32, 225))
        body = rng.integers(0, 50257, n).tolist()
        for j in range(4, n - 1, 12):           # a timestamp pair every dozen tokens
            body[j] = NOTS + 1 + (j // 12) * 50
        out.append([SOT, 50259, 50360] + body)
    return out


audio_host = (0.1 * rng.standard_normal((B, 480000))).astype(np.float32)
pinned = torch.from_numpy(audio_host).pin_memory()
audio_dev = torch.empty((B, 480000), dtype=torch.float32, device=dev)
col = DataCollatorSpeechSeq2SeqWithPadding(max_target_length=448, device=dev, decoder_start_token_id=SOT,
                                           decoder_prev_token_id=PREV, report_valid_len="per_sequence")
lab_rng = np.random.RandomState(1)


def t_host(fn, reps=REPS):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, r


def stage_labels():
    toks = make_token_lists()
    return prepare_train_labels(toks, None, timestamp_begin=NOTS, timestamp_position=3, decoder_prev_token_id=PREV,
                                timestamp_probability=0.2, condition_on_prev_probability=0.2, max_label_length=448, rng=lab_rng)


def stage_collate(labels):
    return col([{"labels": l} for l in labels])


def stage_h2d():
    audio_dev.copy_(pinned, non_blocking=True)


def stage_mel():
    return ops.logmel(audio_dev, filt)


def chain():
    labels = stage_labels()
    batch = stage_collate(labels)
    stage_h2d()
    batch["input_features"] = stage_mel()
    return batch


res = {"batch": B, "step_ms_assumed": STEP_MS, "clips_per_s_the_step_consumes": B / STEP_MS * 1e3}
dt, labels = t_host(stage_labels)
res["prepare_train_labels_ms"] = dt * 1e3
dt, _ = t_host(lambda: stage_collate(labels))
res["collator_ms"] = dt * 1e3
dt, _ = t_host(stage_h2d)
res["h2d_audio_ms"] = dt * 1e3
res["h2d_audio_GBps"] = audio_host.nbytes / dt / 1e9
dt, feats = t_host(stage_mel)
res["logmel_gpu_ms"] = dt * 1e3
dt, _ = t_host(chain)
res["chain_ms_per_batch"] = dt * 1e3
res["chain_clips_per_s"] = B / dt
res["chain_over_step_consumption"] = (B / dt) / res["clips_per_s_the_step_consumes"]

# the chain from a prefetch thread while the GPU is busy with step-sized work on the main stream
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
torch.matmul(a, a)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    torch.matmul(a, a)
e1.record()
torch.cuda.synchronize()
per = e0.elapsed_time(e1) / 20
n_mm = max(1, int(STEP_MS / per))
produced, stop = [0], [False]
side = torch.cuda.Stream()


def producer():
    torch.cuda.set_device(0)
    with torch.cuda.stream(side):
        while not stop[0]:
            chain()
            side.synchronize()
            produced[0] += 1


th = threading.Thread(target=producer, daemon=True)
t0 = time.perf_counter()
th.start()
steps = 10
for _ in range(steps):
    for _ in range(n_mm):
        torch.matmul(a, a)
    torch.cuda.current_stream().synchronize()
wall = time.perf_counter() - t0
stop[0] = True
th.join(timeout=10)
res["under_load"] = {"gpu_steps": steps, "wall_s": wall, "ms_per_step_with_prefetch_thread": wall / steps * 1e3,
                     "ms_per_step_of_the_load_alone": n_mm * per, "batches_prefetched": produced[0],
                     "prefetched_clips_per_s": produced[0] * B / wall}

# the reference's front end on the host cores (what the GPU log-mel + audio H2D replace)
try:
    from transformers import WhisperFeatureExtractor
    fe = WhisperFeatureExtractor(feature_size=128)
    clips = [a_ for a_ in audio_host[:8]]
    torch.set_num_threads(len(os.sched_getaffinity(0)))
    t0 = time.perf_counter()
    f = fe(clips, sampling_rate=16000, return_tensors="pt").input_features
    d_np = time.perf_counter() - t0
    res["reference_feature_extractor_cpu"] = {"cores": len(os.sched_getaffinity(0)), "clips": 8,
                                              "numpy_path_clips_per_s": 8 / d_np}
    try:
        t0 = time.perf_counter()
        f = fe(clips, sampling_rate=16000, return_tensors="pt", device="cpu").input_features
        res["reference_feature_extractor_cpu"]["torch_cpu_path_clips_per_s"] = 8 / (time.perf_counter() - t0)
    except Exception as e:      # noqa: BLE001
        res["reference_feature_extractor_cpu"]["torch_cpu_path_error"] = str(e)[:120]
    err = (feats[:8].cpu() - f).abs().max().item()
    res["logmel_gpu_vs_reference_fe_max_abs"] = err
except ImportError as e:
    res["reference_feature_extractor_cpu"] = {"error": str(e)}
print(json.dumps(res))
