"""KV-cache greedy decoding at the dimensions of the BASELINE config 5 model (distil-large-v3 decoder: d_model 1280,
20 heads, FFN 5120, vocabulary 51866, 1500 encoder positions, 2 layers) and of the pseudo-labelling teacher (32 decoder
layers) against `transformers.generate` fixtures (tests/golden/decode_large_v3.json from
oracle/gen_golden_decode_large.py; run_eval.py:806-844 `benchmark_gen` shape: random encoder outputs,
min_new_tokens = max_new_tokens):

  * free-running: the generated ids must be IDENTICAL (eager and HIP-graph replay, fp32-master and bf16 models); the
    fixture keeps the input whose smallest top-1/top-2 margin is widest (stored, in logit standard deviations);
  * teacher-forced: fed the reference's own tokens through the KV cache, every step's logits at the reference's top-8
    ids must equal the reference's values to bf16 noise, and the argmax must agree wherever the reference's margin is
    clear of that noise.  Tolerances are in units of the reference's logit standard deviation and written below.
"""
import json
import os

import pytest
import torch

from oracle import gen_golden_decode_large as gl

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "decode_large_v3.json")
GOLD = json.load(open(PATH))
SC = {s["name"]: s for s in GOLD["scenarios"]}
# largest |logit - reference| / sigma allowed at the reference's top-8 ids: fp32 restatement on CPU; bf16 HIP kernels
# (measured on the MI355X, maximum over 384 values: 0.037 sigma with the fp32 residual stream, 0.040 with the bf16 one,
# for the 2-layer decoder -- a third of the free run's smallest margin; the bounds leave 1.5x)
TOL = {"cpu": 2e-4, "student_2_layer_decoder": 0.06, "teacher_32_layer_decoder": 0.15}


def _build(ops, s, dtype):
    from distil_whisper_amd.generation import GenerationConfig
    from distil_whisper_amd.modeling import WhisperForConditionalGeneration
    cfg = gl.CFGS[s["name"]]
    m = WhisperForConditionalGeneration(cfg, ops=ops, state_dict=gl.weights(cfg, s["weight_seed"]), dtype=dtype)
    m.generation_config = GenerationConfig.from_any(gl.generation_fields(s["seed"]))
    return cfg, m, gl.encoder_states(cfg, s["seed"], s["B"]).to(ops.device)


def _free_running(m, enc, s, **kw):
    from distil_whisper_amd.modeling import BaseModelOutput
    n = s["n_free"]
    out = m.generate(encoder_outputs=BaseModelOutput(last_hidden_state=enc), min_new_tokens=n, max_new_tokens=n,
                     return_dict_in_generate=True, **kw)
    assert out.sequences.tolist() == s["sequences_free"], (s["name"], kw, out.sequences.tolist(), s["sequences_free"])


def _teacher_forced(m, cfg, enc, s, tol):
    """Worst |logit - reference| / sigma over the reference's top-8 ids of every step; asserts the argmax over the
    decodable ids wherever the reference's own margin exceeds 3 x tol."""
    eng = m.engine
    B, P, n = s["B"], s["prompt_len"], s["n_forced"]
    seq = torch.tensor(s["sequences_forced"], device=enc.device)
    keep = torch.tensor(gl.kept_ids(s["seed"]), device=enc.device)
    e = enc.reshape(-1, cfg.d_model).to(eng.lowp).contiguous()
    cache = eng.decode_init(e, B, P + n)
    worst = 0.0
    for t in range(P + n - 1):
        logits = eng.decode_step(seq[:, t:t + 1].contiguous(), cache)[:B, :cfg.vocab].float()
        i = t - (P - 1)
        if i < 0:
            continue
        ids = torch.tensor([row[i] for row in s["top8_ids"]], device=enc.device)
        ref = torch.tensor([row[i] for row in s["top8_values"]], device=enc.device)
        worst = max(worst, ((logits.gather(1, ids) - ref).abs().max() / s["sigma"]).item())
        first = i == 0
        sc = logits[:, keep].clone()
        if first:                                              # begin_suppress_tokens: kept_ids[0] (EOS is not kept)
            sc[:, 0] = float("-inf")
        pick = keep[sc.argmax(-1)]
        for b in range(B):
            if s["margins_forced"][b][i] > 3 * tol:
                assert int(pick[b]) == s["sequences_forced"][b][P + i], (s["name"], b, i)
    assert worst <= tol, (s["name"], worst, tol)
    return worst


def test_student_decoder_dims_match_transformers_cpu():
    from oracle.ref_ops import RefOps
    s = SC["student_2_layer_decoder"]
    assert s["margin"] >= GOLD["meta"]["min_margin"]
    cfg, m, enc = _build(RefOps("cpu", lowp=torch.float32), s, torch.float32)
    _free_running(m, enc, s, use_graphs=False)
    _teacher_forced(m, cfg, enc, s, TOL["cpu"])


@pytest.mark.gpu
@pytest.mark.parametrize("name,dtype", [("student_2_layer_decoder", torch.float32), ("student_2_layer_decoder", torch.bfloat16),
                                        ("teacher_32_layer_decoder", torch.bfloat16)])
def test_large_v3_dims_decode_matches_transformers_gpu(name, dtype):
    from distil_whisper_amd.ops_hip import HipOps
    s = SC[name]
    assert s["margin"] >= GOLD["meta"]["min_margin"]
    cfg, m, enc = _build(HipOps("cuda:0"), s, dtype)
    worst = _teacher_forced(m, cfg, enc, s, TOL[name])
    print(f"{name} {dtype}: worst top-8 logit deviation {worst:.4f} sigma (reference margin of the free run {s['margin']:.3f})")
    _free_running(m, enc, s, use_graphs=False)
    _free_running(m, enc, s, use_graphs=True)
