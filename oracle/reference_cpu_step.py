"""TEST / MEASUREMENT INFRASTRUCTURE -- the reference path itself, timed on the host cores (bench.py `cpu_baseline`).

What runs: `transformers.WhisperForConditionalGeneration` (teacher + student) and
`transformers.WhisperFeatureExtractor` -- the third-party classes the reference's hot path lives in -- driven by the
reference's `train_step` / `kl_divergence` (run_distillation.py:1453-1495), its two AdamW parameter groups
(1377-1407, `get_parameter_names` 760-778) and `clip_grad_norm_(1.0)` (1611), restated around them exactly as
oracle/gen_golden.py does for the parity fixtures.  fp32, `torch.set_num_threads(usable cores)`, synthetic 30 s clips
(log-mel INCLUDED in the timed step, as in the GPU number), random-init weights of the named configuration,
>= 1 warm-up step + the timed steps.  Only bench.py's `cpu_baseline` leg and tests import this module.
"""
import time

import numpy as np
import torch
import torch.nn as nn


def _hf_config(c):
    from transformers import WhisperConfig
    return WhisperConfig(vocab_size=c.vocab, num_mel_bins=c.n_mels, d_model=c.d_model, encoder_layers=c.enc_layers,
                         decoder_layers=c.dec_layers, encoder_attention_heads=c.heads, decoder_attention_heads=c.heads,
                         encoder_ffn_dim=c.ffn, decoder_ffn_dim=c.ffn, max_source_positions=c.max_src,
                         max_target_positions=c.max_tgt, pad_token_id=c.pad_token_id, bos_token_id=c.pad_token_id,
                         eos_token_id=c.pad_token_id, decoder_start_token_id=c.decoder_start_token_id)


def get_parameter_names(model, forbidden_layer_types, forbidden_module=None):
    """run_distillation.py:760-778."""
    result = []
    for name, child in model.named_children():
        if forbidden_module is not None and isinstance(child, tuple(forbidden_module)):
            continue
        result += [f"{name}.{n}" for n in get_parameter_names(child, forbidden_layer_types, forbidden_module)
                   if not isinstance(child, tuple(forbidden_layer_types))]
    result += list(model._parameters.keys())
    return result


def kl_divergence(target_distribution, log_predicted_distribution, labels):
    """run_distillation.py:1453-1462."""
    kl_loss = nn.KLDivLoss(reduction="none")
    divergence = kl_loss(log_predicted_distribution, target_distribution)
    padding_mask = labels >= 0
    padding_mask = padding_mask.unsqueeze(-1)
    divergence = divergence * padding_mask
    divergence = divergence.sum() / padding_mask.sum()
    return divergence


def timed_reference_steps(cfg_t, enc_s, dec_s, batch=1, recipe=False, warmup=1, steps=3, budget_s=240.0, seed=1234,
                          threads=None):
    """Returns dict(seconds_per_step, steps, warmup, batch, loss, threads).  `recipe`: --freeze_encoder with the
    shared encoder output (run_distillation.py:1018-1049, 1473-1478).  The number of timed steps shrinks to what fits
    `budget_s` (never below 1)."""
    from transformers import WhisperFeatureExtractor, WhisperForConditionalGeneration
    from transformers.modeling_outputs import BaseModelOutput
    from . import whisper_oracle as wo
    if threads:
        torch.set_num_threads(int(threads))
    torch.manual_seed(0)
    cfg_s = wo.OracleConfig(**{**cfg_t.__dict__, "enc_layers": enc_s, "dec_layers": dec_s})
    teacher = WhisperForConditionalGeneration(_hf_config(cfg_t)).eval()
    student = WhisperForConditionalGeneration(_hf_config(cfg_s))
    # student layers <- maximally spaced teacher layers (create_student_model.py:129-182)
    t_sd = teacher.state_dict()
    s_sd = {k: v for k, v in t_sd.items() if ".layers." not in k}
    for part, nt, ns in (("encoder", cfg_t.enc_layers, enc_s), ("decoder", cfg_t.dec_layers, dec_s)):
        for si, ti in enumerate(wo.student_layer_map(nt, ns)):
            for k, v in t_sd.items():
                if k.startswith(f"model.{part}.layers.{ti}."):
                    s_sd[k.replace(f".layers.{ti}.", f".layers.{si}.")] = v
    student.load_state_dict(s_sd, strict=False)
    if recipe:
        student.freeze_encoder()
    for p in teacher.parameters():
        p.requires_grad_(False)
    decay = [n for n in get_parameter_names(student, [nn.LayerNorm]) if "bias" not in n]
    groups = [{"params": [p for n, p in student.named_parameters() if n in decay and p.requires_grad],
               "weight_decay": 0.0},
              {"params": [p for n, p in student.named_parameters() if n not in decay and p.requires_grad],
               "weight_decay": 0.0}]
    opt = torch.optim.AdamW(groups, lr=1e-4, betas=(0.9, 0.999), eps=1e-8)
    fe = WhisperFeatureExtractor(feature_size=cfg_t.n_mels)
    b = wo.synthetic_batch(cfg_t, batch, seed=seed)
    temperature, kl_weight = 2.0, 1.0

    def step():
        feats = fe([a for a in b["audio"]], sampling_rate=16000, return_tensors="pt").input_features
        data = {"input_features": feats, "decoder_input_ids": b["decoder_input_ids"], "labels": b["labels"]}
        student.train()
        student_outputs = student(**data)
        with torch.no_grad():
            if recipe:
                enc = BaseModelOutput(student_outputs.encoder_last_hidden_state)
                teacher_outputs = teacher(encoder_outputs=enc, labels=data["labels"])
            else:
                teacher_outputs = teacher(**data)
        ce_loss = student_outputs.loss
        teacher_distribution = nn.functional.softmax(teacher_outputs.logits / temperature, dim=-1)
        student_distribution = nn.functional.log_softmax(student_outputs.logits / temperature, dim=-1)
        kl_loss = kl_divergence(teacher_distribution, student_distribution, data["labels"]) * temperature ** 2
        loss = 0.8 * ce_loss + kl_weight * kl_loss
        loss.backward()
        torch.nn.utils.clip_grad_norm_(student.parameters(), 1.0)
        opt.step()
        opt.zero_grad()
        return float(loss.item())

    t_begin = time.perf_counter()
    w_times = []
    for _ in range(max(1, warmup)):
        t0 = time.perf_counter()
        loss = step()
        w_times.append(time.perf_counter() - t0)
    left = budget_s - (time.perf_counter() - t_begin)
    n = max(1, min(int(steps), int(left // max(w_times[-1], 1e-3))))
    t0 = time.perf_counter()
    for _ in range(n):
        loss = step()
    dt = (time.perf_counter() - t0) / n
    return {"seconds_per_step": dt, "steps": n, "warmup": max(1, warmup), "batch": batch, "loss": loss,
            "threads": torch.get_num_threads(), "warmup_seconds": w_times}
