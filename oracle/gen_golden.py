"""Generates tests/golden/*.npz by running THE REFERENCE CLASSES (transformers' WhisperForConditionalGeneration and
WhisperFeatureExtractor, the third-party package the reference's hot path lives in) on seeded inputs, with the
reference `train_step` / optimizer set-up restated verbatim around them (run_distillation.py:1377-1407, 1453-1495,
1609-1614).  Run in the build container only (`python oracle/gen_golden.py`); the fixtures are committed so that the
GPU box needs neither /root/reference nor this script.

Weights: oracle.whisper_oracle.init_state_dict(seed) loaded into the HF modules with load_state_dict (strict), so the
oracle and the reference see bit-identical parameters.
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import whisper_oracle as wo  # noqa: E402

from transformers import WhisperConfig, WhisperFeatureExtractor, WhisperForConditionalGeneration  # noqa: E402
from transformers.modeling_outputs import BaseModelOutput  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def hf_model(cfg: wo.OracleConfig, sd):
    hc = WhisperConfig(vocab_size=cfg.vocab, num_mel_bins=cfg.n_mels, d_model=cfg.d_model,
                       encoder_layers=cfg.enc_layers, decoder_layers=cfg.dec_layers,
                       encoder_attention_heads=cfg.heads, decoder_attention_heads=cfg.heads,
                       encoder_ffn_dim=cfg.ffn, decoder_ffn_dim=cfg.ffn, max_source_positions=cfg.max_src,
                       max_target_positions=cfg.max_tgt, pad_token_id=cfg.pad_token_id, bos_token_id=cfg.pad_token_id,
                       eos_token_id=cfg.pad_token_id, decoder_start_token_id=cfg.decoder_start_token_id,
                       use_cache=False)
    m = WhisperForConditionalGeneration(hc)
    full = dict(sd)
    full["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
    missing, unexpected = m.load_state_dict(full, strict=False)
    assert not unexpected and all("proj_out" in k for k in missing), (missing, unexpected)
    assert m.proj_out.weight.data_ptr() == m.model.decoder.embed_tokens.weight.data_ptr()
    return m


def get_parameter_names(model, forbidden_layer_types, forbidden_module=None):
    """run_distillation.py:760-778, verbatim logic."""
    result = []
    for name, child in model.named_children():
        if forbidden_module is not None and isinstance(child, tuple(forbidden_module)):
            continue
        result += [f"{name}.{n}" for n in get_parameter_names(child, forbidden_layer_types, forbidden_module)
                   if not isinstance(child, tuple(forbidden_layer_types))]
    result += list(model._parameters.keys())
    return result


def reference_step(name, cfg_t, enc_s, dec_s, B, seed, temperature=2.0, kl_weight=1.0, weight_decay=0.0,
                   share_hidden_states=False, autocast=False):
    t_sd = wo.init_state_dict(cfg_t, seed)
    s_sd, cfg_s = wo.student_from_teacher(t_sd, cfg_t, enc_s, dec_s)
    teacher, student = hf_model(cfg_t, t_sd), hf_model(cfg_s, s_sd)
    batch_np = wo.synthetic_batch(cfg_t, B, seed=seed + 1)
    fe = WhisperFeatureExtractor(feature_size=cfg_t.n_mels)
    feats = fe([a for a in batch_np["audio"]], sampling_rate=16000, return_tensors="pt").input_features
    batch = {"input_features": feats, "decoder_input_ids": batch_np["decoder_input_ids"],
             "labels": batch_np["labels"]}
    if autocast:
        teacher = teacher.to(torch.bfloat16)  # teacher_dtype = bf16 (run_distillation.py:800-806, 991)

    # optimizer exactly as run_distillation.py:1386-1407
    decay_parameters = get_parameter_names(student, [nn.LayerNorm])
    decay_parameters = [n for n in decay_parameters if "bias" not in n]
    groups = [
        {"params": [p for n, p in student.named_parameters() if n in decay_parameters and p.requires_grad],
         "weight_decay": weight_decay},
        {"params": [p for n, p in student.named_parameters() if n not in decay_parameters and p.requires_grad],
         "weight_decay": 0.0},
    ]
    opt = torch.optim.AdamW(groups, lr=1e-4, betas=(0.9, 0.999), eps=1e-8)

    def kl_divergence(target_distribution, log_predicted_distribution, labels):
        kl_loss = nn.KLDivLoss(reduction="none")
        divergence = kl_loss(log_predicted_distribution, target_distribution)
        padding_mask = labels >= 0
        padding_mask = padding_mask.unsqueeze(-1)
        divergence = divergence * padding_mask
        divergence = divergence.sum() / padding_mask.sum()
        return divergence

    student.train()
    teacher.eval()
    ctx = torch.autocast("cpu", dtype=torch.bfloat16) if autocast else torch.autocast("cpu", enabled=False)
    with ctx:
        student_outputs = student(**batch)
        with torch.no_grad():
            if share_hidden_states:
                enc = BaseModelOutput(student_outputs.encoder_last_hidden_state.to(teacher.dtype))
                teacher_outputs = teacher(encoder_outputs=enc, labels=batch["labels"])
            else:
                teacher_outputs = teacher(**batch)
    s_logits = student_outputs.logits.float()
    t_logits = teacher_outputs.logits.float()
    ce_loss = student_outputs.loss.float()
    teacher_distribution = nn.functional.softmax(t_logits / temperature, dim=-1)
    student_distribution = nn.functional.log_softmax(s_logits / temperature, dim=-1)
    kl_loss = kl_divergence(teacher_distribution, student_distribution, batch["labels"]) * temperature ** 2
    loss = 0.8 * ce_loss + kl_weight * kl_loss
    out = {"ce": ce_loss.item(), "kl": kl_loss.item(), "loss": loss.item(), "seed": seed, "B": B,
           "mel_slice": feats[:, ::9, ::97].numpy().copy(),
           "enc_slice": student_outputs.encoder_last_hidden_state.float()[:, ::211, ::37].detach().numpy().copy(),
           "s_logits_slice": s_logits[:, ::61, ::977].detach().numpy().copy(),
           "t_logits_slice": t_logits[:, ::61, ::977].detach().numpy().copy()}
    if not autocast:
        loss.backward()
        gnorm = torch.nn.utils.clip_grad_norm_(student.parameters(), 1.0)
        out["grad_norm"] = gnorm.item()
        named = dict(student.named_parameters())
        probe = ["model.encoder.conv1.weight", "model.encoder.conv2.bias",
                 "model.encoder.layers.0.self_attn.q_proj.weight", "model.encoder.layers.1.fc1.bias",
                 "model.encoder.layer_norm.weight", "model.decoder.embed_tokens.weight",
                 "model.decoder.embed_positions.weight", "model.decoder.layers.0.encoder_attn.k_proj.weight",
                 "model.decoder.layers.0.encoder_attn.v_proj.bias", "model.decoder.layers.0.fc2.weight",
                 "model.decoder.layers.0.self_attn_layer_norm.bias"]
        coef = min(1.0, 1.0 / (gnorm.item() + 1e-6))
        for i, n in enumerate(probe):
            g = named[n].grad.reshape(-1) / coef  # store the UNCLIPPED gradient
            out[f"grad{i}"] = g[:: max(1, g.numel() // 256)][:256].numpy().copy()
        out["probe_names"] = np.array(probe)
        opt.step()
        for i, n in enumerate(probe):
            p = named[n].detach().reshape(-1)
            out[f"param{i}"] = p[:: max(1, p.numel() // 256)][:256].numpy().copy()
    np.savez_compressed(os.path.join(GOLD, f"{name}.npz"), **out)
    print(name, {k: v for k, v in out.items() if isinstance(v, float)})


def logmel_fixture():
    rng = np.random.default_rng(7)
    audio = (0.1 * rng.standard_normal((3, 480000))).astype(np.float32)
    audio[1, 161234:] = 0.0          # zero-padded clip (collator pads short audio with zeros)
    audio[2] *= np.linspace(0.0, 1.0, 480000, dtype=np.float32) ** 2
    out = {"seed": 7}
    for M in (80, 128):
        fe = WhisperFeatureExtractor(feature_size=M)
        f = fe([a for a in audio], sampling_rate=16000, return_tensors="np").input_features
        out[f"mel{M}"] = f[:, :, ::25].astype(np.float32)
        out[f"filters{M}_sum"] = np.asarray(fe.mel_filters, dtype=np.float64).sum(0)
    np.savez_compressed(os.path.join(GOLD, "logmel.npz"), **out)
    print("logmel fixture written")


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    logmel_fixture()
    reference_step("micro_fp32", wo.CONFIGS["micro"], 2, 1, B=2, seed=11)
    reference_step("micro_shared_wd", wo.CONFIGS["micro"], 2, 1, B=2, seed=12, weight_decay=0.1,
                   share_hidden_states=True)
    reference_step("tiny_fp32", wo.CONFIGS["tiny.en"], 4, 1, B=2, seed=13)
    reference_step("tiny_bf16_autocast", wo.CONFIGS["tiny.en"], 4, 1, B=2, seed=13, autocast=True)
