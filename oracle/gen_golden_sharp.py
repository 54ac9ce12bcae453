"""TEST INFRASTRUCTURE -- generates tests/golden/sharp_*.npz: the reference `train_step` (run_distillation.py:1465-1495)
on "trained-like" weights (oracle.whisper_oracle.sharpen_state_dict: attention-logit std ~4, logit std ~5, +-30 outlier
residual channels, x30 LayerNorm gains) instead of the near-uniform softmaxes of plain random init -- the regime in which
the product's default-on deviations live (deferred softmax maximum DW_ATTN_DEFER, gelu' kept in fp16, bf16 dlogits
scaled by 1/n_valid before rounding).  Computed by the `transformers` classes on CPU, once in fp32 and once the way the
reference trains (student fp32 master weights under bf16 autocast, teacher loaded in bf16: SURVEY.md 8a'), WITH the
backward: probe gradients and the gradient norm of BOTH runs are stored, so the GPU test can hold the HIP gradients
against the bf16-autocast run (apples to apples) as well as against fp32.

  sharp_tiny.npz        whisper-tiny.en 4/4 -> 4/1, B = 2, full mode (audio -> log-mel -> step)
  sharp_large_v3.npz    large-v3-shaped 32/32 -> 32/2, B = 3, full mode (features given)
  recipe_large_v3.npz   the same dimensions in the README's recipe mode: --freeze_encoder + shared encoder
                        (run_distillation.py:1018-1049, 1473-1478), sharp weights

Run in the build container:  python oracle/gen_golden_sharp.py [tiny] [large] [recipe]   (large + recipe: ~40 min, 8 cores)
"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import whisper_oracle as wo  # noqa: E402
from oracle.gen_golden import hf_model  # noqa: E402
from oracle.reference_loop import kl_divergence  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

PROBES = {
    "full": ["model.encoder.conv1.weight", "model.encoder.layers.0.self_attn.q_proj.weight",
             "model.encoder.layers.1.self_attn_layer_norm.weight", "model.encoder.layers.1.fc1.weight",
             "model.encoder.layers.{last}.fc2.weight", "model.encoder.layers.{last}.self_attn.k_proj.weight",
             "model.encoder.layer_norm.bias", "model.decoder.embed_positions.weight",
             "model.decoder.layers.0.encoder_attn.v_proj.weight", "model.decoder.layers.0.fc2.bias",
             "model.decoder.layers.{dlast}.self_attn.out_proj.weight", "model.decoder.layer_norm.weight"],
    "recipe": ["model.decoder.embed_positions.weight", "model.decoder.layers.0.self_attn.q_proj.weight",
               "model.decoder.layers.0.encoder_attn.k_proj.weight", "model.decoder.layers.0.encoder_attn.v_proj.bias",
               "model.decoder.layers.0.fc1.weight", "model.decoder.layers.{dlast}.fc2.weight",
               "model.decoder.layers.{dlast}.final_layer_norm.weight", "model.decoder.layer_norm.weight"],
}


def probe_slice(g):
    g = g.reshape(-1)
    return g[:: max(1, g.numel() // 512)][:512]


def inputs(cfg_t, B, seed, with_audio):
    b = wo.synthetic_batch(cfg_t, B, seed=seed + 1, with_audio=with_audio)
    if with_audio:
        from transformers import WhisperFeatureExtractor
        fe = WhisperFeatureExtractor(feature_size=cfg_t.n_mels)
        feats = fe([a for a in b["audio"]], sampling_rate=16000, return_tensors="pt").input_features
    else:
        feats = torch.randn(B, cfg_t.n_mels, 3000, generator=torch.Generator().manual_seed(seed + 2)) * 0.5
    return {"input_features": feats, "decoder_input_ids": b["decoder_input_ids"], "labels": b["labels"]}


def weights(cfg_t, seed, enc_s, dec_s):
    t_sd = wo.sharpen_state_dict(wo.init_state_dict(cfg_t, seed), cfg_t)
    s_sd, cfg_s = wo.student_from_teacher(t_sd, cfg_t, enc_s, dec_s)
    return t_sd, s_sd, cfg_s


def run(name, cfg_name, enc_s, dec_s, B, seed, recipe, with_audio):
    from transformers.modeling_outputs import BaseModelOutput
    cfg_t = wo.CONFIGS[cfg_name]
    batch = inputs(cfg_t, B, seed, with_audio)
    t_sd, s_sd, cfg_s = weights(cfg_t, seed, enc_s, dec_s)
    probes = [p.format(last=enc_s - 1, dlast=dec_s - 1) for p in PROBES["recipe" if recipe else "full"]]
    out = {"seed": seed, "B": B, "recipe": int(recipe), "probe_names": np.array(probes)}
    for tag, autocast in (("fp32", False), ("bf16", True)):
        t0 = time.time()
        teacher, student = hf_model(cfg_t, t_sd).eval(), hf_model(cfg_s, s_sd).train()
        for p in teacher.parameters():
            p.requires_grad_(False)
        teacher_dtype = torch.bfloat16 if autocast else torch.float32
        if autocast:
            teacher = teacher.to(torch.bfloat16)      # teacher_dtype = bf16 (run_distillation.py:800-806, 991)
        if recipe:
            student.freeze_encoder()                  # run_distillation.py:1018-1021
        ctx = torch.autocast("cpu", dtype=torch.bfloat16) if autocast else torch.autocast("cpu", enabled=False)
        with ctx:                                     # train_step, run_distillation.py:1465-1495
            so = student(**batch)
            with torch.no_grad():
                if recipe:
                    enc = BaseModelOutput(so.encoder_last_hidden_state.to(dtype=teacher_dtype))
                    to = teacher(encoder_outputs=enc, labels=batch["labels"])
                else:
                    to = teacher(**batch)
        # accelerate's autocast wraps model.forward only (convert_outputs_to_fp32): the KD lines run in fp32 outside it
        ce = so.loss.float()
        t_dist = nn.functional.softmax(to.logits.float() / 2.0, dim=-1)
        s_dist = nn.functional.log_softmax(so.logits.float() / 2.0, dim=-1)
        kl = kl_divergence(t_dist, s_dist, batch["labels"]) * 2.0 ** 2
        loss = 0.8 * ce + 1.0 * kl
        del t_dist, s_dist
        loss.backward()
        named = dict(student.named_parameters())
        gn = torch.sqrt(sum((p.grad.float() ** 2).sum() for p in named.values() if p.grad is not None))
        s_logits, t_logits = so.logits.detach().float(), to.logits.detach().float()
        out.update({f"ce_{tag}": ce.item(), f"kl_{tag}": kl.item(), f"loss_{tag}": loss.item(), f"grad_norm_{tag}": gn.item(),
                    f"s_logits_{tag}": s_logits[:, ::41, ::1777].numpy().copy(),
                    f"t_logits_{tag}": t_logits[:, ::41, ::1777].numpy().copy(),
                    f"enc_{tag}": so.encoder_last_hidden_state.detach().float()[:, ::211, ::97].numpy().copy()})
        # how sharp the fixture is (documentation of the regime; the test prints them)
        lab = batch["labels"]
        out[f"logit_std_{tag}"] = s_logits[lab >= 0].std().item()
        out[f"s_max_prob_{tag}"] = torch.softmax(s_logits[lab >= 0], -1).max(-1).values.mean().item()
        for i, n in enumerate(probes):
            out[f"grad{i}_{tag}"] = probe_slice(named[n].grad.detach().float()).numpy().copy()
        print(name, tag, {k: round(v, 6) for k, v in out.items() if isinstance(v, float) and k.endswith(tag)},
              f"{time.time() - t0:.0f} s", flush=True)
        del teacher, student, so, to, named, loss, ce, kl
    np.savez_compressed(os.path.join(GOLD, f"{name}.npz"), **out)


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count())
    what = sys.argv[1:] or ["tiny", "large", "recipe"]
    if "tiny" in what:
        run("sharp_tiny", "tiny.en", 4, 1, B=2, seed=51, recipe=False, with_audio=True)
    if "large" in what:
        run("sharp_large_v3", "large-v3", 32, 2, B=3, seed=53, recipe=False, with_audio=False)
    if "recipe" in what:
        run("recipe_large_v3", "large-v3", 32, 2, B=3, seed=55, recipe=True, with_audio=False)
