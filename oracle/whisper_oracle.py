"""TEST INFRASTRUCTURE (CPU oracle) -- never imported by the product path.

Plain-torch (fp32, CPU) restatement of the reference distillation hot path of huggingface/distil-whisper.  The
arithmetic of that path lives in the un-vendored third-party package `transformers` (reference pin
`transformers>=4.35.1`, training/setup.py:22; installed 5.15.0): every function below cites the reference lines it
restates (`TF:` = transformers/models/whisper/, other paths relative to /root/reference/training).

Parity pin: the reference ships NO tests or golden vectors for this path, so the oracle is pinned
against outputs of the reference classes themselves, generated in the build container by oracle/gen_golden.py
(imports `transformers`' WhisperForConditionalGeneration / WhisperFeatureExtractor, loads the same seeded weights,
runs the reference `train_step` verbatim) and committed under tests/golden/.  tests/test_oracle.py checks this file
against those fixtures.
"""
import math
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class OracleConfig:
    d_model: int
    heads: int
    ffn: int
    enc_layers: int
    dec_layers: int
    vocab: int
    n_mels: int
    max_src: int = 1500
    max_tgt: int = 448
    pad_token_id: int = 50256
    decoder_start_token_id: int = 50257


# dimensions of the BASELINE.json configs (TF:configuration_whisper.py:127-164)
CONFIGS = {
    "tiny.en": OracleConfig(384, 6, 1536, 4, 4, 51864, 80),
    "small.en": OracleConfig(768, 12, 3072, 12, 12, 51864, 80),
    "large-v3": OracleConfig(1280, 20, 5120, 32, 32, 51866, 128, pad_token_id=50256, decoder_start_token_id=50258),
    # plumbing-size config for fast CPU tests (same code paths, 2 encoder / 2 decoder layers)
    "micro": OracleConfig(128, 2, 256, 2, 2, 1000, 80, pad_token_id=0, decoder_start_token_id=1),
}


def sinusoids(length, channels, max_timescale=10000.0):
    """TF:modeling_whisper.py:55-64."""
    inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2))
    t = torch.arange(length).view(-1, 1) * inv.view(1, -1)
    return torch.cat([t.sin(), t.cos()], dim=1)


def layer_param_names(prefix, cross):
    names = []
    for blk in (["self_attn", "encoder_attn"] if cross else ["self_attn"]):
        names += [f"{prefix}.{blk}.k_proj.weight", f"{prefix}.{blk}.v_proj.weight", f"{prefix}.{blk}.v_proj.bias",
                  f"{prefix}.{blk}.q_proj.weight", f"{prefix}.{blk}.q_proj.bias", f"{prefix}.{blk}.out_proj.weight",
                  f"{prefix}.{blk}.out_proj.bias", f"{prefix}.{blk}_layer_norm.weight", f"{prefix}.{blk}_layer_norm.bias"]
    names += [f"{prefix}.fc1.weight", f"{prefix}.fc1.bias", f"{prefix}.fc2.weight", f"{prefix}.fc2.bias",
              f"{prefix}.final_layer_norm.weight", f"{prefix}.final_layer_norm.bias"]
    return names


def init_state_dict(cfg: OracleConfig, seed: int, std: float = 0.02, bias_std: float = 0.02):
    """Seeded random weights with the HF parameter names/shapes (proj_out.weight is tied to embed_tokens and not
    stored).  Biases and LayerNorm parameters are randomised too (HF zero-inits them) so that parity tests exercise
    them."""
    g = torch.Generator().manual_seed(seed)
    D, Fd = cfg.d_model, cfg.ffn

    def rn(*shape, s=std):
        return torch.randn(*shape, generator=g) * s

    sd = {
        "model.encoder.conv1.weight": rn(D, cfg.n_mels, 3, s=0.05), "model.encoder.conv1.bias": rn(D, s=bias_std),
        "model.encoder.conv2.weight": rn(D, D, 3, s=0.03), "model.encoder.conv2.bias": rn(D, s=bias_std),
        "model.encoder.embed_positions.weight": sinusoids(cfg.max_src, D),
    }

    def layer(prefix, cross):
        for n in layer_param_names(prefix, cross):
            if n.endswith("layer_norm.weight"):
                sd[n] = 1.0 + rn(D, s=0.1)
            elif n.endswith("layer_norm.bias"):
                sd[n] = rn(D, s=0.1)
            elif n.endswith("fc1.weight"):
                sd[n] = rn(Fd, D)
            elif n.endswith("fc1.bias"):
                sd[n] = rn(Fd, s=bias_std)
            elif n.endswith("fc2.weight"):
                sd[n] = rn(D, Fd)
            elif n.endswith(".weight"):
                sd[n] = rn(D, D)
            else:
                sd[n] = rn(D, s=bias_std)

    for i in range(cfg.enc_layers):
        layer(f"model.encoder.layers.{i}", False)
    sd["model.encoder.layer_norm.weight"] = 1.0 + rn(D, s=0.1)
    sd["model.encoder.layer_norm.bias"] = rn(D, s=0.1)
    sd["model.decoder.embed_tokens.weight"] = rn(cfg.vocab, D)
    sd["model.decoder.embed_positions.weight"] = rn(cfg.max_tgt, D)
    for i in range(cfg.dec_layers):
        layer(f"model.decoder.layers.{i}", True)
    sd["model.decoder.layer_norm.weight"] = 1.0 + rn(D, s=0.1)
    sd["model.decoder.layer_norm.bias"] = rn(D, s=0.1)
    return sd


def sharpen_state_dict(sd, cfg: OracleConfig, attn_logit_std: float = 4.0, logit_std: float = 5.0,
                       outlier: float = 30.0):
    """"Trained-like" statistics on top of init_state_dict (in place, deterministic; no pretrained weights exist
    offline).  Random-init weights at std 0.02 give near-uniform softmaxes everywhere (CE ~ ln V); a trained Whisper
    has peaked attention rows, peaked logits and a few residual / LayerNorm channels two orders of magnitude above the
    rest.  This puts the arithmetic into that regime:
      * q_proj / k_proj weights rescaled so that q.k/sqrt(64) over LayerNorm outputs has std ~ attn_logit_std
        (q std = sqrt(D) sigma, 64-term dot product / 8 => D sigma^2);
      * the decoder's final LayerNorm gain and bias multiplied so that the logits have std ~ logit_std
        (sqrt(D) x 0.02 x gain; scaling the tied embedding instead would put |E[id]|^2 / std(x) ~ 100 on the INPUT
        token's own logit and saturate every softmax at exactly 1);
      * outlier channels: three residual channels receive +-outlier/2 through fc2.bias in the first two layers of each
        stack (a large-mean residual stream, as Whisper's massive activations), and two LayerNorm gains per fourth
        layer are multiplied by `outlier`.
    Used by oracle/gen_golden_sharp.py and the GPU parity tests that consume its fixtures."""
    D = cfg.d_model
    s_qk = math.sqrt(attn_logit_std / D) / 0.02
    s_e = (logit_std / math.sqrt(D)) / 0.02
    chans = [7 % D, (D // 4 + 3) % D, D - 1]
    gains = [(D // 2 + 5) % D, (3 * D // 4 + 1) % D]
    for k in list(sd.keys()):
        if k.endswith("q_proj.weight") or k.endswith("k_proj.weight"):
            sd[k] = sd[k] * s_qk
    sd["model.decoder.layer_norm.weight"] = sd["model.decoder.layer_norm.weight"] * s_e
    sd["model.decoder.layer_norm.bias"] = sd["model.decoder.layer_norm.bias"] * s_e
    for part, n in (("encoder", cfg.enc_layers), ("decoder", cfg.dec_layers)):
        for i in range(n):
            if i < 2:
                b = sd[f"model.{part}.layers.{i}.fc2.bias"].clone()
                for j, c in enumerate(chans):
                    b[c] += (outlier / 2) * (1.0 if j % 2 == 0 else -1.0)     # two layers add up to +-outlier
                sd[f"model.{part}.layers.{i}.fc2.bias"] = b
            if i % 4 == 1:
                for ln in ("self_attn_layer_norm", "final_layer_norm"):
                    w = sd[f"model.{part}.layers.{i}.{ln}.weight"].clone()
                    w[gains] *= outlier
                    sd[f"model.{part}.layers.{i}.{ln}.weight"] = w
    return sd


def student_layer_map(n_teacher: int, n_student: int):
    """create_student_model.py:129-144: maximally spaced teacher layers, last one forced to the teacher's last."""
    m = np.linspace(0, n_teacher - 1, n_student, dtype=int)
    m[-1] = n_teacher - 1
    return [int(x) for x in m]


def student_from_teacher(teacher_sd, cfg_t: OracleConfig, enc_layers: int, dec_layers: int):
    """create_student_model.py:92-216: copy every non-layer weight, keep the mapped encoder/decoder layers."""
    cfg_s = OracleConfig(**{**cfg_t.__dict__, "enc_layers": enc_layers, "dec_layers": dec_layers})
    sd = {k: v.clone() for k, v in teacher_sd.items() if ".layers." not in k}
    for part, nt, ns, cross in (("encoder", cfg_t.enc_layers, enc_layers, False),
                                ("decoder", cfg_t.dec_layers, dec_layers, True)):
        for si, ti in enumerate(student_layer_map(nt, ns)):
            for n in layer_param_names(f"model.{part}.layers.{ti}", cross):
                sd[n.replace(f".layers.{ti}.", f".layers.{si}.")] = teacher_sd[n].clone()
    return sd, cfg_s


def shift_tokens_right(labels, pad_token_id, decoder_start_token_id):
    """TF:modeling_whisper.py:68-81."""
    out = labels.new_zeros(labels.shape)
    out[:, 1:] = labels[:, :-1].clone()
    out[:, 0] = decoder_start_token_id
    out.masked_fill_(out == -100, pad_token_id)
    return out


def _attention(sd, p, x, kv, H, causal):
    """WhisperAttention.forward (TF:modeling_whisper.py:284-356) + eager_attention_forward (215-238): q is scaled by
    head_dim**-0.5 BEFORE the product, k_proj has no bias, pure causal mask for decoder self-attention."""
    B, L, D = x.shape
    hd = D // H
    q = F.linear(x, sd[f"{p}.q_proj.weight"], sd[f"{p}.q_proj.bias"]) * hd ** -0.5
    src = x if kv is None else kv
    k = F.linear(src, sd[f"{p}.k_proj.weight"])
    v = F.linear(src, sd[f"{p}.v_proj.weight"], sd[f"{p}.v_proj.bias"])
    q = q.view(B, L, H, hd).transpose(1, 2)
    k = k.view(B, -1, H, hd).transpose(1, 2)
    v = v.view(B, -1, H, hd).transpose(1, 2)
    w = torch.matmul(q, k.transpose(2, 3))
    if causal:
        Lk = k.shape[2]
        w = w + torch.full((L, Lk), float("-inf"), dtype=w.dtype).triu(1)
    w = F.softmax(w, dim=-1)
    o = torch.matmul(w, v).transpose(1, 2).reshape(B, L, D)
    return F.linear(o, sd[f"{p}.out_proj.weight"], sd[f"{p}.out_proj.bias"])


def _ln(sd, name, x):
    return F.layer_norm(x, (x.shape[-1],), sd[f"{name}.weight"], sd[f"{name}.bias"], 1e-5)


def encoder_forward(sd, cfg: OracleConfig, input_features):
    """WhisperEncoder.forward (TF:modeling_whisper.py:592-646) and WhisperEncoderLayer.forward (379-413)."""
    x = F.gelu(F.conv1d(input_features, sd["model.encoder.conv1.weight"], sd["model.encoder.conv1.bias"], padding=1))
    x = F.gelu(F.conv1d(x, sd["model.encoder.conv2.weight"], sd["model.encoder.conv2.bias"], stride=2, padding=1))
    x = x.permute(0, 2, 1) + sd["model.encoder.embed_positions.weight"]
    for i in range(cfg.enc_layers):
        p = f"model.encoder.layers.{i}"
        x = x + _attention(sd, f"{p}.self_attn", _ln(sd, f"{p}.self_attn_layer_norm", x), None, cfg.heads, False)
        h = _ln(sd, f"{p}.final_layer_norm", x)
        h = F.gelu(F.linear(h, sd[f"{p}.fc1.weight"], sd[f"{p}.fc1.bias"]))
        x = x + F.linear(h, sd[f"{p}.fc2.weight"], sd[f"{p}.fc2.bias"])
    return _ln(sd, "model.encoder.layer_norm", x)


def decoder_forward(sd, cfg: OracleConfig, decoder_input_ids, enc_out):
    """WhisperDecoder.forward (TF:modeling_whisper.py:690-795), WhisperDecoderLayer.forward (448-505), tied LM head
    (965, 1080)."""
    T = decoder_input_ids.shape[1]
    x = sd["model.decoder.embed_tokens.weight"][decoder_input_ids] + sd["model.decoder.embed_positions.weight"][:T]
    for i in range(cfg.dec_layers):
        p = f"model.decoder.layers.{i}"
        x = x + _attention(sd, f"{p}.self_attn", _ln(sd, f"{p}.self_attn_layer_norm", x), None, cfg.heads, True)
        x = x + _attention(sd, f"{p}.encoder_attn", _ln(sd, f"{p}.encoder_attn_layer_norm", x), enc_out, cfg.heads,
                           False)
        h = _ln(sd, f"{p}.final_layer_norm", x)
        h = F.gelu(F.linear(h, sd[f"{p}.fc1.weight"], sd[f"{p}.fc1.bias"]))
        x = x + F.linear(h, sd[f"{p}.fc2.weight"], sd[f"{p}.fc2.bias"])
    x = _ln(sd, "model.decoder.layer_norm", x)
    return F.linear(x, sd["model.decoder.embed_tokens.weight"])


def model_forward(sd, cfg, input_features=None, decoder_input_ids=None, labels=None, encoder_outputs=None):
    """WhisperForConditionalGeneration.forward (TF:modeling_whisper.py:994-1099): returns (loss, logits, enc_out)."""
    if decoder_input_ids is None:
        decoder_input_ids = shift_tokens_right(labels, cfg.pad_token_id, cfg.decoder_start_token_id)
    enc = encoder_forward(sd, cfg, input_features) if encoder_outputs is None else encoder_outputs
    logits = decoder_forward(sd, cfg, decoder_input_ids, enc)
    loss = None
    if labels is not None:
        loss = F.cross_entropy(logits.reshape(-1, cfg.vocab), labels.reshape(-1), ignore_index=-100)
    return loss, logits, enc


def kl_divergence(target_distribution, log_predicted_distribution, labels):
    """run_distillation.py:1453-1462."""
    divergence = F.kl_div(log_predicted_distribution, target_distribution, reduction="none")
    padding_mask = (labels >= 0).unsqueeze(-1)
    divergence = divergence * padding_mask
    return divergence.sum() / padding_mask.sum()


def train_step(student_sd, cfg_s, teacher_sd, cfg_t, batch, temperature=2.0, kl_weight=1.0, share_hidden_states=False):
    """run_distillation.py:1465-1495.  Returns (loss, metrics, student_logits, teacher_logits, enc_out)."""
    s_loss, s_logits, enc = model_forward(student_sd, cfg_s, **batch)
    with torch.no_grad():
        if share_hidden_states:
            _, t_logits, _ = model_forward(teacher_sd, cfg_t, labels=batch["labels"], encoder_outputs=enc.detach())
        else:
            _, t_logits, _ = model_forward(teacher_sd, cfg_t, **batch)
    teacher_distribution = F.softmax(t_logits / temperature, dim=-1)
    student_distribution = F.log_softmax(s_logits / temperature, dim=-1)
    kl_loss = kl_divergence(teacher_distribution, student_distribution, batch["labels"]) * temperature ** 2
    loss = 0.8 * s_loss + kl_weight * kl_loss
    return loss, {"loss": loss, "ce_loss": s_loss, "kl_loss": kl_loss}, s_logits, t_logits, enc


def decay_parameter_names(sd):
    """run_distillation.py:760-778, 1386-1391: weight decay for everything except LayerNorm parameters and biases."""
    return [n for n in sd if "layer_norm" not in n and not n.endswith(".bias")]


def clip_and_adamw(params, grads, state, step, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                   max_grad_norm=1.0, decay_names=None):
    """accelerator.clip_grad_norm_ (run_distillation.py:1611) + torch.optim.AdamW.step with the two param groups of
    run_distillation.py:1392-1407.  In place on `params`; returns the pre-clip global norm."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(max_grad_norm / (total + 1e-6), max=1.0)
    b1, b2 = betas
    for n, p in params.items():
        if n not in grads:
            continue
        g = grads[n] * coef
        st = state.setdefault(n, {"m": torch.zeros_like(p), "v": torch.zeros_like(p)})
        wd = weight_decay if (decay_names is None or n in decay_names) else 0.0
        p.mul_(1.0 - lr * wd)
        st["m"].mul_(b1).add_(g, alpha=1.0 - b1)
        st["v"].mul_(b2).addcmul_(g, g, value=1.0 - b2)
        denom = st["v"].sqrt() / math.sqrt(1.0 - b2 ** step) + eps
        p.addcdiv_(st["m"], denom, value=-lr / (1.0 - b1 ** step))
    return total


def collate(label_lists, decoder_start_token_id, max_target_length=448, pad_token_id=50256):
    """DataCollatorSpeechSeq2SeqWithPadding.__call__ label side (run_distillation.py:438-478), restated with plain
    loops: tokenizer.pad to max_target_length (pad id + attention mask), decoder_input_ids = labels[:, :-1],
    labels = labels[:, 1:], padding -> -100, then every position before (and including) a <|startoftranscript|>
    found at index > 0 of the shifted labels (i.e. a prompt) -> -100."""
    B = len(label_lists)
    ids = torch.full((B, max_target_length), pad_token_id, dtype=torch.long)
    att = torch.zeros((B, max_target_length), dtype=torch.long)
    for i, l in enumerate(label_lists):
        ids[i, : len(l)] = torch.tensor(l, dtype=torch.long)
        att[i, : len(l)] = 1
    dec_in = ids[:, :-1].clone()
    labels = ids[:, 1:].clone()
    labels[att[:, 1:] != 1] = -100
    for i in range(B):
        bos = 0
        for j in range(labels.shape[1]):
            if labels[i, j] == decoder_start_token_id:
                bos = j
                break
        if bos > 0:
            labels[i, : bos + 1] = -100
    return dec_in, labels


# ---- log-mel front end ------------------------------------------------------------------------------------------
def hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    mels = 3.0 * f / 200.0
    log_region = f >= 1000.0
    return np.where(log_region, 15.0 + np.log(np.maximum(f, 1e-10) / 1000.0) * (27.0 / np.log(6.4)), mels)


def mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f = 200.0 * m / 3.0
    log_region = m >= 15.0
    return np.where(log_region, 1000.0 * np.exp(np.log(6.4) / 27.0 * (m - 15.0)), f)


def mel_filter_bank(n_mels, n_freq=201, sr=16000, fmin=0.0, fmax=8000.0):
    """TF:audio_utils.py:638-729 mel_filter_bank(201, M, 0, 8000, 16000, norm="slaney", mel_scale="slaney")
    (called from TF:feature_extraction_whisper.py:95-103).  Returns [n_freq, n_mels] float64."""
    mel_pts = np.linspace(hz_to_mel_slaney(fmin), hz_to_mel_slaney(fmax), n_mels + 2)
    filter_freqs = mel_to_hz_slaney(mel_pts)
    fft_freqs = np.linspace(0, sr // 2, n_freq)
    fdiff = np.diff(filter_freqs)
    slopes = filter_freqs[None, :] - fft_freqs[:, None]
    down = -slopes[:, :-2] / fdiff[:-1]
    up = slopes[:, 2:] / fdiff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    enorm = 2.0 / (filter_freqs[2: n_mels + 2] - filter_freqs[:n_mels])
    return fb * enorm[None, :]


def logmel(audio: np.ndarray, n_mels: int) -> np.ndarray:
    """WhisperFeatureExtractor._torch_extract_fbank_features (TF:feature_extraction_whisper.py:135-168), in float64
    numpy with an explicit DFT: reflect-pad centre STFT (n_fft 400, hop 160, periodic Hann), |X|^2, drop last frame,
    mel, log10(clamp 1e-10), max(x, max-8), (x+4)/4.  audio [B, N] -> [B, n_mels, N/160] float32."""
    B, N = audio.shape
    x = np.pad(audio.astype(np.float64), ((0, 0), (200, 200)), mode="reflect")
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(400) / 400.0)
    T = N // 160
    idx = np.arange(T)[:, None] * 160 + np.arange(400)[None, :]
    frames = x[:, idx] * win                                   # [B, T, 400]
    spec = np.fft.rfft(frames, n=400, axis=-1)                 # [B, T, 201]
    power = spec.real ** 2 + spec.imag ** 2
    mel = power @ mel_filter_bank(n_mels)                      # [B, T, M]
    log_spec = np.log10(np.maximum(mel, 1e-10)).transpose(0, 2, 1)
    mx = log_spec.max(axis=(1, 2), keepdims=True)
    log_spec = np.maximum(log_spec, mx - 8.0)
    return np.ascontiguousarray(((log_spec + 4.0) / 4.0).astype(np.float32))


# ---- synthetic workload of BASELINE.md section 3 / SURVEY.md section 8(d) -----------------------------------------------
def synthetic_batch(cfg: OracleConfig, B: int, seed: int = 1234, T: int = 447, with_audio: bool = True):
    rng = np.random.default_rng(seed)
    out = {}
    if with_audio:
        out["audio"] = (0.1 * rng.standard_normal((B, 480000))).astype(np.float32)
    ids = rng.integers(0, min(cfg.vocab, 50257), size=(B, T + 1))
    ids[:, 0] = cfg.decoder_start_token_id
    lens = rng.integers(32, 225, size=B) if T >= 224 else rng.integers(max(T // 4, 1), T + 1, size=B)
    dec_in = torch.tensor(ids[:, :-1], dtype=torch.long)
    labels = torch.tensor(ids[:, 1:], dtype=torch.long)
    mask = torch.arange(T)[None, :] >= torch.tensor(lens)[:, None]
    labels = labels.masked_fill(mask, -100)
    out["decoder_input_ids"] = dec_in
    out["labels"] = labels
    return out
