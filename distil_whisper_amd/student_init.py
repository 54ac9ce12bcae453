"""Synthetic weights and student initialisation for the MI355X path.

`student_from_teacher` follows create_student_model.py:92-216 of the reference: every non-layer weight is copied and
the student keeps maximally spaced teacher layers (np.linspace(0, L_teacher-1, L_student, dtype=int), last one forced
to the teacher's last layer, lines 129-144; per-layer copy 169-182).  `random_state_dict` builds seeded random weights
with the HF parameter names directly on the device (there are no checkpoints on disk in this environment).
"""
import math

import numpy as np
import torch

from .engine import WhisperDims, layer_names


def sinusoids(length, channels, max_timescale=10000.0):
    """Encoder positional table (TF:modeling_whisper.py:55-64)."""
    inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2))
    t = torch.arange(length).view(-1, 1) * inv.view(1, -1)
    return torch.cat([t.sin(), t.cos()], dim=1)


def random_state_dict(dims: WhisperDims, seed: int, device="cpu", std=0.02):
    g = torch.Generator(device=device).manual_seed(seed)
    D, Fd = dims.d_model, dims.ffn

    def rn(*shape, s=std):
        return torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * s

    sd = {"model.encoder.conv1.weight": rn(D, dims.n_mels, 3, s=0.05), "model.encoder.conv1.bias": rn(D),
          "model.encoder.conv2.weight": rn(D, D, 3, s=0.03), "model.encoder.conv2.bias": rn(D),
          "model.encoder.embed_positions.weight": sinusoids(dims.max_src, D).to(device)}

    def layer(prefix, cross):
        for n, kind in layer_names(prefix, cross):
            if kind == "zero":
                continue
            if kind == "ln":
                sd[n] = (1.0 + rn(D, s=0.05)) if n.endswith("weight") else rn(D, s=0.05)
            elif n.endswith("fc1.weight"):
                sd[n] = rn(Fd, D)
            elif n.endswith("fc1.bias"):
                sd[n] = rn(Fd)
            elif n.endswith("fc2.weight"):
                sd[n] = rn(D, Fd)
            elif kind == "w":
                sd[n] = rn(D, D)
            else:
                sd[n] = rn(D)

    for i in range(dims.enc_layers):
        layer(f"model.encoder.layers.{i}", False)
    sd["model.encoder.layer_norm.weight"] = 1.0 + rn(D, s=0.05)
    sd["model.encoder.layer_norm.bias"] = rn(D, s=0.05)
    sd["model.decoder.embed_tokens.weight"] = rn(dims.vocab, D)
    sd["model.decoder.embed_positions.weight"] = rn(dims.max_tgt, D)
    for i in range(dims.dec_layers):
        layer(f"model.decoder.layers.{i}", True)
    sd["model.decoder.layer_norm.weight"] = 1.0 + rn(D, s=0.05)
    sd["model.decoder.layer_norm.bias"] = rn(D, s=0.05)
    return sd


def student_layer_map(n_teacher: int, n_student: int):
    m = np.linspace(0, n_teacher - 1, n_student, dtype=int)
    m[-1] = n_teacher - 1
    return [int(x) for x in m]


def student_from_teacher(teacher_sd, tdims: WhisperDims, enc_layers=None, dec_layers: int = 2,
                         decoder_layers_numbers=None):
    """`init_student_model_from_teacher` (create_student_model.py:92-182).  The reference first loads the teacher's
    state dict non-strictly into the smaller student (so student layer i starts as teacher layer i), then overwrites
    the mapped layers: decoder layers always (129-144, 169-175), through `{teacher_layer: student_layer}` dictionaries
    built from `np.linspace` or from the `decoder_layers_numbers` override (136-139; a teacher layer listed twice keeps
    its LAST student slot); encoder layers only when `encoder_layers` was given (177-182)."""
    if decoder_layers_numbers is not None and len(decoder_layers_numbers) != dec_layers:
        raise ValueError(f"Got {len(decoder_layers_numbers)} layers number for {dec_layers} decoder layers.")
    n_enc = enc_layers if enc_layers is not None else tdims.enc_layers
    sdims = WhisperDims(**{**tdims.__dict__, "enc_layers": n_enc, "dec_layers": dec_layers})
    sd = {k: v for k, v in teacher_sd.items() if ".layers." not in k and k != "proj_out.weight"}
    enc_map = {t: s for s, t in enumerate(student_layer_map(tdims.enc_layers, n_enc))}
    dec_src = student_layer_map(tdims.dec_layers, dec_layers) if decoder_layers_numbers is None \
        else [int(x) for x in decoder_layers_numbers]
    dec_map = {t: s for s, t in enumerate(dec_src)}

    def copy_layer(part, cross, ti, si):
        for n, kind in layer_names(f"model.{part}.layers.{ti}", cross):
            if kind != "zero":
                sd[n.replace(f".layers.{ti}.", f".layers.{si}.")] = teacher_sd[n]

    for part, nt, ns, cross, mp, remap in (("encoder", tdims.enc_layers, n_enc, False, enc_map, enc_layers is not None),
                                           ("decoder", tdims.dec_layers, dec_layers, True, dec_map, True)):
        for i in range(min(ns, nt)):                          # non-strict load_state_dict: same-index layers
            copy_layer(part, cross, i, i)
        if remap:
            for ti in range(nt):
                if ti in mp:
                    copy_layer(part, cross, ti, mp[ti])
        missing = [i for i in range(ns) if f"model.{part}.layers.{i}.fc1.weight" not in sd]
        if missing:
            raise RuntimeError(f"student {part} layers {missing} have no teacher layer to start from")
    return sd, sdims


# dimensions of the BASELINE.json configurations (TF:configuration_whisper.py:127-164 + public checkpoint shapes)
PRESETS = {
    "tiny.en": WhisperDims(384, 6, 1536, 4, 4, 51864, 80),
    "small.en": WhisperDims(768, 12, 3072, 12, 12, 51864, 80),
    "large-v3": WhisperDims(1280, 20, 5120, 32, 32, 51866, 128, decoder_start_token_id=50258),
}
STUDENT_LAYERS = {"tiny.en": (4, 1), "small.en": (12, 4), "large-v3": (32, 2)}


def mel_filter_bank(n_mels, n_freq=201, sr=16000, fmin=0.0, fmax=8000.0):
    """Slaney-scale, slaney-normalised triangular mel filters [n_freq, n_mels] (float64 numpy), the table
    WhisperFeatureExtractor builds with mel_filter_bank(201, M, 0, 8000, 16000, "slaney", "slaney")
    (TF:feature_extraction_whisper.py:95-103, TF:audio_utils.py:638-729)."""
    def hz2mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-10) / 1000.0) * (27.0 / np.log(6.4)), 3.0 * f / 200.0)

    def mel2hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= 15.0, 1000.0 * np.exp(np.log(6.4) / 27.0 * (m - 15.0)), 200.0 * m / 3.0)

    pts = mel2hz(np.linspace(hz2mel(fmin), hz2mel(fmax), n_mels + 2))
    freqs = np.linspace(0, sr // 2, n_freq)
    fdiff = np.diff(pts)
    slopes = pts[None, :] - freqs[:, None]
    fb = np.maximum(0.0, np.minimum(-slopes[:, :-2] / fdiff[:-1], slopes[:, 2:] / fdiff[1:]))
    return fb * (2.0 / (pts[2: n_mels + 2] - pts[:n_mels]))[None, :]
