"""Drop-in boundary: the attention plug point, checkpoint round trips, post-construction freezing
and the feature extractor's `pad`, each against the `transformers` behaviour it stands in for."""
import os

import numpy as np
import pytest
import torch

from oracle import gen_golden_decode as gd
from oracle import whisper_oracle as wo
from oracle.ref_ops import RefOps


def _hf_micro(attn, seed=3, cfg=None):
    from transformers import WhisperForConditionalGeneration
    cfg = cfg or wo.CONFIGS["micro"]
    hc = gd.hf_config(cfg)
    hc._attn_implementation = attn
    m = WhisperForConditionalGeneration(hc)
    sd = wo.init_state_dict(cfg, seed)
    full = dict(sd)
    full["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
    m.load_state_dict(full, strict=False)
    return m, sd, cfg


def _batch(cfg, B=2, T=40, seed=5):
    b = wo.synthetic_batch(cfg, B, seed=seed, T=T, with_audio=False)
    feats = torch.randn(B, cfg.n_mels, 3000, generator=torch.Generator().manual_seed(seed)) * 0.5
    return feats, b["decoder_input_ids"], b["labels"]


def test_attention_interface_runs_unmodified_hf_whisper_cpu():
    """`AttentionInterface.register` (TF:modeling_utils.py:5093-5131): an unmodified transformers Whisper whose
    config selects "hip_attention" goes through distil_whisper_amd.attention_interface for all three attention shapes,
    forward and backward.  CPU leg: the torch restatement of the kernels (fp32) stands in for the GPU."""
    pytest.importorskip("transformers")
    import distil_whisper_amd.attention_interface as ai
    ai._OPS["cpu"] = RefOps("cpu", lowp=torch.float32)
    name = ai.register()
    ref, _, cfg = _hf_micro("sdpa")
    new, _, _ = _hf_micro(name)
    assert new.config._attn_implementation == name
    feats, ids, labels = _batch(cfg)
    a = ref(input_features=feats, decoder_input_ids=ids, labels=labels)
    b = new(input_features=feats, decoder_input_ids=ids, labels=labels)
    assert torch.allclose(a.logits, b.logits, atol=2e-4, rtol=1e-4)
    a.loss.backward()
    b.loss.backward()
    for (n, p), (_, q) in zip(ref.named_parameters(), new.named_parameters()):
        if p.grad is not None:
            assert torch.allclose(p.grad, q.grad, atol=1e-5, rtol=2e-3), n
    g1 = ref.generate(feats, max_new_tokens=6)
    g2 = new.generate(feats, max_new_tokens=6)          # 1-query cached decode steps through the same callable
    assert torch.equal(g1, g2)
    with pytest.raises(NotImplementedError, match="mask"):
        ai.hip_attention_forward(new.model.decoder.layers[0].self_attn, torch.zeros(1, 2, 4, 64),
                                       torch.zeros(1, 2, 4, 64), torch.zeros(1, 2, 4, 64),
                                       attention_mask=torch.zeros(1, 1, 4, 4))
    del ai._OPS["cpu"]


@pytest.mark.gpu
def test_attention_interface_runs_unmodified_hf_whisper_gpu():
    """Same on the MI355X: transformers' own Whisper under bf16 autocast, sdpa vs the HIP kernel (tolerance = bf16
    rounding of the attention output, 2^-8 relative), and identical greedy tokens on margin-selected weights."""
    pytest.importorskip("transformers")
    import distil_whisper_amd.attention_interface as ai
    name = ai.register()
    ref, _, cfg = _hf_micro("sdpa")
    new, _, _ = _hf_micro(name)
    ref, new = ref.cuda(), new.cuda()
    feats, ids, labels = (t.cuda() for t in _batch(cfg))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        a = ref(input_features=feats, decoder_input_ids=ids, labels=labels)
        b = new(input_features=feats, decoder_input_ids=ids, labels=labels)
    assert abs(a.loss.item() - b.loss.item()) < 2e-3 * abs(a.loss.item())
    rel = ((a.logits.float() - b.logits.float()).norm() / a.logits.float().norm()).item()
    assert rel < 2e-2, rel
    a.loss.backward()
    b.loss.backward()
    for (n, p), (_, q) in zip(ref.named_parameters(), new.named_parameters()):
        if p.grad is not None and p.grad.norm() > 0:
            r = ((p.grad - q.grad).norm() / p.grad.norm()).item()
            assert r < 0.06, (n, r)


def test_checkpoint_round_trip_and_generation_config(tmp_path):
    """`save_pretrained` / `from_pretrained` (run_distillation.py:986-1004, 1765, 1791) through a checkpoint directory
    that `transformers` itself can read back."""
    from distil_whisper_amd.modeling import WhisperConfig, WhisperFeatureExtractor, WhisperForConditionalGeneration
    ops = RefOps("cpu", lowp=torch.float32)
    cfg = wo.CONFIGS["micro"]
    sd = wo.init_state_dict(cfg, 9)
    hc = WhisperConfig(vocab_size=cfg.vocab, num_mel_bins=cfg.n_mels, encoder_layers=cfg.enc_layers,
                       encoder_attention_heads=cfg.heads, decoder_layers=cfg.dec_layers,
                       decoder_attention_heads=cfg.heads, decoder_ffn_dim=cfg.ffn, encoder_ffn_dim=cfg.ffn,
                       d_model=cfg.d_model, pad_token_id=cfg.pad_token_id, bos_token_id=cfg.pad_token_id,
                       eos_token_id=cfg.pad_token_id, decoder_start_token_id=cfg.decoder_start_token_id)
    m = WhisperForConditionalGeneration(hc, ops=ops, state_dict=sd)
    m.generation_config.max_length = 77
    m.generation_config.suppress_tokens = [5, 6]
    d = str(tmp_path / "ckpt")
    m.save_pretrained(d)
    WhisperFeatureExtractor(feature_size=cfg.n_mels, ops=ops).save_pretrained(d)
    assert sorted(os.listdir(d)) == ["config.json", "generation_config.json", "model.safetensors",
                                     "preprocessor_config.json"]
    m2 = WhisperForConditionalGeneration.from_pretrained(d, ops=ops, torch_dtype=torch.float32,
                                                         attn_implementation="sdpa", low_cpu_mem_usage=True)
    for (n, p), (n2, p2) in zip(m.named_parameters(), m2.named_parameters()):
        assert n == n2 and torch.equal(p, p2), n
    assert m2.generation_config.max_length == 77 and m2.generation_config.suppress_tokens == [5, 6]
    assert m2.proj_out.weight is m2.model.decoder.embed_tokens.weight
    fe2 = WhisperFeatureExtractor.from_pretrained(d, ops=ops)
    assert fe2.feature_size == cfg.n_mels and fe2.sampling_rate == 16000
    with pytest.raises(OSError):
        WhisperForConditionalGeneration.from_pretrained("openai/whisper-tiny.en", ops=ops)
    with pytest.raises(TypeError):
        WhisperForConditionalGeneration.from_pretrained(d, ops=ops, not_an_argument=1)
    tf = pytest.importorskip("transformers")
    hf = tf.WhisperForConditionalGeneration.from_pretrained(d)               # the reference class reads our directory
    feats, ids, labels = _batch(cfg)
    with torch.no_grad():
        want = hf(input_features=feats, decoder_input_ids=ids).logits
        got = m2(input_features=feats, decoder_input_ids=ids).logits
    assert torch.allclose(want, got, atol=3e-4, rtol=1e-4)
    hf.save_pretrained(str(tmp_path / "hf"))                                  # ... and we read the reference's
    m3 = WhisperForConditionalGeneration.from_pretrained(str(tmp_path / "hf"), ops=ops)
    assert torch.equal(m3.model.encoder.conv1.weight, m.model.encoder.conv1.weight)
    # bf16 model (teacher / eval scripts load with torch_dtype=bfloat16): weights rounded, inference only
    mb = WhisperForConditionalGeneration.from_pretrained(d, ops=RefOps("cpu"), torch_dtype=torch.bfloat16)
    assert not any(p.requires_grad for p in mb.parameters())
    w = mb.model.decoder.layers[0].fc1.weight
    assert torch.equal(w, w.to(torch.bfloat16).float())


def test_freezing_after_construction_matches_reference_semantics():
    """run_distillation.py:1018-1040: `freeze_encoder()` then `embed_positions.requires_grad_(False)` on a constructed
    model; gradients of the rest are unchanged, frozen parameters get none, and the encoder backward is skipped."""
    from distil_whisper_amd.modeling import WhisperForConditionalGeneration
    ops = RefOps("cpu", lowp=torch.float32)
    cfg = wo.CONFIGS["micro"]
    sd = wo.init_state_dict(cfg, 11)
    feats, ids, labels = _batch(cfg, seed=12)
    full = WhisperForConditionalGeneration(cfg, ops=ops, state_dict=sd)
    full(input_features=feats, decoder_input_ids=ids, labels=labels).loss.backward()
    m = WhisperForConditionalGeneration(cfg, ops=ops, state_dict=sd)
    m.freeze_encoder()
    m.model.decoder.embed_positions.requires_grad_(False)
    m.gradient_checkpointing_enable()
    assert m.is_gradient_checkpointing and m.model.encoder.gradient_checkpointing
    calls = []
    orig = m.engine.backward_encoder
    m.engine.backward_encoder = lambda *a, **k: calls.append(1) or orig(*a, **k)
    out = m(input_features=feats, decoder_input_ids=ids, labels=labels)
    out.loss.backward()
    assert not calls
    for (n, p), (_, q) in zip(m.named_parameters(), full.named_parameters()):
        if n.startswith("model.encoder.") or n == "model.decoder.embed_positions.weight":
            assert p.grad is None, n
        else:
            assert torch.allclose(p.grad, q.grad, atol=1e-6, rtol=1e-5), n
    # the reference's optimizer grouping still sees the same parameter objects
    assert sum(p.requires_grad for p in m.parameters()) < sum(p.requires_grad for p in full.parameters())


def test_feature_extractor_pad_matches_transformers():
    tf = pytest.importorskip("transformers")
    from distil_whisper_amd.modeling import WhisperFeatureExtractor
    fe = WhisperFeatureExtractor(feature_size=80, ops=RefOps("cpu", lowp=torch.float32))
    ref = tf.WhisperFeatureExtractor(feature_size=80)
    rng = np.random.default_rng(0)
    items = [rng.standard_normal((80, 3000)).astype(np.float32) for _ in range(3)]
    for f in (ref, fe):
        with pytest.raises(ValueError, match="max_length is defined"):
            f.pad({"input_features": items}, padding="max_length", return_tensors="pt")
    for padding in ("longest", True):
        a = ref.pad({"input_features": items}, padding=padding, return_tensors="pt")
        b = fe.pad({"input_features": items}, padding=padding, return_tensors="pt")
        assert list(b.keys()) == list(a.keys())
        assert b["input_features"].dtype == a["input_features"].dtype and torch.equal(a["input_features"],
                                                                                      b["input_features"])
    c = fe.pad([{"input_features": x} for x in items], padding="longest", return_tensors="pt")
    assert torch.equal(c.input_features, b.input_features)
    with pytest.raises(ValueError):
        fe.pad({"input_features": [items[0], items[1][:, :2000]]}, padding="longest", return_tensors="pt")
    with pytest.raises(ValueError):
        fe.pad({"labels": [1]})
