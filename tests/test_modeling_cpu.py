"""The drop-in module surface (distil_whisper_amd/modeling.py) over the torch restatement of the kernels (CPU):
HF parameter names, tied head, LayerNorm instance types for the decay grouping, autograd integration (loss.backward()
fills .grad like the reference), in-place optimizer updates reaching the bf16 shadow weights."""
import pytest
import torch
import torch.nn as nn

from distil_whisper_amd.modeling import WhisperForConditionalGeneration
from oracle import whisper_oracle as wo
from oracle.ref_ops import RefOps



def _seq(model, *a, **k):
    """prompt + generated tokens (the `.sequences` of the reference's return_dict_in_generate=True output)"""
    return model.generate(*a, return_dict_in_generate=True, **k).sequences

def relerr(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()


def build(seed=9):
    cfg_t = wo.CONFIGS["micro"]
    t_sd = wo.init_state_dict(cfg_t, seed)
    s_sd, cfg_s = wo.student_from_teacher(t_sd, cfg_t, 2, 1)
    model = WhisperForConditionalGeneration(cfg_s, ops=RefOps("cpu", lowp=torch.float32), state_dict=s_sd)
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn(2, cfg_s.n_mels, 3000, generator=g) * 0.5
    b = wo.synthetic_batch(cfg_s, 2, seed=seed + 1, T=29, with_audio=False)
    return cfg_s, s_sd, model, feats, b["decoder_input_ids"], b["labels"]


def test_state_dict_keys_and_module_types_match_transformers():
    from transformers import WhisperConfig
    from transformers import WhisperForConditionalGeneration as HFModel
    cfg_s, s_sd, model, *_ = build()
    hc = WhisperConfig(vocab_size=cfg_s.vocab, num_mel_bins=cfg_s.n_mels, d_model=cfg_s.d_model,
                       encoder_layers=cfg_s.enc_layers, decoder_layers=cfg_s.dec_layers,
                       encoder_attention_heads=cfg_s.heads, decoder_attention_heads=cfg_s.heads,
                       encoder_ffn_dim=cfg_s.ffn, decoder_ffn_dim=cfg_s.ffn, pad_token_id=0, bos_token_id=0,
                       eos_token_id=0, decoder_start_token_id=1)
    hf = HFModel(hc)
    ours, theirs = model.state_dict(), hf.state_dict()
    assert set(ours) == set(theirs)
    for k in theirs:
        assert tuple(ours[k].shape) == tuple(theirs[k].shape), k
    assert model.proj_out.weight is model.model.decoder.embed_tokens.weight
    assert {n for n, p in model.named_parameters() if not p.requires_grad} == \
           {n for n, p in hf.named_parameters() if not p.requires_grad} == {"model.encoder.embed_positions.weight"}
    # the reference's decay grouping (run_distillation.py:760-778) sees the same LayerNorm modules
    ln_ours = sorted(n for n, m in model.named_modules() if isinstance(m, nn.LayerNorm))
    ln_hf = sorted(n for n, m in hf.named_modules() if isinstance(m, nn.LayerNorm))
    assert ln_ours == ln_hf
    # HF weights load straight in
    model.load_state_dict(theirs)
    assert torch.equal(model.store.p["model.decoder.layers.0.fc1.weight"], theirs["model.decoder.layers.0.fc1.weight"])


def test_forward_backward_matches_oracle_autograd():
    cfg_s, s_sd, model, feats, ids, labels = build()
    params = {k: v.clone().requires_grad_(k != "model.encoder.embed_positions.weight") for k, v in s_sd.items()}
    loss_ref, logits_ref, enc_ref = wo.model_forward(params, cfg_s, feats, ids, labels)
    (loss_ref * 1.0).backward()
    out = model(input_features=feats, decoder_input_ids=ids, labels=labels)
    assert abs(out.loss.item() - loss_ref.item()) < 1e-5 * abs(loss_ref.item())
    assert relerr(out.logits, logits_ref) < 1e-5
    assert relerr(out.encoder_last_hidden_state, enc_ref) < 1e-5
    out.loss.backward()
    for n, p in model.named_parameters():
        if not p.requires_grad:
            assert p.grad is None
            continue
        assert relerr(p.grad, params[n].grad) < 2e-4, n
    # labels only -> decoder inputs by shift_tokens_right (the shared-encoder teacher call of the reference)
    out2 = model(encoder_outputs=(out.encoder_last_hidden_state,), labels=labels)
    l2, _, _ = wo.model_forward(params, cfg_s, labels=labels, encoder_outputs=enc_ref.detach())
    assert abs(out2.loss.item() - l2.item()) < 1e-5 * abs(l2.item())


def test_optimizer_step_reaches_the_kernels():
    cfg_s, s_sd, model, feats, ids, labels = build()
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-2)
    l0 = model(input_features=feats, decoder_input_ids=ids, labels=labels).loss
    l0.backward()
    opt.step()
    opt.zero_grad()
    l1 = model(input_features=feats, decoder_input_ids=ids, labels=labels).loss
    assert l1.item() < l0.item() - 0.05  # the update was seen by the (shadow) GEMM weights


def test_error_behaviour_mirrors_reference():
    cfg_s, s_sd, model, feats, ids, labels = build()
    with pytest.raises(ValueError, match="mel input features to be of length 3000"):
        model(input_features=feats[:, :, :2000], decoder_input_ids=ids)
    with pytest.raises(ValueError, match="cannot exceed the maximum allowed length"):
        model(input_features=feats, labels=torch.zeros(2, 449, dtype=torch.long))


def test_greedy_generate_kv_cache_equals_prefix_redecode():
    cfg_s, s_sd, model, feats, ids, labels = build()
    a = _seq(model, feats, max_new_tokens=6, use_cache=True)
    b = _seq(model, feats, max_new_tokens=6, use_cache=False)
    assert a.shape == (2, 7) and int(a[0, 0]) == cfg_s.decoder_start_token_id
    assert torch.equal(a, b)
    # with a prompt
    prompt = torch.tensor([[cfg_s.decoder_start_token_id, 5, 9], [cfg_s.decoder_start_token_id, 7, 3]])
    c = _seq(model, feats, max_new_tokens=3, decoder_input_ids=prompt, use_cache=True)
    e = _seq(model, feats, max_new_tokens=3, decoder_input_ids=prompt, use_cache=False)
    assert torch.equal(c, e) and torch.equal(c[:, :3], prompt)
    # and the cached logits equal the teacher-forced forward logits at the same positions
    eng = model.engine
    enc, _ = eng.encode(feats, save=False)
    full, _ = eng.decode(c[:, :-1].contiguous(), enc, save=False)
    cache = eng.decode_init(enc, 2, c.shape[1])
    T = c.shape[1] - 1
    for t in range(T):
        step = eng.decode_step(c[:, t:t + 1], cache)[:, : cfg_s.vocab]
        ref = full[: 2 * T, : cfg_s.vocab].view(2, T, -1)[:, t]
        assert relerr(step, ref) < 1e-5


def test_collator_matches_oracle_and_lr_schedule():
    from distil_whisper_amd.collator import DataCollatorSpeechSeq2SeqWithPadding, linear_schedule_lr
    sot, prev, pad = 50257, 50360, 50256
    lists = [[prev, 11, 12, sot, 5, 6, 7], [sot, 8, 9], [sot] + list(range(100, 140))]
    coll = DataCollatorSpeechSeq2SeqWithPadding(decoder_start_token_id=sot, decoder_prev_token_id=prev,
                                                max_target_length=64, pad_token_id=pad, device="cpu")
    out = coll([{"labels": l} for l in lists])
    dec_in, labels = wo.collate(lists, sot, max_target_length=64, pad_token_id=pad)
    assert torch.equal(out["labels"], labels) and torch.equal(out["decoder_input_ids"], dec_in)
    # transformers' linear schedule, stepped num_processes times per optimizer step (run_distillation.py:1409-1415)
    from transformers import get_scheduler
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1e-4)
    sched = get_scheduler("linear", opt, num_warmup_steps=10 * 2, num_training_steps=50 * 2)
    for step in range(50):
        assert abs(sched.get_last_lr()[0] - linear_schedule_lr(step, 1e-4, 10, 50, 2)) < 1e-12
        sched.step(); sched.step()


def test_returned_gradients_do_not_alias_the_flat_buffer():
    """torch.autograd.grad results (and anything else that keeps the tensors the backward returns: hooks, DDP bucket
    views) must survive the next backward, which zeroes and rewrites the engine's flat gradient buffer."""
    cfg_s, s_sd, model, feats, ids, labels = build()
    w = model.get_parameter("model.decoder.layers.0.fc1.weight")
    b = model.get_parameter("model.encoder.layers.0.self_attn.v_proj.bias")
    loss = model(input_features=feats, decoder_input_ids=ids, labels=labels).loss
    gw, gb = torch.autograd.grad(loss, [w, b])
    keep_w, keep_b = gw.clone(), gb.clone()
    (model(input_features=feats * 0.3, decoder_input_ids=ids, labels=labels).loss * 5.0).backward()
    assert torch.equal(gw, keep_w) and torch.equal(gb, keep_b)
    assert not torch.equal(w.grad, keep_w)
