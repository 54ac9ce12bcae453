"""Host-logic tests (CPU): the engine's hand-written forward/backward, the trainer and the optimizer bookkeeping are
run over the torch restatement of the kernels (oracle/ref_ops.py, injected here -- the product always uses HipOps) and
compared with the CPU oracle's autograd.  In fp32 mode (`lowp=float32`) every bf16 rounding is the identity, so the
comparison is exact up to fp32 round-off and catches any error in the backward orchestration."""
import numpy as np
import pytest
import torch

from distil_whisper_amd.distill import DistillationTrainer
from distil_whisper_amd.engine import ParamStore, WhisperDims, WhisperEngine
from oracle import whisper_oracle as wo
from oracle.ref_ops import RefOps


def setup(seed=3, B=2, T=37, enc_s=2, dec_s=1):
    cfg_t = wo.CONFIGS["micro"]
    t_sd = wo.init_state_dict(cfg_t, seed)
    s_sd, cfg_s = wo.student_from_teacher(t_sd, cfg_t, enc_s, dec_s)
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn(B, cfg_t.n_mels, 3000, generator=g) * 0.5
    b = wo.synthetic_batch(cfg_t, B, seed=seed + 1, T=T, with_audio=False)
    batch = {"input_features": feats, "decoder_input_ids": b["decoder_input_ids"], "labels": b["labels"]}
    return cfg_t, cfg_s, t_sd, s_sd, batch


def oracle_grads(cfg_t, cfg_s, t_sd, s_sd, batch, shared=False, frozen_enc=False):
    params = {}
    for k, v in s_sd.items():
        rg = k != "model.encoder.embed_positions.weight" and not (frozen_enc and k.startswith("model.encoder."))
        params[k] = v.clone().requires_grad_(rg)
    loss, metrics, s_logits, t_logits, enc = wo.train_step(params, cfg_s, t_sd, cfg_t, batch, 2.0, 1.0, shared)
    loss.backward()
    return params, loss, metrics, s_logits, t_logits


def relerr(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()


@pytest.mark.parametrize("shared", [False, True])
def test_engine_fp32_matches_oracle_autograd(shared):
    cfg_t, cfg_s, t_sd, s_sd, batch = setup()
    params, loss, metrics, s_logits, t_logits = oracle_grads(cfg_t, cfg_s, t_sd, s_sd, batch, shared, shared)
    ops = RefOps("cpu", lowp=torch.float32)
    tr = DistillationTrainer(ops, s_sd, cfg_s, t_sd, cfg_t, freeze_encoder=shared, share_encoder=shared)
    # teacher weights are bf16-rounded by the trainer (teacher_dtype=bf16); use unrounded ones for the exact check
    tr.teacher_store.load_state_dict(t_sd, round_bf16=False)
    losses = tr.forward_backward(batch["input_features"], batch["decoder_input_ids"], batch["labels"])
    assert abs(losses[0].item() - metrics["ce_loss"].item()) < 2e-5 * abs(metrics["ce_loss"].item())
    assert abs(losses[1].item() - metrics["kl_loss"].item()) < 2e-4 * abs(metrics["kl_loss"].item()) + 1e-7
    assert abs(losses[2].item() - loss.item()) < 2e-5 * abs(loss.item())
    st = tr.student_store
    checked = 0
    for name, p in params.items():
        if p.grad is None:
            assert not st.is_trainable(name), name
            continue
        assert st.is_trainable(name), name
        e = relerr(st.g[name], p.grad)
        assert e < 2e-4, (name, e)
        checked += 1
    assert checked > 20
    # dummy k_proj.bias slots never receive gradient
    for n, (o, shape, kind) in st.entries.items():
        if kind == "zero" and o >= st.train_start:
            assert st.G[o:o + shape[0]].abs().max().item() == 0.0

    # optimizer: clip + AdamW (two param groups), two steps
    detached = {k: v.detach().clone() for k, v in params.items()}
    grads = {k: p.grad for k, p in params.items() if p.grad is not None}
    state = {}
    decay = set(wo.decay_parameter_names(detached))
    tr.weight_decay = 0.1
    tr.segments = st.adam_segments(0.1)
    for step in (1, 2):
        wo.clip_and_adamw(detached, grads, state, step=step, weight_decay=0.1, decay_names=decay)
        tr.optimizer_step()
    for name in grads:
        assert relerr(st.p[name], detached[name]) < 5e-6, name
        assert relerr(st.s[name], detached[name]) < 5e-6, name
    # the packed conv weights follow the master weights
    if not shared:
        assert relerr(st.conv2_packed, ops.pack_conv_weight(st.p["model.encoder.conv2.weight"], 3 * cfg_s.d_model)) == 0


def test_engine_bf16_emulation_close_to_fp32_oracle():
    """With bf16 rounding at the kernel boundaries (what the HIP path does) the loss stays within the 1e-3 budget."""
    cfg_t, cfg_s, t_sd, s_sd, batch = setup(seed=5)
    params, loss, metrics, s_logits, t_logits = oracle_grads(cfg_t, cfg_s, t_sd, s_sd, batch)
    tr = DistillationTrainer(RefOps("cpu"), s_sd, cfg_s, t_sd, cfg_t)
    losses = tr.forward_backward(batch["input_features"], batch["decoder_input_ids"], batch["labels"])
    assert abs(losses[2].item() - loss.item()) < 2e-3 * abs(loss.item())
    st = tr.student_store
    for name in ("model.decoder.layers.0.fc1.weight", "model.encoder.layers.0.self_attn.q_proj.weight",
                 "model.encoder.conv1.weight", "model.decoder.embed_tokens.weight"):
        assert relerr(st.g[name], params[name].grad) < 0.08, name


def test_param_store_layout_and_state_dict_roundtrip():
    cfg_t, cfg_s, t_sd, s_sd, _ = setup()
    ops = RefOps("cpu")
    st = ParamStore(ops, WhisperDims.from_any(cfg_s), s_sd, trainable=True, frozen_prefixes=("model.encoder.",))
    sd = st.state_dict()
    assert set(sd) == set(s_sd) | {"proj_out.weight"}  # HF key names, tied head
    for k, v in s_sd.items():
        assert torch.equal(sd[k], v), k
    assert not st.is_trainable("model.encoder.layers.0.fc1.weight")
    assert st.is_trainable("model.decoder.layers.0.fc1.weight")
    segs = st.adam_segments(0.1)
    assert segs[0][0] == st.train_start and segs[-1][1] == st.train_end
    assert all(segs[i][1] == segs[i + 1][0] for i in range(len(segs) - 1))
    av = st.attn_views("model.decoder.layers.0.self_attn")
    D = cfg_s.d_model
    assert torch.equal(av["wqkv"][D:2 * D].float(), s_sd["model.decoder.layers.0.self_attn.k_proj.weight"].bfloat16().float())
    assert av["bqkv"][D:2 * D].abs().max().item() == 0.0


def test_gradient_accumulation_equals_mean_of_microbatch_gradients():
    cfg_t, cfg_s, t_sd, s_sd, batch = setup(seed=8, B=2)
    ops = RefOps("cpu", lowp=torch.float32)
    f, d, l = batch["input_features"], batch["decoder_input_ids"], batch["labels"]
    a = DistillationTrainer(ops, s_sd, cfg_s, t_sd, cfg_t)
    b = DistillationTrainer(ops, s_sd, cfg_s, t_sd, cfg_t)
    c = DistillationTrainer(ops, s_sd, cfg_s, t_sd, cfg_t)
    a.forward_backward(f[:1], d[:1], l[:1])
    b.forward_backward(f[1:], d[1:], l[1:])
    mean_g = 0.5 * (a.student_store.G + b.student_store.G)
    c.forward_backward(f[:1], d[:1], l[:1], zero_grad=True)
    c.forward_backward(f[1:], d[1:], l[1:], zero_grad=False)
    assert relerr(0.5 * c.student_store.G, mean_g) < 1e-6
    # and the fused optimizer applies the 1/n average
    ref = DistillationTrainer(ops, s_sd, cfg_s, t_sd, cfg_t)
    ref.student_store.G.copy_(mean_g)
    ref.optimizer_step()
    d2 = DistillationTrainer(ops, s_sd, cfg_s, t_sd, cfg_t)
    d2.train_step_accumulated([(f[:1], d[:1], l[:1]), (f[1:], d[1:], l[1:])])
    assert relerr(d2.student_store.P, ref.student_store.P) < 1e-6


def test_freeze_embed_positions_and_recipe_layout():
    """--freeze_embed_positions (run_distillation.py:1034-1040) + --freeze_encoder: frozen tensors leave the trainable
    range, everything else still matches autograd."""
    cfg_t, cfg_s, t_sd, s_sd, batch = setup(seed=4)
    params = {}
    for k, v in s_sd.items():
        frozen = k.startswith("model.encoder.") or k == "model.decoder.embed_positions.weight"
        params[k] = v.clone().requires_grad_(not frozen)
    loss, metrics, *_ = wo.train_step(params, cfg_s, t_sd, cfg_t, batch, 2.0, 1.0, True)
    loss.backward()
    tr = DistillationTrainer(RefOps("cpu", lowp=torch.float32), s_sd, cfg_s, t_sd, cfg_t, freeze_encoder=True,
                             share_encoder=True, freeze_embed_positions=True)
    tr.teacher_store.load_state_dict(t_sd, round_bf16=False)
    losses = tr.forward_backward(batch["input_features"], batch["decoder_input_ids"], batch["labels"])
    assert abs(losses[2].item() - loss.item()) < 2e-5 * abs(loss.item())
    st = tr.student_store
    assert not st.is_trainable("model.decoder.embed_positions.weight")
    assert st.dec_start == st.train_start  # only decoder-side tensors are trainable
    for name, p in params.items():
        if p.grad is None:
            assert not st.is_trainable(name), name
        else:
            assert relerr(st.g[name], p.grad) < 2e-4, name
    tr.optimizer_step()  # frozen range untouched
    assert torch.equal(st.p["model.decoder.embed_positions.weight"], s_sd["model.decoder.embed_positions.weight"])
    assert torch.equal(st.p["model.encoder.layers.0.fc1.weight"], s_sd["model.encoder.layers.0.fc1.weight"])


def test_teacher_decoder_over_padded_gemm_rows_gives_the_same_logits():
    """`pad_gemm_rows`: the forward-only (teacher) decoder pass runs its projections over a row count padded to a
    multiple of 320 (B*T = 74 -> 320 here; 14304 -> 14400 in the benchmark) with garbage in the pad rows -- every
    operation between the embedding and the logits is row-local, so the B*T valid rows of the logits are unchanged, and
    the trainer's loss with it."""
    cfg_t, cfg_s, t_sd, s_sd, batch = setup()
    ops = RefOps("cpu", lowp=torch.float32)
    dims = WhisperDims.from_any(cfg_t)
    eng = WhisperEngine(ops, ParamStore(ops, dims, t_sd, trainable=False), torch.float32)
    enc, _ = eng.encode(batch["input_features"], save=False)
    ids = batch["decoder_input_ids"]
    want, _ = eng.decode(ids, enc, save=False)
    eng.pad_gemm_rows, eng.pad_gemm_rows_min, eng.pad_gemm_rows_slack = True, 1, 100.0
    assert eng._gemm_rows(ids.numel(), False) == 320 and eng._gemm_rows(ids.numel(), True) == ids.numel()
    got, _ = eng.decode(ids, enc, save=False)
    R = ids.numel()
    assert torch.equal(got[:R], want[:R])
    eng.pad_gemm_rows_slack = 1 / 32
    assert eng._gemm_rows(14304, False) == 14400 and eng._gemm_rows(14000, False) == 14080 and eng._gemm_rows(100, False) == 100
    a = DistillationTrainer(ops, s_sd, cfg_s, t_sd, cfg_t)
    b = DistillationTrainer(ops, s_sd, cfg_s, t_sd, cfg_t, pad_teacher_rows=True)
    b.teacher.pad_gemm_rows_min, b.teacher.pad_gemm_rows_slack = 1, 100.0
    la = a.forward_backward(batch["input_features"], batch["decoder_input_ids"], batch["labels"])
    lb = b.forward_backward(batch["input_features"], batch["decoder_input_ids"], batch["labels"])
    assert torch.equal(la, lb)


def test_lm_head_backward_over_zero_padded_rows_gives_the_same_gradients():
    """`pad_lm_rows`: in a training pass hf / logits get zero rows up to a multiple of 320 so that dhf = dlogits . E runs
    over M = 320 k rows (one round of 320-row GEMM tiles at the benchmark's 14304 -> 14400); zero rows add nothing to
    dE = dlogits^T . hf and their dhf rows are never read: every gradient equals the unpadded run's."""
    cfg_t, cfg_s, t_sd, s_sd, batch = setup()
    ops = RefOps("cpu", lowp=torch.float32)

    def grads(pad):
        tr = DistillationTrainer(ops, s_sd, cfg_s, t_sd, cfg_t)
        tr.student.pad_lm_rows = pad
        tr.student.pad_gemm_rows_min, tr.student.pad_gemm_rows_slack = 1, 100.0
        losses = tr.forward_backward(batch["input_features"], batch["decoder_input_ids"], batch["labels"])
        return losses, tr.student_store.G.clone()
    l0, g0 = grads(False)
    l1, g1 = grads(True)
    assert torch.equal(l0, l1)
    assert relerr(g1, g0) < 1e-6     # (dE sums 320 rows instead of 128: zero terms, but the split of K into slices differs)


def test_batch_without_labels_skips_the_optimizer_step():
    """All labels -100: the reference's loss is 0/0 = NaN (run_distillation.py:1486-1493).  Here the losses report NaN,
    the gradient is zero and the optimizer step is skipped by the device-side gate (dw_adam_tick) -- parameters,
    moments and the step count stay put instead of drifting by momentum / weight decay; the next normal step equals
    the step a trainer takes that never saw the empty batch."""
    cfg_t, cfg_s, t_sd, s_sd, batch = setup(seed=6)
    ops = RefOps("cpu", lowp=torch.float32)
    f, d, l = batch["input_features"], batch["decoder_input_ids"], batch["labels"]
    a = DistillationTrainer(ops, s_sd, cfg_s, t_sd, cfg_t, weight_decay=0.1)
    b = DistillationTrainer(ops, s_sd, cfg_s, t_sd, cfg_t, weight_decay=0.1)
    a.train_step(f, d, l)
    b.train_step(f, d, l)
    p1 = a.student_store.P.clone()
    losses = a.train_step(f, d, torch.full_like(l, -100))
    assert torch.isnan(losses[:3]).all() and losses[3].item() == 0
    assert torch.equal(a.student_store.P, p1) and a.step_count == 1
    a.train_step(f, d, l)
    b.train_step(f, d, l)
    assert a.step_count == b.step_count == 2
    assert torch.equal(a.student_store.P, b.student_store.P)
    # accumulation: one empty micro-batch neither poisons the reported loss nor blocks the step
    c = DistillationTrainer(ops, s_sd, cfg_s, t_sd, cfg_t)
    out = c.train_step_accumulated([(f, d, torch.full_like(l, -100)), (f, d, l)])
    assert torch.isfinite(out[:3]).all() and c.step_count == 1


def test_resume_restores_every_hyper_parameter():
    """load_state_dict restores weight decay (and the per-range decay segments built from it), the clip norm, lr,
    betas, eps, temperature, kl_weight and the step count: a trainer constructed with DIFFERENT values continues
    exactly like the one that saved the state."""
    cfg_t, cfg_s, t_sd, s_sd, batch = setup(seed=7)
    ops = RefOps("cpu", lowp=torch.float32)
    f, d, l = batch["input_features"], batch["decoder_input_ids"], batch["labels"]
    a = DistillationTrainer(ops, s_sd, cfg_s, t_sd, cfg_t, weight_decay=0.2, max_grad_norm=0.05, lr=3e-4,
                            betas=(0.8, 0.99), eps=1e-6, temperature=3.0, kl_weight=0.5)
    a.train_step(f, d, l)
    state = a.state_dict()
    a.train_step(f, d, l)
    r = DistillationTrainer(ops, s_sd, cfg_s, t_sd, cfg_t)       # defaults: wd 0, clip 1.0, lr 1e-4 ...
    r.load_state_dict(state)
    assert (r.weight_decay, r.max_grad_norm, r.lr, r.betas, r.eps) == (0.2, 0.05, 3e-4, (0.8, 0.99), 1e-6)
    assert r.segments == a.segments and any(wd == 0.2 for _, _, wd in r.segments)
    r.train_step(f, d, l)
    assert r.step_count == a.step_count == 2
    assert relerr(r.student_store.P, a.student_store.P) < 1e-7


def test_eval_step_is_the_reference_eval_step():
    """DistillationTrainer.eval_step = run_distillation.py:1498-1522: forward only, temperature 1, 0.8 CE + kl_weight KL;
    gradients and parameters untouched."""
    cfg_t, cfg_s, t_sd, s_sd, batch = setup(seed=8)
    ops = RefOps("cpu", lowp=torch.float32)
    tr = DistillationTrainer(ops, s_sd, cfg_s, t_sd, cfg_t, temperature=2.0, kl_weight=0.6)
    tr.teacher_store.load_state_dict(t_sd, round_bf16=False)
    f, d, l = batch["input_features"], batch["decoder_input_ids"], batch["labels"]
    g0 = tr.student_store.G.clone()
    ev = tr.eval_step(f, d, l)
    params = {k: v.clone() for k, v in s_sd.items()}
    loss, metrics, *_ = wo.train_step(params, cfg_s, t_sd, cfg_t, batch, 1.0, 0.6, False)
    assert abs(ev[0].item() - metrics["ce_loss"].item()) < 2e-5 * metrics["ce_loss"].item()
    assert abs(ev[1].item() - metrics["kl_loss"].item()) < 2e-4 * metrics["kl_loss"].item() + 1e-7
    assert abs(ev[2].item() - loss.item()) < 2e-5 * loss.item()
    assert torch.equal(tr.student_store.G, g0)


@pytest.mark.parametrize("shared", [False, True])
def test_dead_decoder_positions_can_be_left_out(shared):
    """trim_dead_positions: with every label behind position `valid_len` equal to -100, running the decoders, the LM
    heads and the loss over the first `valid_len` positions only gives the loss and the gradients of the full-length
    step (the reference's collator pads every batch to 448 positions, run_distillation.py:405-478).  Prompt positions
    masked at the START of a row are not dead (later positions attend to them) and stay."""
    cfg_t, cfg_s, t_sd, s_sd, batch = setup(T=37)
    labels = batch["labels"].clone()
    labels[0, 19:] = -100
    labels[1, 11:] = -100
    labels[1, :3] = -100                       # a masked prompt prefix
    valid_len = 19
    ops = RefOps("cpu", lowp=torch.float32)

    def run(vl):
        tr = DistillationTrainer(ops, s_sd, cfg_s, t_sd, cfg_t, freeze_encoder=shared, share_encoder=shared)
        losses = tr.forward_backward(batch["input_features"], batch["decoder_input_ids"], labels, valid_len=vl)
        return tr, losses.clone(), tr.student_store.G.clone()
    tr0, l0, g0 = run(None)
    tr1, l1, g1 = run(valid_len)
    assert l0[3].item() == l1[3].item() == float((labels != -100).sum())
    assert torch.allclose(l0[:3], l1[:3], rtol=1e-6, atol=0)
    assert relerr(g1, g0) < 1e-6
    # the embedding rows of the dead positions get exactly zero gradient either way
    gp = tr1.student_store.g["model.decoder.embed_positions.weight"]
    assert float(gp[valid_len:].abs().max()) == 0.0 and float(gp[:valid_len].abs().max()) > 0.0
    # eval_step and a whole optimizer step agree too; a too-long valid_len is clamped
    e0 = tr0.eval_step(batch["input_features"], batch["decoder_input_ids"], labels)
    e1 = tr1.eval_step(batch["input_features"], batch["decoder_input_ids"], labels, valid_len=valid_len)
    assert torch.allclose(e0[:3], e1[:3], rtol=1e-6, atol=0)
    tr0.train_step(batch["input_features"], batch["decoder_input_ids"], labels)
    tr1.train_step(batch["input_features"], batch["decoder_input_ids"], labels, valid_len=10 ** 6)
    tr2 = run(None)[0]
    tr2.train_step(batch["input_features"], batch["decoder_input_ids"], labels, valid_len=valid_len)
    assert torch.equal(tr0.student_store.P, tr1.student_store.P)
    assert relerr(tr2.student_store.P, tr0.student_store.P) < 1e-7
    # per-sequence lengths: the teacher's decoder, both LM heads and the loss over the packed live rows (30 of 2 x 19)
    seen = []
    gemm = ops.gemm
    ops.gemm = lambda a, *r, **k: (seen.append(a.shape[0]), gemm(a, *r, **k))[1]
    try:
        tr3, l3, g3 = run([19, 11])
    finally:
        ops.gemm = gemm
    assert 30 in seen                                           # the packed GEMMs ran
    assert torch.allclose(l0, l3, rtol=1e-6, atol=0) and relerr(g3, g0) < 1e-6
    e3 = tr3.eval_step(batch["input_features"], batch["decoder_input_ids"], labels, valid_len=[19, 11])
    assert torch.allclose(e0[:3], e3[:3], rtol=1e-6, atol=0)
    with pytest.raises(ValueError):
        tr3.forward_backward(batch["input_features"], batch["decoder_input_ids"], labels, valid_len=[19])


def test_collator_reports_the_last_labelled_position():
    from distil_whisper_amd.collator import DataCollatorSpeechSeq2SeqWithPadding
    feats = [{"labels": [50257, 50362, 11, 12, 13, 50256]}, {"labels": [50257, 50362, 21, 50256]}]
    col = DataCollatorSpeechSeq2SeqWithPadding(max_target_length=16, device="cpu", report_valid_len=True)
    b = col(feats)
    last = int((b["labels"] != -100).any(0).nonzero().max())
    assert b["valid_len"] == last + 1 == 5
    assert "valid_len" not in DataCollatorSpeechSeq2SeqWithPadding(max_target_length=16, device="cpu")(feats)
    per = DataCollatorSpeechSeq2SeqWithPadding(max_target_length=16, device="cpu", report_valid_len="per_sequence")(feats)
    assert per["valid_len"] == [1 + int((row != -100).nonzero().max()) for row in per["labels"]] == [5, 3]


def test_live_rows_index_and_scatter_buffers():
    """engine.LiveRows lists row b*T + t for t < lens[b], sequence by sequence (lengths clamped to [1, T]); the scatter
    targets of the packed pass are zero-initialised, shared by every request they can hold, and never replaced (a
    captured step has their addresses baked in)."""
    from distil_whisper_amd.engine import LiveRows
    idx = LiveRows.host_index([3, 0, 9, 2], 5)
    assert idx.dtype == torch.int32 and idx.tolist() == [0, 1, 2, 5, 10, 11, 12, 13, 14, 15, 16]
    live = LiveRows.build([3, 0, 9, 2], 5, "cpu")
    assert (live.n, live.B, live.T) == (11, 4, 5) and torch.equal(live.idx, idx)
    cfg_t, cfg_s, t_sd, s_sd, batch = setup()
    ops = RefOps("cpu", lowp=torch.float32)
    eng = DistillationTrainer(ops, s_sd, cfg_s, t_sd, cfg_t).teacher
    a = eng._scatter_buf(100, 384)
    assert a.shape == (128, 384) and float(a.abs().max()) == 0.0
    assert eng._scatter_buf(70, 384) is a and eng._scatter_buf(128, 384) is a
    b = eng._scatter_buf(200, 384)
    assert b is not a and b.shape[0] == 256 and eng._scatter_buf(100, 384) is a and eng._scatter_buf(130, 384) is b
    assert eng._scatter_buf(100, 128) is not a


def test_live_rows_filled_with_dead_rows_and_integer_spellings():
    """A row list filled up with dead rows of the rectangle (quantised plan keys of train_step_graphed) gives the same loss
    and gradients; valid_len may arrive as a numpy integer or a 0-d tensor."""
    import numpy as np
    from distil_whisper_amd.engine import LiveRows
    idx = LiveRows.host_index([3, 1], 5, fill_to=7)
    assert idx.tolist() == [0, 1, 2, 5, 3, 4, 6] and len(set(idx.tolist())) == 7
    with pytest.raises(ValueError):
        LiveRows.host_index([5, 5], 5, fill_to=11)
    cfg_t, cfg_s, t_sd, s_sd, batch = setup(T=37)
    labels = batch["labels"].clone()
    labels[0, 19:] = -100
    labels[1, 11:] = -100
    ops = RefOps("cpu", lowp=torch.float32)

    def run(vl, live=None):
        tr = DistillationTrainer(ops, s_sd, cfg_s, t_sd, cfg_t)
        losses = tr.forward_backward(batch["input_features"], batch["decoder_input_ids"], labels, valid_len=vl, _live=live)
        return losses.clone(), tr.student_store.G.clone()
    l0, g0 = run(None)
    for vl in (np.int64(19), torch.tensor(19), 19):
        l1, g1 = run(vl)
        assert torch.allclose(l0, l1, rtol=1e-6, atol=0) and relerr(g1, g0) < 1e-6
    # rectangle of 24 positions, 30 live rows filled up to 40 with dead ones
    Te = 24
    h = LiveRows.host_index([19, 11], Te, fill_to=40)
    l2, g2 = run(Te, LiveRows(h, 40, 2, Te))
    assert l2[3].item() == l0[3].item()
    assert torch.allclose(l0, l2, rtol=1e-6, atol=0) and relerr(g2, g0) < 1e-6


def test_packed_training_decoder_layers_equal_rectangular_layers():
    """engine.pack_train_layers: with per-sequence label lengths the student's decoder layers run their row-local work (GEMMs,
    LayerNorm, the backward of both) over the packed live rows and only attention over the (batch, position) rectangle; losses and
    every gradient equal the rectangular layers' (fp32 restatement: summation order only), also with filler rows in the list and
    over two steps (stale dead rows of the per-layer rectangles from the first step are harmless in the second)."""
    from distil_whisper_amd.engine import LiveRows, WhisperEngine
    cfg_t, cfg_s, t_sd, s_sd, batch = setup(T=37)
    labels = batch["labels"].clone()
    labels[0, 19:] = -100
    labels[1, 11:] = -100
    labels2 = batch["labels"].clone()
    labels2[0, 7:] = -100
    labels2[1, 23:] = -100
    ops = RefOps("cpu", lowp=torch.float32)
    Te = 24

    def run(pack):
        WhisperEngine.pack_train_layers = pack
        try:
            tr = DistillationTrainer(ops, s_sd, cfg_s, t_sd, cfg_t)
            out = []
            for lab, lens, fill in ((labels, [19, 11], 40), (labels2, [7, 23], None)):
                h = LiveRows.host_index(lens, Te, fill_to=fill)
                losses = tr.forward_backward(batch["input_features"], batch["decoder_input_ids"], lab, valid_len=Te,
                                             _live=LiveRows(h, h.numel(), 2, Te))
                out.append((losses.clone(), tr.student_store.G.clone()))
            rows = tr.student._last_decode_rows
            return out, rows
        finally:
            WhisperEngine.pack_train_layers = True
    a, rows_a = run(True)
    b, rows_b = run(False)
    assert rows_a == 30 and rows_b == 2 * Te          # (second step: 7 + 23 live rows against the 48-row rectangle)
    for (la, ga), (lb, gb) in zip(a, b):
        assert torch.allclose(la, lb, rtol=1e-6, atol=0) and relerr(ga, gb) < 1e-6


def test_live_rows_sequence_table():
    """LiveRows.seq_table: (start, length) of every sequence's packed rows -- what dw_attn_fwd_varlen walks -- consistent with
    host_index; filler rows are cut into pseudo-sequences of at most T rows and a captured plan's table is padded with empty
    entries to its fixed size."""
    from distil_whisper_amd.engine import LiveRows
    lens, T = [5, 0, 8, 1], 8                      # (lengths are clamped to [1, T])
    st, ln = LiveRows.seq_table(lens, T)
    assert ln.tolist() == [5, 1, 8, 1] and st.tolist() == [0, 5, 6, 14] and st.dtype == torch.int32
    idx = LiveRows.host_index(lens, T)
    for b, (s0, n) in enumerate(zip(st.tolist(), ln.tolist())):
        assert idx[s0:s0 + n].tolist() == [b * T + t for t in range(n)]
    st, ln = LiveRows.seq_table(lens, T, fill_to=15 + 11, entries=8)
    assert ln.tolist() == [5, 1, 8, 1, 8, 3, 0, 0] and st.tolist() == [0, 5, 6, 14, 15, 23, 26, 26]
    assert int(ln.sum()) == LiveRows.host_index(lens, T, fill_to=26).numel()
    with pytest.raises(ValueError):
        LiveRows.seq_table(lens, T, fill_to=26, entries=5)
    live = LiveRows.build(lens, T, "cpu")
    assert live.max_q == 8 and live.seq_len.tolist() == [5, 1, 8, 1] and live.attn_flops == (91.0, 15.0)
