"""Parity where the arithmetic is stressed (round-5 review, "What's weak" 1-3).

Every other model-level fixture of this suite sits on random-init weights: near-uniform softmaxes, CE ~ ln V, KL ~ 0.1.
The fixtures here (oracle/gen_golden_sharp.py -> tests/golden/sharp_*.npz, recipe_large_v3.npz) come from the
`transformers` classes under the reference's `train_step` on "trained-like" weights (oracle.whisper_oracle.
sharpen_state_dict: attention-logit std ~4, logit std ~5 -- mean top-1 probability ~0.3 -- +-30 outlier residual channels,
x30 LayerNorm gains), computed in fp32 AND the way the reference trains (bf16 autocast student, bf16 teacher), with the
backward: probe gradients and the gradient norm of both runs.  So the HIP step is held

  * to ce / kl / loss of both reference runs (tolerances in `TOL` below),
  * gradient for gradient to the bf16-AUTOCAST run -- the apples-to-apples comparison SURVEY section 7 asks for --
    with the distance between the two REFERENCE runs (bf16-autocast vs fp32) printed next to it as the yardstick,
  * in the README's recipe mode (--freeze_encoder + shared encoder) at large-v3 dimensions,

and each default-on deviation of the product is priced by an A/B on these inputs: the deferred softmax maximum of the
attention forward (dw_debug_set key 23: 8 -> 0 = exact), gelu' kept in fp16 (engine.ffn_keeps_gelu_grad -> z kept in
bf16, gelu' re-evaluated in the backward).
"""
import os

import numpy as np
import pytest
import torch

from oracle import whisper_oracle as wo

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# relative tolerances on (ce, kl, loss) against BOTH reference runs.  north_star: loss within 1e-3.
TOL = {"ce": 1e-3, "kl": 1e-3, "loss": 1e-3}
# probe gradients against the bf16-autocast reference run: relative error / cosine.  The yardstick printed next to every
# probe is the distance between the two reference runs themselves (bf16 autocast vs fp32).
GRAD_REL, GRAD_COS = 0.06, 0.998


def relerr(a, b):
    a, b = torch.as_tensor(a).float().cpu().reshape(-1), torch.as_tensor(b).float().cpu().reshape(-1)
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def cosine(a, b):
    a, b = torch.as_tensor(a).float().cpu().reshape(-1), torch.as_tensor(b).float().cpu().reshape(-1)
    return (a @ b / (a.norm() * b.norm() + 1e-30)).item()


def probe_slice(g):
    g = g.reshape(-1)
    return g[:: max(1, g.numel() // 512)][:512]


@pytest.fixture(scope="module")
def ops():
    from distil_whisper_amd.ops_hip import HipOps
    return HipOps("cuda:0")


def _setup(ops, name, cfg_name, enc_s, dec_s, with_audio, **trainer_kw):
    from distil_whisper_amd.distill import DistillationTrainer
    g = np.load(os.path.join(GOLD, f"{name}.npz"))
    seed, B, recipe = int(g["seed"]), int(g["B"]), bool(int(g["recipe"]))
    cfg_t = wo.CONFIGS[cfg_name]
    t_sd = wo.sharpen_state_dict(wo.init_state_dict(cfg_t, seed), cfg_t)
    s_sd, cfg_s = wo.student_from_teacher(t_sd, cfg_t, enc_s, dec_s)
    b = wo.synthetic_batch(cfg_t, B, seed=seed + 1, with_audio=with_audio)
    filt = torch.tensor(wo.mel_filter_bank(cfg_t.n_mels), dtype=torch.float32).cuda().contiguous()
    tr = DistillationTrainer(ops, s_sd, cfg_s, t_sd, cfg_t, mel_filters=filt, freeze_encoder=recipe, share_encoder=recipe,
                             **trainer_kw)
    del t_sd, s_sd
    if with_audio:
        feats = tr.features(torch.tensor(b["audio"]).cuda())
    else:
        feats = (torch.randn(B, cfg_t.n_mels, 3000, generator=torch.Generator().manual_seed(seed + 2)) * 0.5).cuda()
    return g, tr, feats, b["decoder_input_ids"].cuda(), b["labels"].cuda()


def _step(tr, feats, ids, labels, probes):
    l = tr.forward_backward(feats, ids, labels).cpu()
    torch.cuda.synchronize()
    st = tr.student_store
    grads = {n: probe_slice(st.g[n].detach()).cpu().clone() for n in probes}
    gn = float(torch.sqrt((st.G[st.train_start:st.train_end].double() ** 2).sum()))
    return l, grads, gn


def _check(name, g, tr, feats, ids, labels, tol=TOL, grad_rel=GRAD_REL, grad_cos=GRAD_COS):
    probes = [str(x) for x in g["probe_names"]]
    l, grads, gn = _step(tr, feats, ids, labels, probes)
    print(f"\n[{name}] fixture regime: logit std {float(g['logit_std_fp32']):.2f}, mean top-1 probability "
          f"{float(g['s_max_prob_fp32']):.3f}; reference ce/kl/loss fp32 {float(g['ce_fp32']):.5f} / {float(g['kl_fp32']):.5f} / "
          f"{float(g['loss_fp32']):.5f}, bf16-autocast {float(g['ce_bf16']):.5f} / {float(g['kl_bf16']):.5f} / {float(g['loss_bf16']):.5f}")
    print(f"[{name}] HIP ce/kl/loss {l[0].item():.5f} / {l[1].item():.5f} / {l[2].item():.5f}")
    for i, k in enumerate(("ce", "kl", "loss")):
        for tag in ("fp32", "bf16"):
            ref = float(g[f"{k}_{tag}"])
            e = abs(l[i].item() - ref) / abs(ref)
            print(f"[{name}] {k} vs {tag}: rel {e:.2e} (reference runs apart: {abs(float(g[f'{k}_bf16']) - float(g[f'{k}_fp32'])) / abs(float(g[f'{k}_fp32'])):.2e})")
            assert e < tol[k], (name, k, tag, l[i].item(), ref)
    for tag in ("bf16", "fp32"):
        e = abs(gn - float(g[f"grad_norm_{tag}"])) / float(g[f"grad_norm_{tag}"])
        print(f"[{name}] gradient norm {gn:.5f} vs {tag} {float(g[f'grad_norm_{tag}']):.5f}: rel {e:.2e}")
        assert e < 2e-2, (name, "grad_norm", tag, gn)
    worst = 0.0
    for i, n in enumerate(probes):
        gb, gf = g[f"grad{i}_bf16"], g[f"grad{i}_fp32"]
        e_b, c_b = relerr(grads[n], gb), cosine(grads[n], gb)
        e_f = relerr(grads[n], gf)
        yard = relerr(gb, gf)
        worst = max(worst, e_b)
        print(f"[{name}] {n}: vs bf16-autocast run relerr {e_b:.3e} cos {c_b:.6f} | vs fp32 run {e_f:.3e} | the two reference runs apart {yard:.3e}")
        # apples to apples: within the stated bound, or no further from the bf16 reference than twice the distance
        # between the two reference runs (both are bf16 rounding noise on the same gradient)
        assert (e_b < grad_rel and c_b > grad_cos) or e_b < 2.0 * yard, (name, n, e_b, c_b, yard)
    return l, grads, gn


def _ab(name, g, ops, tr, feats, ids, labels, base, probes):
    """Each default-on deviation against its exact form, on the sharp inputs: what it moves in the losses and the probe
    gradients.  The yardstick is the distance between the two REFERENCE runs (bf16 autocast vs fp32) on the same quantity:
    a perturbation at the level of one bf16 rounding is amplified by 32 sharp layers to the same few per cent on an early
    layer's gradient whatever its source (measured at large-v3: reference runs 5.5e-2 apart on conv1.weight, the deferred
    maximum moves it by 4.4e-2, gelu' in fp16 by less).  Bound: no deviation moves anything further than the reference's
    own two precisions are apart."""
    l0, g0, gn0 = base
    yard_l = max(abs(float(g[f"{k}_bf16"]) - float(g[f"{k}_fp32"])) / abs(float(g[f"{k}_fp32"])) for k in ("ce", "kl", "loss"))
    yard_g = max(relerr(g[f"grad{i}_bf16"], g[f"grad{i}_fp32"]) for i in range(len(probes)))
    legs = []
    assert ops.lib.dw_debug_set(23, 0) == 0                       # exact running maximum in the attention forward
    try:
        legs.append(("attention: exact running maximum (key 23 = 0) instead of the deferred one", _step(tr, feats, ids, labels, probes)))
    finally:
        ops.lib.dw_debug_set(23, 8)
    keep = tr.student.ffn_keeps_gelu_grad
    tr.student.ffn_keeps_gelu_grad = False                        # z in bf16, gelu' re-evaluated in the backward
    try:
        legs.append(("FFN: z kept in bf16 and gelu' re-evaluated instead of gelu' kept in fp16", _step(tr, feats, ids, labels, probes)))
    finally:
        tr.student.ffn_keeps_gelu_grad = keep
    for what, (l1, g1, gn1) in legs:
        dl = [abs(l1[i].item() - l0[i].item()) / abs(l0[i].item()) for i in range(3)]
        dg = max(relerr(g1[n], g0[n]) for n in probes)
        print(f"[{name}] A/B {what}: ce/kl/loss move by {dl[0]:.1e} / {dl[1]:.1e} / {dl[2]:.1e} relative, gradient norm by "
              f"{abs(gn1 - gn0) / gn0:.1e}, worst probe gradient by {dg:.2e}  (reference runs apart: losses {yard_l:.1e}, worst probe {yard_g:.2e})")
        assert max(dl) < max(2e-4, yard_l) and dg < max(2e-2, yard_g), (name, what, dl, dg, yard_l, yard_g)


def test_sharp_tiny_en_step_audio_to_gradients(ops):
    """BASELINE config 1 dimensions (tiny.en 4/4 -> 4/1, B = 2), audio -> log-mel -> step, trained-like weights."""
    g, tr, feats, ids, labels = _setup(ops, "sharp_tiny", "tiny.en", 4, 1, True)
    base = _check("sharp_tiny", g, tr, feats, ids, labels)
    _ab("sharp_tiny", g, ops, tr, feats, ids, labels, base, [str(x) for x in g["probe_names"]])


def test_sharp_large_v3_step_with_gradients_against_the_bf16_autocast_run(ops):
    """The benchmark's models (large-v3-shaped 32/32 -> 32/2) at B = 3 on trained-like weights: losses against both
    reference runs, probe gradients against the bf16-autocast run, the deviations' A/B."""
    g, tr, feats, ids, labels = _setup(ops, "sharp_large_v3", "large-v3", 32, 2, False)
    base = _check("sharp_large_v3", g, tr, feats, ids, labels)
    _ab("sharp_large_v3", g, ops, tr, feats, ids, labels, base, [str(x) for x in g["probe_names"]])
    # the bench's trainer flags on the same inputs (side streams, padded teacher rows, dead positions left out / packed)
    lens = [int((row != -100).nonzero().max()) + 1 for row in labels.cpu()]
    tr.overlap_teacher = True
    tr.set_overlap_wgrad(True)
    for vl in (None, max(lens), lens):
        l = tr.forward_backward(feats, ids, labels, valid_len=vl).cpu()
        torch.cuda.synchronize()
        assert relerr(l[:3], base[0][:3]) < 1e-4, (vl, l, base[0])       # (fp32 summation order of the packed / trimmed passes)


def test_recipe_mode_large_v3_frozen_shared_encoder(ops):
    """README recipe at large-v3 dimensions: --freeze_encoder with the teacher's encoder shared (run_distillation.py:
    1018-1049, 1473-1478: one encoder forward, the teacher's decoder inputs rebuilt from the labels by shift_tokens_right),
    trained-like weights, B = 3: losses against both reference runs, decoder-side probe gradients against the
    bf16-autocast run; no gradient reaches the encoder."""
    g, tr, feats, ids, labels = _setup(ops, "recipe_large_v3", "large-v3", 32, 2, False)
    assert tr.freeze_encoder and tr.share_encoder
    _check("recipe_large_v3", g, tr, feats, ids, labels)
    st = tr.student_store
    assert not st.is_trainable("model.encoder.layers.0.fc1.weight")
