"""distil_whisper_amd/labels.py against the reference's own `prepare_train_dataset` (run_distillation.py:1167-1229),
exec'd from /root/reference with stubbed tokenizer / feature extractor when the reference tree is present (build
container), and against known answers it produced (tests/golden/labels.json, oracle/gen_golden_labels.py)."""
import json
import os
import textwrap

import numpy as np
import pytest

from distil_whisper_amd.labels import prepare_train_labels

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/training/run_distillation.py"
TB, PREV, TPOS, MAXLEN = 50363, 50361, 3, 448          # multilingual layout: <|notimestamps|>, <|startofprev|>


def make_batch(rng, n, with_column):
    toks, prevs = [], []
    for _ in range(n):
        L = int(rng.integers(4, 300))
        ids = [50258, 50259, 50359] + rng.integers(0, 50257, size=L).tolist() + [50257]
        if rng.random() < 0.6:                        # pseudo-labels with timestamps
            for p in sorted(rng.integers(3, len(ids) - 1, size=4).tolist(), reverse=True):
                ids.insert(p, int(TB + 1 + rng.integers(0, 1500)))
        toks.append(ids)
        r = rng.random()
        prevs.append(None if r < 0.3 else rng.integers(0, 51000, size=int(rng.integers(1, 400))).tolist())
    return toks, (prevs if with_column else None)


def reference_labels(toks, prevs, seed, tp, cp):
    src = open(REF).read()
    a = src.index("    def prepare_train_dataset(batch):")
    b = src.index("    def prepare_eval_dataset(batch):")
    fn = textwrap.dedent(src[a:b])
    table = {f"s{i}": t for i, t in enumerate(toks)}

    class Out:
        def __init__(self, ids): self.input_ids = ids
    ns = {"np": np, "feature_extractor": lambda audio, sampling_rate: type("F", (), {"input_features": [0] * len(audio)})(),
          "sampling_rate": 16000, "train_text_column_name": "text", "tokenizer": lambda s, add_special_tokens: Out(list(table[s])),
          "use_pseudo_labels": True, "timestamp_ids": set(range(TB + 1, TB + 1502)), "timestamp_probability": tp,
          "timestamp_begin": TB, "timestamp_position": TPOS, "condition_on_prev_probability": cp,
          "prompt_cutoff_length": MAXLEN // 2, "max_label_length": MAXLEN, "decoder_prev_token_id": PREV}
    exec(fn, ns)
    batch = {"audio": [{"array": [0.0]}] * len(toks), "text": list(table)}
    if prevs is not None:
        batch["condition_on_prev"] = prevs
    np.random.seed(seed)
    return ns["prepare_train_dataset"](batch)["labels"]


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not present (GPU box)")
def test_matches_reference_function():
    rng = np.random.default_rng(0)
    for trial in range(40):
        toks, prevs = make_batch(rng, int(rng.integers(1, 9)), with_column=trial % 2 == 0)
        tp, cp = [(0.2, 0.2), (1.0, 1.0), (0.0, 1.0), (0.5, 0.9)][trial % 4]
        want = reference_labels(toks, prevs, 100 + trial, tp, cp)
        np.random.seed(100 + trial)
        got = prepare_train_labels(toks, prevs, timestamp_begin=TB, timestamp_position=TPOS, decoder_prev_token_id=PREV,
                                   timestamp_probability=tp, condition_on_prev_probability=cp, max_label_length=MAXLEN)
        assert got == want, trial
        assert all(len(x) <= MAXLEN + 1 for x in got)


def test_matches_golden_known_answers():
    gold = json.load(open(os.path.join(HERE, "golden", "labels.json")))
    for case in gold:
        np.random.seed(case["seed"])
        got = prepare_train_labels(case["tokens"], case["prevs"], timestamp_begin=TB, timestamp_position=TPOS,
                                   decoder_prev_token_id=PREV, timestamp_probability=case["tp"],
                                   condition_on_prev_probability=case["cp"], max_label_length=MAXLEN)
        assert got == case["labels"]


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not present (GPU box)")
def test_weight_decay_grouping_matches_reference_get_parameter_names():
    """run_distillation.py:760-778 + 1386-1391 exec'd on the drop-in module: the decay set the reference would build
    equals the oracle's name rule and the flat store's per-range weight-decay classification."""
    import torch
    import torch.nn as nn
    from distil_whisper_amd.modeling import WhisperForConditionalGeneration
    from oracle import whisper_oracle as wo
    from oracle.ref_ops import RefOps
    src = open(REF).read()
    a = src.index("def get_parameter_names(model, forbidden_layer_types, forbidden_module=None):")
    b = src.index("\n\n\n", a)
    ns = {}
    exec(src[a:b], ns)
    cfg_t = wo.CONFIGS["micro"]
    t_sd = wo.init_state_dict(cfg_t, 3)
    s_sd, cfg_s = wo.student_from_teacher(t_sd, cfg_t, 2, 1)
    model = WhisperForConditionalGeneration(cfg_s, ops=RefOps("cpu", lowp=torch.float32), state_dict=s_sd)
    decay = ns["get_parameter_names"](model, [nn.LayerNorm], forbidden_module=None)
    decay = {n for n in decay if "bias" not in n}                     # run_distillation.py:1391
    named = {n for n, _ in model.named_parameters()}                  # (tied proj_out.weight is listed once by torch)
    ref_decay = {n for n in named if n in decay}
    assert ref_decay == set(wo.decay_parameter_names({n: None for n in named}))
    # the flat store: every trainable entry's range carries weight decay iff the reference puts it in the decay group
    st = model.store
    segs = st.adam_segments(0.1)
    for name, (o, shape, kind) in st.entries.items():
        if o < st.train_start or name not in named:
            continue
        wd = next(w for lo, hi, w in segs if lo <= o < hi)
        assert (wd > 0) == (name in ref_decay), name
