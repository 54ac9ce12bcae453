"""Long-form scheduler (distil_whisper_amd/longform.py) and the graph-capable greedy decoder (decoding.py), CPU side:
chunk windows and token stitching against known answers produced by the `transformers` functions the reference's
pipeline path runs (tests/golden/longform.json, made by oracle/gen_golden_longform.py), and against those functions
directly when `transformers` is importable; the transcriber end to end against per-window generate + stitching."""
import json
import os

import numpy as np
import pytest
import torch

from distil_whisper_amd.longform import LongFormTranscriber, chunk_spans, merge_sequences
from distil_whisper_amd.decoding import GreedyDecoder
from distil_whisper_amd.modeling import WhisperFeatureExtractor, WhisperForConditionalGeneration
from oracle import whisper_oracle as wo
from oracle.ref_ops import RefOps


def _seq(model, *a, **k):
    """prompt + generated tokens (the `.sequences` of the reference's return_dict_in_generate=True output)"""
    return model.generate(*a, return_dict_in_generate=True, **k).sequences

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "longform.json")))


def test_chunk_spans_match_reference_chunk_iter():
    for case in GOLD["chunks"]:
        got = [list(s) for s in chunk_spans(*case["args"])]
        assert got == case["spans"], case["args"]
    # 5-minute clip, 30 s windows, 5 s strides: 15 windows per clip
    assert len(chunk_spans(4800000, 480000, 80000, 80000)) == 15
    with pytest.raises(ValueError, match="superior to stride"):
        chunk_spans(1000, 100, 50, 50)


def test_merge_sequences_match_reference_stitching():
    for case in GOLD["merges"]:
        assert merge_sequences(case["sequences"]) == case["merged"]
    assert merge_sequences([]) == []


def test_against_transformers_functions_directly():
    tw = pytest.importorskip("transformers.models.whisper.tokenization_whisper")
    asr = pytest.importorskip("transformers.pipelines.automatic_speech_recognition")
    rng = np.random.default_rng(3)
    for _ in range(60):
        n_seq = int(rng.integers(1, 6))
        base = rng.integers(0, 30, size=400).tolist()
        seqs, pos = [], 0
        for _ in range(n_seq):
            w = int(rng.integers(1, 50))
            seqs.append(base[max(0, pos - int(rng.integers(0, 12))): pos + w])
            pos += w
        assert merge_sequences(seqs) == list(tw._find_longest_common_sequence(seqs))

    class FE:
        sampling_rate = 16000

        def __call__(self, chunk, **kw):
            return {"n": len(chunk)}
    for _ in range(40):
        c = int(rng.integers(10, 300))
        sl, sr = int(rng.integers(0, c // 3)), int(rng.integers(0, c // 3))
        n = int(rng.integers(1, 2000))
        ref = [(o["stride"], o["is_last"], o["n"]) for o in asr.chunk_iter(np.zeros(n, np.float32), FE(), c, sl, sr)]
        got = [((length, l, r), last, length) for _, length, l, r, last in chunk_spans(n, c, sl, sr)]
        assert got == ref, (n, c, sl, sr)


def _model(seed=4):
    cfg_t = wo.CONFIGS["micro"]
    t_sd = wo.init_state_dict(cfg_t, seed)
    s_sd, cfg_s = wo.student_from_teacher(t_sd, cfg_t, 2, 1)
    ops = RefOps("cpu", lowp=torch.float32)
    model = WhisperForConditionalGeneration(cfg_s, ops=ops, state_dict=s_sd)
    fe = WhisperFeatureExtractor(feature_size=cfg_s.n_mels, ops=ops)
    return cfg_s, model, fe


def test_greedy_decoder_matches_prefix_redecode_with_suppression_and_eos():
    cfg, model, fe = _model()
    g = torch.Generator().manual_seed(1)
    feats = torch.randn(3, cfg.n_mels, 3000, generator=g) * 0.5
    ref = _seq(model, feats, max_new_tokens=7, use_cache=False, suppress_tokens=[3, 4, 5], begin_suppress_tokens=[6])
    eos = int(ref[0, 3])              # a token the model really emits -> exercises the EOS fill
    ref = _seq(model, feats, max_new_tokens=7, use_cache=False, suppress_tokens=[3, 4, 5], begin_suppress_tokens=[6],
                         eos_token_id=eos)
    enc, _ = model.engine.encode(feats, save=False)
    dec = GreedyDecoder(model.engine, 3, 8, eos_token_id=eos, suppress_tokens=[3, 4, 5], begin_suppress_tokens=[6],
                        use_graphs=False, check_every=1)
    prompt = torch.full((3, 1), cfg.decoder_start_token_id, dtype=torch.long)
    out = dec.run(enc, prompt, 7)
    assert torch.equal(out, ref)
    assert not bool(((out[:, 1:] >= 3) & (out[:, 1:] <= 5)).any()) and not bool((out[:, 1] == 6).any())
    # the decoder is reusable: second batch through the same buffers
    feats2 = torch.randn(3, cfg.n_mels, 3000, generator=g) * 0.5
    enc2, _ = model.engine.encode(feats2, save=False)
    ref2 = _seq(model, feats2, max_new_tokens=7, use_cache=False, suppress_tokens=[3, 4, 5],
                          begin_suppress_tokens=[6], eos_token_id=eos)
    assert torch.equal(dec.run(enc2, prompt, 7), ref2)
    with pytest.raises(ValueError, match="exceeds the decoder's max_len"):
        dec.run(enc, prompt, 9)


def test_transcriber_equals_per_window_generate_plus_stitching():
    cfg, model, fe = _model()
    rng = np.random.default_rng(5)
    audios = [0.1 * rng.standard_normal(n).astype(np.float32) for n in (1_000_000, 300_000, 480_000)]
    first_special = cfg.vocab - 8
    tr = LongFormTranscriber(model, fe, batch_size=2, chunk_length_s=30.0, max_new_tokens=6,
                             first_special_id=first_special, use_graphs=False)
    assert (tr.chunk_len, tr.stride_left, tr.stride_right) == (480000, 80000, 80000)
    got = tr(audios)
    want = []
    for a in audios:
        seqs = []
        for start, length, _, _, _ in chunk_spans(len(a), 480000, 80000, 80000):
            f = fe(a[start:start + length], sampling_rate=16000, return_tensors="pt").input_features
            ids = _seq(model, f, max_new_tokens=6, use_cache=False)[0, 1:].tolist()
            text = [t for t in ids if t < first_special]
            if text:
                seqs.append(text)
        want.append(merge_sequences(seqs))
    assert got == want and len(got) == 3 and all(len(x) > 0 for x in got)


def test_assisted_greedy_decoding_equals_target_greedy():
    """Speculative decoding (run_eval.py:578-599): student drafts, teacher verifies -> exactly the teacher's greedy
    output, for a good assistant (the teacher itself: everything accepted) and a poor one (unrelated weights)."""
    cfg_t = wo.CONFIGS["micro"]
    t_sd = wo.init_state_dict(cfg_t, 11)
    s_sd, cfg_s = wo.student_from_teacher(t_sd, cfg_t, 2, 1)
    ops = RefOps("cpu", lowp=torch.float32)
    teacher = WhisperForConditionalGeneration(cfg_t, ops=ops, state_dict=t_sd)
    student = WhisperForConditionalGeneration(cfg_s, ops=ops, state_dict=s_sd)
    other = WhisperForConditionalGeneration(cfg_s, ops=ops, state_dict=wo.student_from_teacher(
        wo.init_state_dict(cfg_t, 12), cfg_t, 2, 1)[0])
    feats = torch.randn(2, cfg_t.n_mels, 3000, generator=torch.Generator().manual_seed(2)) * 0.5
    ref = _seq(teacher, feats, max_new_tokens=9, use_cache=False)
    for assistant, k in ((student, 4), (other, 3), (teacher, 5)):
        out = _seq(teacher, feats, max_new_tokens=9, assistant_model=assistant, num_assistant_tokens=k)
        assert torch.equal(out, ref)
        assert teacher.last_accepted <= teacher.last_drafted
    assert teacher.last_accepted == teacher.last_drafted        # the teacher as its own assistant: all drafts accepted
    eos = int(ref[0, 4])
    ref_e = _seq(teacher, feats, max_new_tokens=9, use_cache=False, eos_token_id=eos)
    out_e = _seq(teacher, feats, max_new_tokens=9, assistant_model=student, num_assistant_tokens=4, eos_token_id=eos)
    n = min(ref_e.shape[1], out_e.shape[1])
    assert torch.equal(out_e[:, :n], ref_e[:, :n])
    assert bool((out_e[:, n:] == eos).all()) and bool((ref_e[:, n:] == eos).all())


def test_timestamp_rules_match_transformers_processor():
    lp = pytest.importorskip("transformers.generation.logits_process")
    import types
    from distil_whisper_amd.decoding import apply_timestamp_rules
    V, no_ts, eos, begin = 120, 60, 50, 3
    tb = no_ts + 1
    cfg = types.SimpleNamespace(no_timestamps_token_id=no_ts, eos_token_id=eos, bos_token_id=eos,
                                max_initial_timestamp_index=7, _detect_timestamp_from_logprob=True)
    proc = lp.WhisperTimeStampLogitsProcessor(cfg, begin_index=begin)
    g = torch.Generator().manual_seed(0)
    for trial in range(200):
        B = 5
        n = begin + int(torch.randint(0, 9, (1,), generator=g))
        ids = torch.randint(0, eos, (B, n), generator=g)
        # sprinkle timestamps (in non-decreasing order, as the rules themselves would produce) and some pairs
        for b in range(B):
            cur = tb
            for j in range(begin, n):
                u = float(torch.rand(1, generator=g))
                if u < 0.45:
                    cur = min(V - 1, cur + int(torch.randint(0, 4, (1,), generator=g)))
                    ids[b, j] = cur
        scores = torch.randn(B, V, generator=g) * (3.0 if trial % 2 else 0.3)
        if trial % 3 == 0:
            scores[:, tb:] += 2.0                 # makes the "timestamps more probable than text" branch fire
        want = proc(ids, scores)
        buf = torch.zeros(B, n + 4, dtype=torch.long)
        buf[:, :n] = ids
        got = apply_timestamp_rules(scores, buf, n, begin, no_ts, eos, 7)
        assert torch.equal(torch.isinf(got), torch.isinf(want)), trial
        assert torch.equal(got[~torch.isinf(got)], want[~torch.isinf(want)])


def test_greedy_decoder_with_timestamp_rules_follows_them():
    cfg, model, fe = _model()
    V = cfg.vocab
    no_ts, eos = V - 40, V - 60          # toy layout: text < eos < specials < <|notimestamps|> < timestamps
    feats = torch.randn(2, cfg.n_mels, 3000, generator=torch.Generator().manual_seed(8)) * 0.5
    enc, _ = model.engine.encode(feats, save=False)
    prompt = torch.tensor([[cfg.decoder_start_token_id, 7]] * 2)
    dec = GreedyDecoder(model.engine, 2, 14, eos_token_id=eos, use_graphs=False, check_every=1,
                        timestamp_rules=dict(begin_index=2, no_timestamps_token_id=no_ts, max_initial_timestamp_index=5))
    out = dec.run(enc, prompt, 12)
    gen = out[:, 2:]
    tb = no_ts + 1
    assert bool((gen[:, 0] >= tb).all()) and bool((gen[:, 0] <= tb + 5).all())      # starts with a timestamp
    assert not bool((gen == no_ts).any())
    for row in gen.tolist():
        ts = [t for t in row if t >= tb]
        assert ts == sorted(ts)                                                    # never decreasing


def test_pack_plan_follows_reference_concatenation_rule():
    from distil_whisper_amd.pseudo_label import pack_plan, shard
    rng = np.random.default_rng(9)
    for _ in range(50):
        n = int(rng.integers(1, 40))
        lengths = rng.integers(1, 300, size=n).tolist()
        spk = sorted(rng.integers(0, 4, size=n).tolist()) if rng.random() < 0.7 else None
        packs, cond = pack_plan(lengths, spk, 480)
        # straight restatement of run_pseudo_labelling.py:649-663 on lists of lengths
        cat_len, cat_spk, cat_idx, cprev = [lengths[0]], [spk[0] if spk else None], [[0]], [0]
        for i in range(1, n):
            s = spk[i] if spk else None
            same = s == cat_spk[-1]
            if same and lengths[i] + cat_len[-1] <= 480:
                cat_len[-1] += lengths[i]; cat_idx[-1].append(i)
            else:
                cat_len.append(lengths[i]); cat_spk.append(s); cat_idx.append([i]); cprev.append(1 if same else 0)
        assert packs == cat_idx and cond == cprev
        assert sorted(sum(packs, [])) == list(range(n)) and all(sum(lengths[i] for i in p) <= 480 or len(p) == 1 for p in packs)
    assert pack_plan([], None) == ([], [])
    assert [shard(list(range(10)), r, 4) for r in range(4)] == [[0, 1, 2], [3, 4, 5], [6, 7, 8], [9]]


def test_pseudo_labeller_equals_generate_on_packed_audio():
    from distil_whisper_amd.pseudo_label import PseudoLabeller
    cfg, model, fe = _model()
    rng = np.random.default_rng(10)
    audios = [0.1 * rng.standard_normal(n).astype(np.float32) for n in (200_000, 150_000, 300_000, 100_000)]
    spk = [0, 0, 0, 1]
    eos = cfg.vocab - 3
    pl = PseudoLabeller(model, fe, batch_size=2, max_new_tokens=6, eos_token_id=eos, use_graphs=False)
    toks, packs, cond = pl(audios, spk)
    assert packs == [[0, 1], [2], [3]] and cond == [0, 1, 0]
    for p, t in zip(packs, toks):
        wave = np.concatenate([audios[i] for i in p])
        f = fe(wave, sampling_rate=16000, return_tensors="pt").input_features
        ref = _seq(model, f, max_new_tokens=6, use_cache=False, eos_token_id=eos)[0, 1:].tolist()
        ref = ref[:ref.index(eos)] if eos in ref else ref
        assert t == ref
    # two ranks cover the packs disjointly
    a = PseudoLabeller(model, fe, batch_size=2, max_new_tokens=6, eos_token_id=eos, use_graphs=False, rank=0, world=2)(audios, spk)[0]
    b = PseudoLabeller(model, fe, batch_size=2, max_new_tokens=6, eos_token_id=eos, use_graphs=False, rank=1, world=2)(audios, spk)[0]
    assert [x if x is not None else y for x, y in zip(a, b)] == toks and a[2] is None and b[0] is None


def test_pseudo_labeller_with_beam_search_equals_generate():
    """generation_num_beams > 1 of the pseudo-labelling script (run_pseudo_labelling.py:835-843): the pack decoder runs
    decoding.beam_search_decode; same tokens as generate(num_beams=) on the packed audio."""
    from distil_whisper_amd.pseudo_label import PseudoLabeller
    cfg, model, fe = _model()
    rng = np.random.default_rng(12)
    audios = [0.1 * rng.standard_normal(n).astype(np.float32) for n in (200_000, 150_000, 300_000)]
    eos = cfg.vocab - 3
    pl = PseudoLabeller(model, fe, batch_size=2, max_new_tokens=5, eos_token_id=eos, use_graphs=False, num_beams=2)
    toks, packs, _ = pl(audios, [0, 0, 1])
    assert packs == [[0, 1], [2]]
    for p, t in zip(packs, toks):
        wave = np.concatenate([audios[i] for i in p])
        f = fe(wave, sampling_rate=16000, return_tensors="pt").input_features
        ref = _seq(model, f, max_new_tokens=5, num_beams=2, eos_token_id=eos)[0, 1:].tolist()
        ref = ref[:ref.index(eos)] if eos in ref else ref
        assert t == ref


# ---- chunked long-form WITH timestamps (run_eval.py:566-576 -> pipeline(..., return_timestamps=True) -> _decode_asr) ----
class _StubTokenizer:
    """What `_decode_asr` needs of a WhisperTokenizer, over a micro vocabulary: ids >= TS0 are timestamps."""
    EOS, SOT, PREV, NTS = 900, 901, 909, 911
    TS0 = NTS + 1
    all_special_ids = list(range(900, 912))

    def convert_tokens_to_ids(self, tok):
        return {"<|notimestamps|>": self.NTS, "<|startofprev|>": self.PREV, "<|startoftranscript|>": self.SOT}[tok]

    def _strip_prompt(self, token_ids, prompt_token_id, decoder_start_token_id):
        from transformers.models.whisper.tokenization_whisper import WhisperTokenizer
        return WhisperTokenizer._strip_prompt(self, token_ids, prompt_token_id, decoder_start_token_id)

    def decode(self, ids, **_):
        return "".join(f"<{int(t)}>" for t in ids)


def _reference_segments(windows, time_precision=0.02):
    from transformers.models.whisper.tokenization_whisper import _decode_asr
    tok = _StubTokenizer()
    outs = []
    for w in windows:
        o = {"tokens": np.asarray([w["tokens"]], dtype=np.int64)}
        if w.get("stride") is not None:
            o["stride"] = tuple(w["stride"])
        outs.append(o)
    _, opt = _decode_asr(tok, outs, return_timestamps=True, return_language=False, time_precision=time_precision)
    return [(c["timestamp"], c["text"]) for c in opt["chunks"]]


def test_timestamped_stitching_equals_transformers_decode_asr():
    """`stitch_timestamped` against the reference's `_decode_asr` (imported) on random window sequences: timestamp
    pairs, lone timestamps, repeats, timestamps inside both strides, windows without text, a <|startofprev|> prompt,
    special tokens in between, seek-loop style restarts inside one output."""
    pytest.importorskip("transformers")
    from distil_whisper_amd.longform import stitch_timestamped
    T = _StubTokenizer()
    rng = np.random.default_rng(17)
    nonempty = 0
    for case in range(400):
        n_win = int(rng.integers(1, 6))
        strided = case % 5 != 0
        windows = []
        for wi in range(n_win):
            toks = []
            if rng.random() < 0.15:
                toks += [T.PREV, 5, 6, T.SOT]
            elif rng.random() < 0.5:
                toks += [T.SOT, 902]
            t = int(rng.integers(0, 200))
            for _ in range(int(rng.integers(0, 6))):
                kind = rng.random()
                if kind < 0.75:
                    toks.append(T.TS0 + t)                                   # <|start|> text <|end|>
                    toks += rng.integers(0, 40, size=int(rng.integers(0, 5))).tolist()
                    t = min(1500, t + int(rng.integers(0, 300)))
                    toks.append(T.TS0 + t)
                    if rng.random() < 0.2:
                        toks.append(T.TS0 + t)                               # repeated timestamp
                elif kind < 0.9:
                    toks += rng.integers(0, 40, size=int(rng.integers(1, 4))).tolist()
                else:
                    t = int(rng.integers(0, 100))                            # restart: the seek loop moved on
            toks.append(T.EOS)
            stride = None
            if strided:
                left = 0.0 if wi == 0 else 5.0
                right = 0.0 if wi == n_win - 1 else 5.0
                stride = (30.0 if wi < n_win - 1 else float(rng.integers(11, 31)), left, right)
            windows.append({"tokens": toks, "stride": stride})
        want = _reference_segments(windows)
        got = stitch_timestamped(windows, T.TS0, T.all_special_ids, prompt_token_id=T.PREV, decoder_start_token_id=T.SOT)
        got = [(g["timestamp"], T.decode(g["tokens"])) for g in got]
        assert got == want, (case, windows, got, want)
        nonempty += bool(want)
    assert nonempty > 300


def test_transcriber_with_timestamps_equals_per_window_decode_plus_decode_asr():
    """LongFormTranscriber(return_timestamps=True): windows batched through the graph-capable decoder under the
    timestamp rules, stitched by `stitch_timestamped` == the same windows decoded one at a time, stitched by the
    reference's `_decode_asr`."""
    pytest.importorskip("transformers")
    cfg = wo.OracleConfig(128, 2, 256, 2, 2, 1000, 80, pad_token_id=900, decoder_start_token_id=901)
    T = _StubTokenizer()
    sd = wo.init_state_dict(cfg, 23, std=0.1)
    ops = RefOps("cpu", lowp=torch.float32)
    model = WhisperForConditionalGeneration(cfg, ops=ops, state_dict=sd)
    fe = WhisperFeatureExtractor(feature_size=80, ops=ops)
    rng = np.random.default_rng(6)
    audios = [0.1 * rng.standard_normal(n).astype(np.float32) for n in (1_100_000, 480_000, 700_000)]
    suppress = list(range(902, 912))
    kw = dict(batch_size=2, chunk_length_s=30.0, max_new_tokens=12, prompt_ids=[T.SOT], eos_token_id=T.EOS,
              suppress_tokens=suppress, use_graphs=False)
    tr = LongFormTranscriber(model, fe, return_timestamps=True, no_timestamps_token_id=T.NTS, **kw)
    got = tr(audios)
    assert len(got) == 3
    one = GreedyDecoder(model.engine, 1, 1 + 12, eos_token_id=T.EOS, suppress_tokens=suppress, use_graphs=False,
                        timestamp_rules=dict(begin_index=1, no_timestamps_token_id=T.NTS, max_initial_timestamp_index=50))
    n_ts = 0
    for a, segs in zip(audios, got):
        windows = []
        for start, length, sl, sr, _ in chunk_spans(len(a), 480000, 80000, 80000):
            f = fe(a[start:start + length], sampling_rate=16000, return_tensors="pt").input_features
            enc, _ = model.engine.encode(f, save=False)
            row = one.run(enc, torch.tensor([[T.SOT]]), 12)[0, 1:].tolist()
            n_ts += sum(t >= T.TS0 for t in row)
            windows.append({"tokens": row, "stride": (length / 16000.0, sl / 16000.0, sr / 16000.0)})
        want = _reference_segments(windows)
        assert [(g["timestamp"], T.decode(g["tokens"])) for g in segs] == want
    assert n_ts >= 6          # the timestamp rules were active: the windows carry timestamp tokens
    with pytest.raises(ValueError, match="no_timestamps_token_id"):
        LongFormTranscriber(model, fe, return_timestamps=True, **kw)
