"""Decode family vs the reference: token ids of `generate` (greedy, suppress / begin-suppress rules, language / task /
language detection prefixes, timestamp rules, prompt_ids, min/max_new_tokens on given encoder outputs, speculative
decoding), of the chunked long-form scheduler and of the pseudo-labelling packs must be IDENTICAL to what
`transformers.WhisperForConditionalGeneration.generate` (+ `chunk_iter` / `_find_longest_common_sequence`) produced on
the same seeded weights and inputs (tests/golden/decode.json, made by oracle/gen_golden_decode.py).

Integer work: the bar is bit-exact.  The GPU path computes in bf16 and the fixtures come from the fp32 reference, so
the generator only keeps cases whose smallest top-1/top-2 logit margin is far above the bf16 rounding noise (margin
stored per scenario; `meta.min_margin`).  CPU leg: the same host logic over the torch restatement of the kernels in
fp32; GPU leg (`-m gpu`): the HIP kernels, with HIP-graph replay on."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import gen_golden_decode as gd
from oracle import whisper_oracle as wo

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "decode.json")))
SC = {s["name"]: s for s in GOLD["scenarios"]}


def _ops(kind):
    if kind == "hip":
        from distil_whisper_amd.ops_hip import HipOps
        return HipOps("cuda:0")
    from oracle.ref_ops import RefOps
    return RefOps("cpu", lowp=torch.float32)


def _model(ops, cfg, sd, fields, dtype=torch.float32):
    from distil_whisper_amd.generation import GenerationConfig
    from distil_whisper_amd.modeling import WhisperForConditionalGeneration
    m = WhisperForConditionalGeneration(cfg, ops=ops, state_dict=sd, dtype=dtype)
    m.generation_config = GenerationConfig.from_any(fields)
    return m


def _models(ops, seed, fields):
    sd_t = gd.weights(seed)
    sd_s, cfg_s = gd.student(sd_t)
    return _model(ops, gd.CFG_T, sd_t, fields), _model(ops, cfg_s, sd_s, fields)


def _check_short(ops, s, graphs):
    teacher, student = _models(ops, s["seed"], s["generation_config"])
    model = teacher if s["model"] == "teacher" else student
    kw = dict(s["gen_kwargs"])
    if "prompt_ids" in kw:
        kw["prompt_ids"] = torch.tensor(kw["prompt_ids"], device=ops.device)
    B = s["B"]
    if s["use_encoder_outputs"]:
        from distil_whisper_amd.modeling import BaseModelOutput
        g = torch.Generator().manual_seed(s["seed"] + 7)
        enc = torch.randn(B, gd.CFG_T.max_src, gd.CFG_T.d_model, generator=g).to(ops.device)
        args = ()
        kw["encoder_outputs"] = BaseModelOutput(last_hidden_state=enc)
    else:
        f = gd.features(s["seed"] + 1, B)
        if s.get("frames"):                       # shorter inputs: the seek loop pads every window to 30 s itself
            f = f[..., :s["frames"]].contiguous()
        args = (f.to(ops.device),)
    if s["assistant"]:
        kw["assistant_model"] = student
    else:
        kw["use_graphs"] = graphs
    if s["sequences"] is not None:
        seq = model.generate(*args, return_dict_in_generate=True, **kw).sequences
        assert seq.tolist() == s["sequences"], (s["name"], seq.tolist(), s["sequences"])
    plain = model.generate(*args, **kw)
    assert plain.tolist() == s["plain"], (s["name"], "plain return value")
    if s["assistant"]:
        assert model.last_accepted >= 0 and model.last_drafted >= model.last_accepted


def _check_longform(ops, s, graphs):
    from distil_whisper_amd.longform import LongFormTranscriber
    from distil_whisper_amd.modeling import WhisperFeatureExtractor
    _, student = _models(ops, s["seed"], s["generation_config"])
    fe = WhisperFeatureExtractor(feature_size=80, ops=ops)
    audios = [gd.audio(s["seed"] * 10 + i, n) for i, n in enumerate(s["lengths"])]
    gc = s["generation_config"]
    tr = LongFormTranscriber(student, fe, batch_size=s["batch"], chunk_length_s=30.0, max_new_tokens=s["max_new_tokens"],
                             prompt_ids=[gc["decoder_start_token_id"], gc["no_timestamps_token_id"]],
                             eos_token_id=gc["eos_token_id"], first_special_id=gc["eos_token_id"],
                             suppress_tokens=gc["suppress_tokens"], use_graphs=graphs)
    got = tr(audios)
    assert got == s["merged"], (got, s["merged"])


def _check_pseudo_label(ops, s, graphs):
    from distil_whisper_amd.modeling import WhisperFeatureExtractor
    from distil_whisper_amd.pseudo_label import PseudoLabeller, pack_plan
    teacher, _ = _models(ops, s["seed"], s["generation_config"])
    fe = WhisperFeatureExtractor(feature_size=80, ops=ops)
    audios = [gd.audio(s["seed"] * 10 + i, n) for i, n in enumerate(s["lengths"])]
    packs, _ = pack_plan(s["lengths"], s["speakers"], 480000)
    assert packs == s["packs"]
    gc = s["generation_config"]
    prompt = [gc["decoder_start_token_id"], gc["lang_to_id"]["<|en|>"], gc["task_to_id"]["transcribe"]]
    lab = PseudoLabeller(teacher, fe, batch_size=2, max_new_tokens=s["max_new_tokens"], prompt_ids=prompt,
                         eos_token_id=gc["eos_token_id"], suppress_tokens=gc["suppress_tokens"],
                         begin_suppress_tokens=gc["begin_suppress_tokens"],
                         timestamp_rules=dict(no_timestamps_token_id=gc["no_timestamps_token_id"],
                                              max_initial_timestamp_index=gc["max_initial_timestamp_index"]),
                         use_graphs=graphs)
    ids, packs2, _ = lab(audios, s["speakers"])
    assert packs2 == s["packs"]
    # the reference's plain return of generate(..., return_timestamps=True): the tokens of all segments of the seek loop
    assert ids == s["sequences"], (ids, s["sequences"])
    assert max(s["segments_per_pack"]) >= 2


def _run_all(ops, graphs):
    assert GOLD["meta"]["min_margin"] >= 0.05      # bf16 logit noise on these weights: ~0.007 sigma (std)
    for s in GOLD["scenarios"]:
        assert s["margin"] >= GOLD["meta"]["min_margin"], (s["name"], s["margin"])
        if s["kind"] == "short":
            _check_short(ops, s, graphs)
        elif s["kind"] == "longform":
            _check_longform(ops, s, graphs)
        else:
            _check_pseudo_label(ops, s, graphs)


def test_decode_family_matches_reference_fixtures_cpu():
    _run_all(_ops("ref"), graphs=False)


@pytest.mark.gpu
def test_decode_family_matches_reference_fixtures_gpu():
    _run_all(_ops("hip"), graphs=True)


@pytest.mark.gpu
def test_decode_family_eager_equals_graph_replay_gpu():
    ops = _ops("hip")
    for name in ("greedy_suppress_student", "timestamps_single_call"):
        _check_short(ops, SC[name], graphs=False)


@pytest.mark.gpu
@pytest.mark.parametrize("fuse_off", [0, 8, 12])
def test_decode_fixtures_under_every_token_step_fusion_mode_gpu(fuse_off):
    """dw_debug_set key 7: 4 is the default (cross-attention with its q projection inside, self-attention as separate launches);
    0 adds the self-attention kernel with its q / k / v projection and cache append inside (measured slower, kept behind the
    switch), 8 / 12 are the separate-launch forms.  Every mode must reproduce the reference's tokens."""
    ops = _ops("hip")
    assert ops.lib.dw_debug_set(7, fuse_off) == 0
    try:
        for name in ("greedy_suppress_student", "timestamps_single_call"):
            _check_short(ops, SC[name], graphs=True)
        for s in GOLD["scenarios"]:
            if s["kind"] == "longform":
                _check_longform(ops, s, True)
    finally:
        ops.lib.dw_debug_set(7, 4)


def _encoder_outputs_and_shared_assistant(ops):
    """`generate(encoder_outputs=...)` in every accepted layout (run_eval.py:806-844 `benchmark_gen`) and an assistant
    that re-uses the target's encoder output (run_eval.py:578-599), on the margin-selected fixture cases: the same
    reference tokens as the calls that start from input_features."""
    from distil_whisper_amd.modeling import BaseModelOutput
    s = SC["language_task_teacher"]
    teacher, _ = _models(ops, s["seed"], s["generation_config"])
    feats = gd.features(s["seed"] + 1, s["B"]).to(ops.device)
    enc, _ = teacher.engine.encode(feats.float().contiguous(), save=False)
    L, B = gd.CFG_T.max_src, s["B"]
    for eo in (enc[: B * L].clone(), enc[: B * L].float().view(B, L, -1),
               BaseModelOutput(last_hidden_state=enc[: B * L].float().view(B, L, -1)), (enc[: B * L].view(B, L, -1),)):
        got = teacher.generate(encoder_outputs=eo, return_dict_in_generate=True, **s["gen_kwargs"]).sequences
        assert got.tolist() == s["sequences"]
    s = SC["assisted_teacher_student"]
    teacher, student = _models(ops, s["seed"], s["generation_config"])
    feats = gd.features(s["seed"] + 1, 1).to(ops.device)
    enc, _ = teacher.engine.encode(feats.float().contiguous(), save=False)
    student.share_encoder_output = True
    a = teacher.generate(feats, assistant_model=student, return_dict_in_generate=True, **s["gen_kwargs"]).sequences
    b = teacher.generate(encoder_outputs=enc[:L].clone(), assistant_model=student, return_dict_in_generate=True,
                         **s["gen_kwargs"]).sequences
    assert a.tolist() == s["sequences"] and b.tolist() == s["sequences"]


def test_encoder_outputs_layouts_and_shared_encoder_assistant_cpu():
    _encoder_outputs_and_shared_assistant(_ops("ref"))


@pytest.mark.gpu
def test_encoder_outputs_layouts_and_shared_encoder_assistant_gpu():
    _encoder_outputs_and_shared_assistant(_ops("hip"))


def test_generate_matches_transformers_live_and_rejects_unsupported_arguments():
    """Direct comparison with the imported reference class on a fresh seed (no margin selection: fp32 both sides), and
    the loud failures for arguments the engine path does not implement."""
    pytest.importorskip("transformers")
    ops = _ops("ref")
    fields = gd.generation_fields(multilingual=True, suppress=True)
    seed = 7
    sd_t = gd.weights(seed)
    model = _model(ops, gd.CFG_T, sd_t, fields)
    feats = gd.features(seed + 1, 2)
    for kw in (dict(max_new_tokens=7, language="hi", task="transcribe"),
               dict(max_new_tokens=5, language=["en", "de"]),
               dict(max_length=9, language="<|fr|>", prompt_ids=[gd.STARTOFPREV, 700, 701])):
        hkw = dict(kw)
        if "prompt_ids" in hkw:
            hkw["prompt_ids"] = torch.tensor(hkw["prompt_ids"])
        ref, ref_plain, _ = gd.hf_generate(gd.CFG_T, sd_t, fields, feats, want_plain=True, **hkw)
        got = model.generate(feats, return_dict_in_generate=True, **kw).sequences
        assert got.tolist() == ref.tolist(), kw
        assert model.generate(feats, **kw).tolist() == ref_plain.tolist(), kw
    # beam search (TF `_beam_search`), live: finishing on EOS is forced by declaring a token the model likes to emit the
    # EOS, so hypotheses finish at different lengths and the length penalty / early-stopping branches decide
    hf = gd.hf_model(gd.CFG_T, sd_t, **fields)
    with torch.no_grad():
        first = hf.generate(feats, return_dict_in_generate=True, max_new_tokens=4, language="en").sequences
    frequent = int(first[0, -2])
    for kw in (dict(num_beams=2, max_new_tokens=6, language="en"),
               dict(num_beams=4, max_new_tokens=8, language="de", eos_token_id=frequent),
               dict(num_beams=3, max_new_tokens=8, language="en", eos_token_id=frequent, length_penalty=2.0,
                    early_stopping=True),
               dict(num_beams=3, max_new_tokens=7, language="en", eos_token_id=frequent, length_penalty=0.5,
                    early_stopping="never")):
        with torch.no_grad():
            ref = gd.hf_model(gd.CFG_T, sd_t, **fields).generate(feats, return_dict_in_generate=True, **kw).sequences
            ref_plain = gd.hf_model(gd.CFG_T, sd_t, **fields).generate(feats, **kw)
        got = model.generate(feats, return_dict_in_generate=True, **kw).sequences
        assert got.tolist() == ref.tolist(), (kw, got.tolist(), ref.tolist())
        assert model.generate(feats, **kw).tolist() == ref_plain.tolist(), kw
    # GenerationMixin's history-dependent processors and plain sampling on the single-window KV-cache decoder (round 6):
    # repetition_penalty / no_repeat_ngram_size under greedy search are deterministic -> token for token; sampling draws
    # from torch's global generator exactly as `_sample` does (one multinomial per step over the warped fp32 scores), so
    # on the same device under the same seed the sampled tokens are the reference's too.  A positive `temperature` IS the
    # sampling switch in the reference's Whisper generate (`do_sample = temperature is not None and temperature > 0.0`):
    # `do_sample=True` alone decodes greedily there, and here.
    for kw in (dict(max_new_tokens=9, language="en", repetition_penalty=1.8),
               dict(max_new_tokens=9, language="en", no_repeat_ngram_size=2, eos_token_id=frequent),
               dict(max_new_tokens=8, language="de", repetition_penalty=1.3, no_repeat_ngram_size=3),
               dict(max_new_tokens=6, language="en", do_sample=True)):
        with torch.no_grad():
            ref = gd.hf_model(gd.CFG_T, sd_t, **fields).generate(feats, return_dict_in_generate=True, **kw).sequences
        got = model.generate(feats, return_dict_in_generate=True, **kw).sequences
        assert got.tolist() == ref.tolist(), (kw, got.tolist(), ref.tolist())
    for kw in (dict(max_new_tokens=8, language="en", temperature=0.8),
               dict(max_new_tokens=8, language="en", temperature=1.3, top_k=5),
               dict(max_new_tokens=8, language="fr", temperature=0.7, top_p=0.6, repetition_penalty=1.2),
               dict(max_new_tokens=7, language="en", temperature=1.0, top_k=20, top_p=0.9, no_repeat_ngram_size=2)):
        hf_s = gd.hf_model(gd.CFG_T, sd_t, **fields)       # (built BEFORE seeding: its random initialisation draws from the generator)
        torch.manual_seed(1234)
        with torch.no_grad():
            ref = hf_s.generate(feats, return_dict_in_generate=True, **kw).sequences
        torch.manual_seed(1234)
        got = model.generate(feats, return_dict_in_generate=True, **kw).sequences
        assert got.tolist() == ref.tolist(), (kw, got.tolist(), ref.tolist())
        torch.manual_seed(99)
        other = model.generate(feats, return_dict_in_generate=True, **kw).sequences
        assert other.shape == got.shape
    for bad, exc in ((dict(num_beams=4, num_beam_groups=2), NotImplementedError),
                     (dict(num_beams=2, repetition_penalty=1.2), NotImplementedError),
                     (dict(temperature=0.5, return_timestamps=True), NotImplementedError),
                     (dict(temperature=(0.2, 0.4)), NotImplementedError),
                     (dict(condition_on_prev_tokens=True), NotImplementedError),
                     (dict(no_speech_threshold=0.6, logprob_threshold=-1.0), NotImplementedError),
                     (dict(return_token_timestamps=True), NotImplementedError),
                     (dict(languge="en"), ValueError), (dict(language="klingon"), ValueError),
                     (dict(task="summarize", language="en"), ValueError),
                     (dict(max_new_tokens=500), ValueError)):
        with pytest.raises(exc):
            model.generate(feats, **bad)
    with pytest.raises(ValueError, match="more than 3000 mel input features"):
        model.generate(torch.zeros(1, 80, 6000))
    with pytest.raises(ValueError, match="attention_mask"):
        model.generate(torch.zeros(2, 80, 6000), return_timestamps=True, language="en")
    en = _model(ops, gd.CFG_T, sd_t, gd.generation_fields(multilingual=False))
    with pytest.raises(ValueError, match="English-only"):
        en.generate(feats, language="en")


def test_assistant_model_in_the_timestamp_seek_loop():
    """run_eval.py:578-599, 706-707 with long-form inputs: `assistant_model` inside the seek loop.  Speculative greedy
    decoding emits the target's own greedy tokens -- that is its contract -- so the result must equal the plain seek loop
    with the begin-suppress rule off (TF:719-721 drops it when an assistant is given), whether the assistant shares the
    encoder or not, for one long utterance and for a batch.  (Not compared with the imported class here: transformers
    5.15's assisted path with `return_timestamps=True` does not reproduce its own greedy output on these inputs -- it
    emits decreasing timestamp pairs, the timestamp rules being applied to candidate positions with the wrong history --
    so there is no reference behaviour to match beyond the contract.)"""
    ops = _ops("ref")
    fields = gd.generation_fields(multilingual=True, suppress=True, timestamps=True)
    plain_fields = dict(fields, begin_suppress_tokens=None)
    for seed in (310,):
        teacher, student = _models(ops, seed, fields)
        plain, _ = _models(ops, seed, plain_fields)
        long1 = torch.cat([gd.features(seed + 1, 1), gd.features(seed + 2, 1)[..., :2200]], -1)
        batch = torch.cat([gd.features(seed + 3, 1), gd.features(seed + 4, 1)], 0)
        for shared in (False, True):
            student.share_encoder_output = shared
            if shared:                      # a student that kept the teacher's encoder
                student = _model(ops, student.dims, {**student.state_dict(), **{k: v for k, v in teacher.state_dict().items()
                                                                                   if k.startswith("model.encoder.")}}, fields)
                student.share_encoder_output = True
            for feats, kw in ((long1, dict(max_new_tokens=6, return_timestamps=True, language="en")),
                              (batch, dict(max_new_tokens=5, return_timestamps=True, language="de"))):
                want = plain.generate(feats, **kw)
                got = teacher.generate(feats, assistant_model=student, **kw)
                assert got.tolist() == want.tolist(), (seed, shared, kw)
                assert teacher.last_drafted > 0 and 0 <= teacher.last_accepted <= teacher.last_drafted
        # one window, one call (force_unique_generate_call): the same contract
        one = gd.features(seed + 5, 2)
        kw1 = dict(max_new_tokens=7, return_timestamps=True, language="en", force_unique_generate_call=True)
        assert teacher.generate(one, assistant_model=student, **kw1).tolist() == plain.generate(one, **kw1).tolist()


def test_timestamp_seek_loop_matches_transformers_live():
    """generate(..., return_timestamps=True) is a seek loop in the reference (TF:generation_whisper.py:784-903): every
    window is decoded with the timestamp rules, `_retrieve_segment` splits it at consecutive timestamp pairs and moves
    the utterance to the last predicted end of segment; inputs longer than 30 s run the same loop (batches need the
    attention mask).  Compared live with the imported class on fresh seeds (fp32 both sides): a 30 s batch, a 75 s
    utterance, and a batch of a 60 s and a 20 s utterance."""
    pytest.importorskip("transformers")
    from distil_whisper_amd.generation import retrieve_segment
    ops = _ops("ref")
    fields = gd.generation_fields(multilingual=True, suppress=True, timestamps=True)
    multi = 0
    for seed in (300, 301):
        sd_t = gd.weights(seed)
        model = _model(ops, gd.CFG_T, sd_t, fields)
        long1 = torch.cat([gd.features(seed + 1, 1), gd.features(seed + 2, 1), gd.features(seed + 3, 1)[..., :1500]], -1)
        a = torch.cat([gd.features(seed + 4, 1), gd.features(seed + 5, 1)], -1)
        b = torch.cat([gd.features(seed + 6, 1)[..., :2000], torch.zeros(1, 80, 4000)], -1)
        mask = torch.ones(2, 6000, dtype=torch.long)
        mask[1, 2000:] = 0
        for feats, kw in ((gd.features(seed + 1, 2), dict(max_new_tokens=6, return_timestamps=True, language="en")),
                          (long1, dict(max_new_tokens=5, return_timestamps=True, language="de")),
                          (torch.cat([a, b], 0), dict(max_new_tokens=4, return_timestamps=True, language="en",
                                                      attention_mask=mask))):
            with torch.no_grad():
                ref = gd.hf_model(gd.CFG_T, sd_t, **fields).generate(feats, **kw)
            out = model.generate(feats, return_dict_in_generate=True, **kw)
            assert out.sequences.tolist() == ref.tolist(), (seed, kw)
            multi += max(len(s) for s in out.segments) > 1
            if seed != 300:          # (the variants below once: they multiply the reference's CPU time)
                continue
            # return_segments=True: the reference's {"sequences", "segments"} with the same start / end / tokens per segment
            with torch.no_grad():
                rs = gd.hf_model(gd.CFG_T, sd_t, **fields).generate(feats, return_segments=True, **kw)
            mine = model.generate(feats, return_segments=True, **kw)
            assert isinstance(mine, dict) and mine["sequences"].tolist() == rs["sequences"].tolist()
            assert len(mine["segments"]) == len(rs["segments"])
            for got_row, ref_row in zip(mine["segments"], rs["segments"]):
                assert len(got_row) == len(ref_row)
                for gs, hs in zip(got_row, ref_row):
                    assert list(gs["tokens"]) == hs["tokens"].tolist()
                    assert abs(float(gs["start"]) - float(hs["start"])) < 1e-6 and abs(float(gs["end"]) - float(hs["end"])) < 1e-6
            # prompt_ids (run_eval.py:709-710 passes them to long-form generate too): in front of every window's decoder
            # prompt without conditioning; segment zero of the utterance with condition_on_prev_tokens
            pid = torch.tensor([fields["prev_sot_token_id"], 31, 32, 33])
            for extra in ({}, {"condition_on_prev_tokens": True},
                          {"condition_on_prev_tokens": True, "prompt_condition_type": "all-segments"}):
                with torch.no_grad():
                    rp = gd.hf_model(gd.CFG_T, sd_t, **fields).generate(feats, prompt_ids=pid, return_segments=True,
                                                                         **kw, **extra)
                mp = model.generate(feats, prompt_ids=pid, return_segments=True, **kw, **extra)
                assert mp["sequences"].tolist() == rp["sequences"].tolist(), (seed, kw, extra)
                assert [[list(sg["tokens"]) for sg in row] for row in mp["segments"]] == \
                       [[sg["tokens"].tolist() for sg in row] for row in rp["segments"]]
            # beam search inside the loop (run_eval.py:693 `--generation_num_beams`)
            with torch.no_grad():
                rb = gd.hf_model(gd.CFG_T, sd_t, **fields).generate(feats, num_beams=2, **kw)
            assert model.generate(feats, num_beams=2, **kw).tolist() == rb.tolist(), (seed, kw, "beams")
    with pytest.raises(ValueError, match="condition_on_prev_tokens=True"):
        model.generate(feats, prompt_ids=pid, prompt_condition_type="all-segments", **kw)
    assert multi >= 4                                  # the loop really ran several passes
    # the segment rule on its own (TF `_retrieve_segment`): pairs, single ending, no timestamps, empty
    tb = 100
    segs, off = retrieve_segment([100, 5, 6, 110, 110, 7, 120], tb, 3000)
    assert [s["tokens"] for s in segs] == [[100, 5, 6, 110], [110, 7, 120]] and off == 3000
    segs, off = retrieve_segment([100, 5, 110, 110, 7, 8], tb, 3000)
    assert [s["tokens"] for s in segs] == [[100, 5, 110, 110]] and off == 10 * 2
    segs, off = retrieve_segment([5, 6, 7], tb, 1234)
    assert [s["tokens"] for s in segs] == [[5, 6, 7]] and off == 1234
    segs, off = retrieve_segment([], tb, 77)
    assert [s["tokens"] for s in segs] == [[]] and off == 77


def test_seek_loop_conditioning_and_fallback_thresholds_match_transformers_live():
    """The long-form heuristics of the reference's seek loop, live against the imported class (fp32 both sides):
    `condition_on_prev_tokens` (earlier segments behind <|startofprev|> in front of the prompt; rows of a batch end up
    with different prompt lengths), and at temperature 0 the decisions of `_need_fallback`: zlib compression ratio of
    the token bytes, average log-probability of the chosen tokens, and the no-speech skip (P(<|nospeech|>) after
    <|startoftranscript|> with a low average log-probability drops the window).  A real temperature fallback samples
    from this process's random stream: it is only checked to run and to be reproducible under a fixed torch seed."""
    pytest.importorskip("transformers")
    ops = _ops("ref")
    fields = gd.generation_fields(multilingual=True, suppress=True, timestamps=True)
    fields["prev_sot_token_id"] = gd.STARTOFPREV
    changed = 0
    for seed in (400, 401):
        sd_t = gd.weights(seed)
        model = _model(ops, gd.CFG_T, sd_t, fields)
        f2 = gd.features(seed + 1, 2)[..., :700].contiguous()
        long1 = torch.cat([gd.features(seed + 2, 1), gd.features(seed + 3, 1)[..., :1000]], -1)
        base = dict(max_new_tokens=6, return_timestamps=True, language="en")
        plain = model.generate(f2, **base).tolist()
        for feats, kw in ((f2, dict(base, condition_on_prev_tokens=True)),
                          (long1, dict(max_new_tokens=5, return_timestamps=True, language="de",
                                       condition_on_prev_tokens=True)),
                          (f2, dict(base, temperature=0.0, logprob_threshold=-6.0, compression_ratio_threshold=1.2,
                                    no_speech_threshold=0.0005)),
                          (long1, dict(max_new_tokens=5, return_timestamps=True, language="en", temperature=(0.0,),
                                       logprob_threshold=-6.5, no_speech_threshold=0.001,
                                       condition_on_prev_tokens=True))):
            with torch.no_grad():
                ref = gd.hf_model(gd.CFG_T, sd_t, **fields).generate(feats, **kw)
            got = model.generate(feats, **kw)
            assert got.tolist() == ref.tolist(), (seed, kw)
            changed += feats is f2 and got.tolist() != plain
        torch.manual_seed(5)
        a = model.generate(f2, **dict(base, temperature=(0.0, 0.4, 0.8), compression_ratio_threshold=0.5, logprob_threshold=-1.0))
        torch.manual_seed(5)
        b = model.generate(f2, **dict(base, temperature=(0.0, 0.4, 0.8), compression_ratio_threshold=0.5, logprob_threshold=-1.0))
        assert a.tolist() == b.tolist() and a.shape[0] == 2
    assert changed >= 1                                # conditioning / the skip rule really changed what was decoded
    with pytest.raises(NotImplementedError, match="seek loop"):
        model.generate(gd.features(1, 1), language="en", logprob_threshold=-1.0)


@pytest.mark.gpu
def test_seek_loop_heuristics_run_on_the_hip_path():
    """The long-form heuristics on the HIP engine (scoring pass through engine.decode, sampling fallback through the
    cached decoder passes): same call as the live CPU comparison; a bf16 GPU run is not required to take every
    near-threshold decision like the fp32 reference, so this checks shapes, token ranges and determinism."""
    ops = _ops("hip")
    fields = gd.generation_fields(multilingual=True, suppress=True, timestamps=True)
    fields["prev_sot_token_id"] = gd.STARTOFPREV
    sd_t = gd.weights(400)
    model = _model(ops, gd.CFG_T, sd_t, fields)
    f2 = gd.features(401, 2)[..., :700].contiguous().cuda()
    kw = dict(max_new_tokens=6, return_timestamps=True, language="en", condition_on_prev_tokens=True, temperature=(0.0,),
              logprob_threshold=-6.0, compression_ratio_threshold=1.2, no_speech_threshold=0.0005)
    a, b = model.generate(f2, **kw), model.generate(f2, **kw)
    assert a.tolist() == b.tolist() and a.shape[0] == 2 and int(a.max()) < gd.V
    torch.manual_seed(3)
    c = model.generate(f2, max_new_tokens=6, return_timestamps=True, language="en", temperature=(0.0, 0.5),
                       compression_ratio_threshold=0.5, logprob_threshold=-1.0)
    assert c.shape[0] == 2 and int(c.max()) < gd.V
