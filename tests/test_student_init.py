"""distil_whisper_amd.student_init.student_from_teacher (the function bench.py builds its student with) against the
reference's own `init_student_model_from_teacher` (create_student_model.py:92-216), exec'd from /root/reference on a
micro checkpoint directory with the `transformers` model class and stub processor / generation-config classes (there
is no tokenizer on disk); plus the known layer maps of the BASELINE configurations."""
import os

import numpy as np
import pytest
import torch

from distil_whisper_amd import student_init as si
from distil_whisper_amd.engine import WhisperDims
from oracle import gen_golden_decode as gd
from oracle import whisper_oracle as wo

REF = "/root/reference/training/create_student_model.py"


def _reference_student(teacher_dir, save_dir, **kw):
    import copy
    import logging
    import transformers
    src = open(REF).read()
    a = src.index("def init_student_model_from_teacher(")
    b = src.index('if __name__ == "__main__":')

    class Processor:
        @classmethod
        def from_pretrained(cls, *_a, **_k):
            return cls()

        def save_pretrained(self, *_a, **_k):
            pass

        def __call__(self, audio, sampling_rate=None, return_tensors=None):
            return type("F", (), {"input_features": torch.zeros(1, 80, 3000)})()

    class GenCfg(Processor):
        forced_decoder_ids = None

    ns = {"WhisperForConditionalGeneration": transformers.WhisperForConditionalGeneration, "WhisperProcessor": Processor,
          "GenerationConfig": GenCfg, "copy": copy, "np": np, "torch": torch, "logger": logging.getLogger("ref")}
    exec(src[a:b], ns)
    ns["init_student_model_from_teacher"](teacher_dir, save_dir=save_dir, **kw)
    return transformers.WhisperForConditionalGeneration.from_pretrained(save_dir).state_dict()


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("kw", [dict(encoder_layers=None, decoder_layers=2),
                                dict(encoder_layers=2, decoder_layers=3),
                                dict(encoder_layers=3, decoder_layers=1),
                                dict(encoder_layers=None, decoder_layers=2, decoder_layers_numbers=[1, 2]),
                                dict(encoder_layers=4, decoder_layers=2, decoder_layers_numbers=[3, 3]),
                                dict(encoder_layers=None, decoder_layers=5)])
def test_matches_reference_init_student_model_from_teacher(tmp_path, kw):
    pytest.importorskip("transformers")
    from transformers import WhisperForConditionalGeneration
    cfg = wo.OracleConfig(64, 1, 128, 4, 5, 300, 80, pad_token_id=0, decoder_start_token_id=1)
    t_sd = wo.init_state_dict(cfg, 3)
    hf = WhisperForConditionalGeneration(gd.hf_config(cfg))
    full = dict(t_sd)
    full["proj_out.weight"] = t_sd["model.decoder.embed_tokens.weight"]
    hf.load_state_dict(full, strict=False)
    hf.save_pretrained(str(tmp_path / "teacher"))
    want = _reference_student(str(tmp_path / "teacher"), str(tmp_path / "student"), **kw)
    got, sdims = si.student_from_teacher(t_sd, WhisperDims.from_any(cfg), kw["encoder_layers"], kw["decoder_layers"],
                                         kw.get("decoder_layers_numbers"))
    assert sdims.enc_layers == (kw["encoder_layers"] or 4) and sdims.dec_layers == kw["decoder_layers"]
    want = {k: v for k, v in want.items() if k != "proj_out.weight"}
    assert sorted(got) == sorted(want)
    for k in want:
        assert torch.equal(got[k], want[k]), k


def test_layer_maps_of_the_baseline_configs_and_errors():
    assert si.student_layer_map(32, 2) == [0, 31]                 # distil-large-v3: first and last teacher layer
    assert si.student_layer_map(12, 4) == [0, 3, 7, 11]           # distil-small.en
    assert si.student_layer_map(4, 1) == [3]                      # tiny 4/1: the single layer is the teacher's LAST
    assert si.student_layer_map(32, 32) == list(range(32))
    d = WhisperDims(64, 1, 128, 2, 3, 50, 80)
    sd = wo.init_state_dict(wo.OracleConfig(64, 1, 128, 2, 3, 50, 80), 1)
    with pytest.raises(ValueError, match="layers number"):
        si.student_from_teacher(sd, d, None, 2, decoder_layers_numbers=[0])
    s, sd_dims = si.student_from_teacher(sd, d, None, 1)
    assert torch.equal(s["model.decoder.layers.0.fc1.weight"], sd["model.decoder.layers.2.fc1.weight"])
    assert "model.decoder.layers.1.fc1.weight" not in s and "proj_out.weight" not in s
    # the oracle's twin (used by the parity tests) agrees with the product function
    o_sd, o_cfg = wo.student_from_teacher(sd, wo.OracleConfig(64, 1, 128, 2, 3, 50, 80), 2, 2)
    p_sd, _ = si.student_from_teacher(sd, d, 2, 2)
    assert sorted(o_sd) == sorted(p_sd) and all(torch.equal(o_sd[k], p_sd[k]) for k in o_sd)
