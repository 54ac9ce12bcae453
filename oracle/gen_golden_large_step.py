"""TEST INFRASTRUCTURE -- generates tests/golden/large_v3_b7.npz: the losses of the reference `train_step`
(run_distillation.py:1465-1495) at the BENCHMARK's model dimensions (whisper-large-v3-shaped 32/32 teacher ->
distil-large-v3 32/2 student, 128 mel bins, vocabulary 51866) at batch 7 -- the smallest batch at which the product's
padded-row GEMM paths of the bench configuration engage (7 x 447 = 3129 rows -> 3200 = 10 x 320) -- computed by the
`transformers` classes on CPU, once in fp32 and once the way the reference trains (student under bf16 autocast,
teacher loaded in bf16: SURVEY.md 8a').  Forward only: the GPU test compares ce / kl / loss (north-star tolerance
1e-3 relative) and slices of both logit tensors with the bench's exact trainer flags switched on together.

Run in the build container (about 10 minutes on 8 cores):  python oracle/gen_golden_large_step.py
"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import whisper_oracle as wo  # noqa: E402
from oracle.gen_golden import hf_model  # noqa: E402
from oracle.reference_loop import kl_divergence  # noqa: E402

SEED, B = 31, 7


def inputs():
    cfg_t = wo.CONFIGS["large-v3"]
    b = wo.synthetic_batch(cfg_t, B, seed=SEED + 1, with_audio=False)
    feats = torch.randn(B, cfg_t.n_mels, 3000, generator=torch.Generator().manual_seed(SEED + 2)) * 0.5
    return cfg_t, {"input_features": feats, "decoder_input_ids": b["decoder_input_ids"], "labels": b["labels"]}


def main():
    torch.set_num_threads(os.cpu_count())
    cfg_t, batch = inputs()
    t_sd = wo.init_state_dict(cfg_t, SEED)
    s_sd, cfg_s = wo.student_from_teacher(t_sd, cfg_t, 32, 2)
    out = {"seed": SEED, "B": B}
    for tag, autocast in (("fp32", False), ("bf16", True)):
        t0 = time.time()
        teacher, student = hf_model(cfg_t, t_sd).eval(), hf_model(cfg_s, s_sd).train()
        if autocast:
            teacher = teacher.to(torch.bfloat16)
        ctx = torch.autocast("cpu", dtype=torch.bfloat16) if autocast else torch.autocast("cpu", enabled=False)
        with torch.no_grad(), ctx:
            so = student(**batch)
            to = teacher(**batch)
        s_logits, t_logits = so.logits.float(), to.logits.float()
        ce = so.loss.float()
        kl = kl_divergence(nn.functional.softmax(t_logits / 2.0, dim=-1), nn.functional.log_softmax(s_logits / 2.0, dim=-1),
                           batch["labels"]) * 4.0
        loss = 0.8 * ce + 1.0 * kl
        out.update({f"ce_{tag}": ce.item(), f"kl_{tag}": kl.item(), f"loss_{tag}": loss.item(),
                    f"s_logits_{tag}": s_logits[:, ::41, ::1777].numpy().copy(),
                    f"t_logits_{tag}": t_logits[:, ::41, ::1777].numpy().copy(),
                    f"enc_{tag}": so.encoder_last_hidden_state.float()[:, ::211, ::97].numpy().copy()})
        print(tag, {k: v for k, v in out.items() if isinstance(v, float)}, f"{time.time() - t0:.0f} s", flush=True)
        del teacher, student, so, to
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "large_v3_b7.npz"), **out)


if __name__ == "__main__":
    main()
