"""Batch assembly of the reference hot loop, producing device-resident inputs for the engine.

`DataCollatorSpeechSeq2SeqWithPadding` mirrors run_distillation.py:405-478 (same constructor fields and the same
returned keys): labels padded to `max_target_length`, `decoder_input_ids = labels[:, :-1]`, `labels = labels[:, 1:]`,
padding -> -100, prompt tokens up to and including <|startoftranscript|> -> -100.  The ragged id lists become one
rectangle by a single numpy scatter into a pinned staging buffer, cross to the device in one copy together with their
lengths, and every mask is derived there (a few KB per batch: not a kernel-worthy hot spot; timed
with the rest of the input side by tools/bench_input_pipeline.py).
`linear_schedule_lr` restates `get_scheduler("linear", ...)` as the reference configures it (1409-1415: warm-up and
total steps are multiplied by the number of processes because every process steps the scheduler).
"""
from dataclasses import dataclass
from typing import Any, Optional, Union

import numpy as np
import torch


@dataclass
class DataCollatorSpeechSeq2SeqWithPadding:
    processor: Any = None
    decoder_start_token_id: int = 50257
    decoder_prev_token_id: int = 50360
    input_padding: Union[bool, str] = "max_length"
    target_padding: Union[bool, str] = "max_length"
    max_target_length: Optional[int] = 448
    pad_token_id: int = 50256
    device: str = "cuda:0"
    # True: the batch also carries "valid_len" = 1 + the last labelled decoder position of the batch, a host integer read
    # off the label lengths before anything goes to the device (DistillationTrainer.train_step(..., valid_len=...) then
    # leaves out the dead positions behind it); "per_sequence": the same per row of the batch (a list of ints: the
    # trainer then also packs the live rows).  Off by default: the reference feeds the batch as `model(**batch)`.
    report_valid_len: Union[bool, str] = False

    def __call__(self, features):
        B = len(features)
        L = self.max_target_length
        # ONE ragged -> rectangular scatter on the host (no per-sample tensor ops), ONE pinned staging buffer, ONE copy to
        # the device carrying the ids and the row lengths; every mask below is derived on the device from the lengths
        lens = np.fromiter((np.asarray(f["labels"]).size for f in features), dtype=np.int64, count=B)
        if B and int(lens.max()) > L:
            raise ValueError(f"labels of length {int(lens.max())} exceed max_target_length={L}")
        stage = torch.empty((B, L + 1), dtype=torch.long, pin_memory=str(self.device).startswith("cuda") and torch.cuda.is_available())
        host = stage.numpy()
        host[:, :L] = self.pad_token_id
        host[:, L] = lens
        if B and int(lens.sum()):
            flat = np.concatenate([np.asarray(f["labels"], dtype=np.int64).reshape(-1) for f in features])
            starts = np.cumsum(lens) - lens
            host[np.repeat(np.arange(B), lens), np.arange(flat.size) - np.repeat(starts, lens)] = flat
        stage = stage.to(self.device, non_blocking=True)
        ids = stage[:, :L]
        att = torch.arange(L, device=stage.device)[None, :] < stage[:, L:]
        decoder_input_ids = ids[:, :-1].contiguous()
        labels = ids[:, 1:].masked_fill(~att[:, 1:], -100)
        bos_index = torch.argmax((labels == self.decoder_start_token_id).long(), dim=1)
        bos_index = torch.where(bos_index > 0, bos_index + 1, bos_index)
        prompt_mask = torch.arange(labels.shape[1], device=labels.device) < bos_index[:, None]
        labels = torch.where(prompt_mask, torch.full_like(labels, -100), labels).contiguous()
        batch = {"labels": labels, "decoder_input_ids": decoder_input_ids}
        if self.report_valid_len:
            per_row = [max(1, int(n) - 1) for n in lens]
            batch["valid_len"] = per_row if self.report_valid_len == "per_sequence" else max(per_row, default=1)
        if "input_features" in features[0]:
            feats = np.stack([np.asarray(f["input_features"], dtype=np.float32) for f in features])
            batch["input_features"] = torch.from_numpy(feats).to(self.device)
        return batch


def linear_schedule_lr(step: int, base_lr: float, warmup_steps: int, total_steps: int, num_processes: int = 1):
    """LR at optimizer step `step` (0-based) under the reference's scheduler set-up: linear warm-up then linear decay,
    the scheduler being stepped `num_processes` times per optimizer step with both horizons scaled accordingly."""
    s, w, t = step * num_processes, warmup_steps * num_processes, total_steps * num_processes
    if s < w:
        return base_lr * s / max(1, w)
    return base_lr * max(0.0, (t - s) / max(1, t - w))
