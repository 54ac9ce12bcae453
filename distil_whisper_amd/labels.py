"""Training-label preparation of the reference hot loop: timestamp filtering and prompt
("condition on previous text") assembly on token-id lists, feeding `collator.DataCollatorSpeechSeq2SeqWithPadding`.

Reference: `prepare_train_dataset` (run_distillation.py:1167-1229) with the constants of 1092-1106:
`timestamp_begin` = <|notimestamps|> (last special id; every id above it is a timestamp), `timestamp_position` = 3 for
multilingual vocabularies else 1, `decoder_prev_token_id` = <|startofprev|>, `prompt_cutoff_length` =
`max_label_length // 2`.  The audio half of that function is the GPU log-mel (`WhisperFeatureExtractor`); this is the
integer half.  The random draws are made in the reference's order from the given generator (one binomial for the
timestamp decision of a sample that has timestamps, then one for the prompt decision of every sample), so a seeded run
reproduces the reference's labels exactly.
"""
import numpy as np


def prepare_train_labels(token_ids_batch, condition_on_prev_batch=None, *, timestamp_begin, timestamp_position,
                         decoder_prev_token_id, timestamp_probability=0.2, condition_on_prev_probability=0.2,
                         max_label_length=448, rng=np.random):
    """token_ids_batch: list of tokenised pseudo-label id lists; condition_on_prev_batch: the data set's
    `condition_on_prev` column (previous-text ids per sample or None entries) or None when the column is absent (the
    prompt is then the previous sample of the batch).  Returns the `labels` lists."""
    prompt_cutoff_length = max_label_length // 2
    has_column = condition_on_prev_batch is not None
    prevs = condition_on_prev_batch if has_column else [None] * len(token_ids_batch)
    out, unprompted = [], []
    for prev_ids, token_ids in zip(prevs, token_ids_batch):
        token_ids = list(token_ids)
        has_timestamps = any(t > timestamp_begin for t in token_ids)
        predict_timestamps = True
        if has_timestamps:
            predict_timestamps = bool(rng.binomial(1, timestamp_probability))
            if not predict_timestamps:
                token_ids = [t for t in token_ids if t < timestamp_begin]
                token_ids.insert(timestamp_position, timestamp_begin)
        unprompted.append(token_ids)
        condition = bool(rng.binomial(1, condition_on_prev_probability))
        if not condition:
            prev_ids = None
        elif not has_column and len(unprompted) > 1:
            prev_ids = unprompted[-2]
        if prev_ids is not None:
            prev_ids = list(prev_ids)
            if has_timestamps and not predict_timestamps:
                prev_ids = [t for t in prev_ids if t < timestamp_begin]
            if len(prev_ids) > prompt_cutoff_length:
                prev_ids = prev_ids[-prompt_cutoff_length + 1:]
            if len(prev_ids) + len(token_ids) + 1 > max_label_length:
                prev_ids = prev_ids[len(token_ids) - max_label_length + 1:]
            token_ids = [decoder_prev_token_id] + prev_ids + token_ids
        out.append(token_ids)
    return out
