"""Chunked long-form transcription scheduler (BASELINE config 5).

Reference behaviour: run_eval.py:566-576, 726-786 drives the `transformers` ASR pipeline with `chunk_length_s`:
  * the audio is cut by `chunk_iter` (TF:pipelines/automatic_speech_recognition.py:61-84) into windows of
    `chunk_len` samples that advance by `chunk_len - stride_left - stride_right` (stride = chunk/6 by default,
    TF:...automatic_speech_recognition.py `preprocess`), every window is padded to 30 s by the feature extractor;
  * the windows of all utterances are batched through encoder + greedy generate;
  * per utterance, the text tokens of consecutive windows are stitched by the sliding longest-common-sequence rule of
    `_find_longest_common_sequence` (TF:models/whisper/tokenization_whisper.py, called from `_decode_asr` when no
    timestamps are requested).
  * with `return_timestamps=True` (run_eval.py:566-576 passes it through to the pipeline) the windows are decoded under
    the timestamp rules and `_decode_asr` stitches them by their timestamp tokens instead: a state machine over the
    windows that places every `<|t|>` on the utterance's time axis (window offset minus left stride), opens a segment at
    the first usable timestamp and closes it at the next, ignores the timestamps that fall into a stride (the pair that
    straddles the right stride is resolved by the NEXT window) and merges the text tokens a segment collected from
    several windows with the same longest-common-sequence rule (`stitch_timestamped` below).
Here the windows are gathered on the GPU straight into the [B, 480000] buffer of the log-mel kernel, encoded and
decoded by decoding.GreedyDecoder (HIP-graph replay of the token steps); chunking and stitching are integer host
logic restated below and pinned against the `transformers` functions in tests/test_longform.py.
"""
import numpy as np
import torch

from .decoding import GreedyDecoder


def chunk_spans(n_samples, chunk_len, stride_left, stride_right):
    """[(start, length, stride_left, stride_right, is_last)] of the windows `chunk_iter` yields for an input of
    n_samples: the first window has no left stride, the last no right stride, a window that only holds left-stride
    samples is dropped, and iteration ends at the first window that reaches the end of the input."""
    step = chunk_len - stride_left - stride_right
    if step <= 0:
        raise ValueError("Chunk length must be superior to stride length")
    spans = []
    for start in range(0, n_samples, step):
        end = start + chunk_len
        length = min(end, n_samples) - start
        sl = 0 if start == 0 else stride_left
        last = end >= n_samples
        sr = 0 if last else stride_right
        if length > sl:
            spans.append((start, length, sl, sr, last))
        if last:
            break
    return spans


def merge_sequences(sequences):
    """Stitch the token lists of consecutive overlapping windows.  For each neighbouring pair every alignment
    i = 1 .. len(left)+len(right)-1 (right slid i tokens into the tail of left) is scored matches/i + i/10000 over the
    overlapped positions; the first best alignment with more than one match wins, the overlap is cut at its middle
    (left half from the left window, right half from the right one); without such an alignment the lists are
    concatenated.  Match counts of all alignments are the diagonal sums of the equality matrix."""
    if len(sequences) == 0:
        return []
    left = list(sequences[0])
    total = []
    for right in sequences[1:]:
        right = list(right)
        L, R = len(left), len(right)
        best = (L, L, 0, 0)
        if L > 0 and R > 0:
            a, b = np.nonzero(np.asarray(left)[:, None] == np.asarray(right)[None, :])
            counts = np.bincount(b - a + L, minlength=L + R)[1:L + R]          # index i-1 -> matches of alignment i
            i = np.arange(1, L + R)
            score = np.where(counts > 1, counts / i + i / 10000.0, 0.0)
            k = int(np.argmax(score))
            if score[k] > 0.0:
                ii = k + 1
                best = (max(0, L - ii), min(L, L + R - ii), max(0, ii - L), min(R, ii))
        left_mid = (best[1] + best[0]) // 2
        right_mid = (best[3] + best[2]) // 2
        total.extend(left[:left_mid])
        left = right[right_mid:]
    total.extend(left)
    return total


def stitch_timestamped(windows, timestamp_begin, special_ids=(), prompt_token_id=None, decoder_start_token_id=None,
                       time_precision=0.02, segment_size=1500):
    """Timestamp branch of the reference's chunk stitching (`_decode_asr`, TF:models/whisper/tokenization_whisper.py, as
    the ASR pipeline calls it for Whisper with `return_timestamps=True`), on token ids only.

    windows: [{"tokens": generated ids of one window (a leading <|startofprev|> prompt is dropped up to
    `decoder_start_token_id`), "stride": (window seconds, left stride seconds, right stride seconds) or None}] in time
    order.  Returns [{"timestamp": (start, end), "tokens": [...]}]; `end` is None for a trailing segment whose closing
    timestamp was never produced.  Rules, per window:
      * a timestamp token t stands for (t - timestamp_begin) * time_precision seconds after the window's first sample;
        windows overlap by their strides, so the window's origin on the utterance axis is the running offset minus its
        left stride; within one `generate` output that itself ran the seek loop, a timestamp smaller than its
        predecessor means the next 30 s segment started (`segment_size` frames, or the last closed pair's end);
      * timestamps of the right stride are deferred: from the first timestamp token at or beyond (window - right
        stride) on, every timestamp sets `skip`, and the first timestamp after a skip is swallowed too (timestamps come
        in end/start pairs); with text pending from an earlier window, timestamps inside the left stride are swallowed;
      * otherwise a timestamp opens the segment if none is open, is ignored if it repeats the opening time, and else
        closes it: the text tokens the segment collected over the windows are merged by `merge_sequences`."""
    special = set(special_ids)
    segments, pending = [], []
    seg_start = None
    offset, skip = 0.0, False
    for w in windows:
        toks = [int(t) for t in w["tokens"]]
        if prompt_token_id is not None and toks and toks[0] == prompt_token_id:
            toks = toks[toks.index(decoder_start_token_id):] if decoder_start_token_id in toks else []
        stride = w.get("stride")
        hold_from = None                      # first deferred timestamp token of the right stride
        first_usable = timestamp_begin        # timestamps below it lie in the left stride
        if stride is not None:
            win, left, right = stride
            offset -= left
            if left:
                first_usable = left / time_precision + timestamp_begin
            if right:
                for t in reversed(toks):
                    if t >= timestamp_begin:
                        if hold_from is not None and (t - timestamp_begin) * time_precision < win - right:
                            break
                        hold_from = t
        text = []
        high, before_high, carried = 0.0, 0.0, 0.0      # seek-loop outputs: running / previous maximum, time of earlier segments
        for i, t in enumerate(toks):
            if t in special:
                continue
            if t < timestamp_begin:
                text.append(t)
                continue
            local = float((t - timestamp_begin) * time_precision)
            if local < high:
                lone_end = i >= 2 and not (toks[i - 1] >= timestamp_begin and toks[i - 2] >= timestamp_begin)
                if lone_end:
                    carried += time_precision * segment_size
                else:
                    high = before_high
                    carried += before_high
            before_high, high = high, local
            when = round((t - timestamp_begin) * time_precision + offset + carried, 2)
            if hold_from and t >= hold_from:
                skip = True
            elif skip or (pending and t < first_usable):
                skip = False
            elif seg_start is None:
                seg_start = when
            elif when != seg_start:
                pending.append(text)
                segments.append({"timestamp": (seg_start, when), "tokens": merge_sequences(pending)})
                pending, text, seg_start = [], [], None
        if stride is not None:
            offset += win - right
        if text:
            pending.append(text)
        elif not any(pending):
            pending, seg_start = [], None
    if pending:
        segments.append({"timestamp": (seg_start, None), "tokens": merge_sequences(pending)})
    return segments


def two_stage_pipeline(owner, ops, dev, batches, encode_fn, decode_fn, overlap, decode_cus, priority=-1):
    """Yields (batch, decode_fn(encode_fn(batch)) as a numpy array) in order.  overlap: the encoder stage of batch i+1 runs on one
    HIP stream while the token loop of batch i runs on another, with `decode_cus` CUs kept out of the persistent GEMM grids
    (dw_debug_set key 9) so that the token-step kernels find a CU at once; the streams live on `owner`.  Used by the long-form
    and the pseudo-labelling schedulers (run_eval.py:566-576, run_pseudo_labelling.py:861-996): same tokens either way."""
    if not overlap or len(batches) < 2:
        for batch in batches:
            yield batch, decode_fn(encode_fn(batch)).cpu().numpy()
        return
    main = torch.cuda.current_stream(dev)
    if not hasattr(owner, "_streams"):
        owner._streams = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev, priority=priority))
    s_enc, s_dec = owner._streams
    s_enc.wait_stream(main)
    s_dec.wait_stream(main)
    cus = 256 - int(decode_cus)
    if decode_cus > 0:
        ops.lib.dw_debug_set(9, cus - cus % 8)
    try:
        with torch.cuda.stream(s_enc):
            enc = encode_fn(batches[0])
            ready = torch.cuda.Event()
            ready.record(s_enc)
        for i, batch in enumerate(batches):
            cur_enc, cur_ready = enc, ready
            if i + 1 < len(batches):
                with torch.cuda.stream(s_enc):            # the NEXT batch's encoder is queued before this batch's token loop
                    enc = encode_fn(batches[i + 1])
                    ready = torch.cuda.Event()
                    ready.record(s_enc)
            with torch.cuda.stream(s_dec):
                s_dec.wait_event(cur_ready)
                ids_dev = decode_fn(cur_enc)
                done = torch.cuda.Event()
                done.record(s_dec)
            done.synchronize()                             # (cur_enc stays referenced until its last reader has finished)
            yield batch, ids_dev.cpu().numpy()
            del cur_enc
    finally:
        if decode_cus > 0:
            ops.lib.dw_debug_set(9, 256)
        main.wait_stream(s_enc)
        main.wait_stream(s_dec)


class LongFormTranscriber:
    """audio (list of 1-D float tensors/arrays at 16 kHz, any length) -> list of stitched text-token id lists."""

    def __init__(self, model, feature_extractor, batch_size=16, chunk_length_s=30.0, stride_length_s=None,
                 max_new_tokens=128, prompt_ids=None, eos_token_id=None, first_special_id=None, suppress_tokens=None,
                 begin_suppress_tokens=None, use_graphs=None, rank=0, world=1, return_timestamps=False,
                 no_timestamps_token_id=None, max_initial_timestamp_index=50, special_ids=None, overlap=False,
                 decode_cus=64):
        """return_timestamps=True: the windows are decoded under the timestamp rules (`prompt_ids` must then not end in
        <|notimestamps|>; `no_timestamps_token_id` is required) and stitched by their timestamp tokens; the call returns
        per utterance a list of {"timestamp": (start, end), "tokens": [...]} (`stitch_timestamped`)."""
        self.model, self.fe = model, feature_extractor
        self.B = int(batch_size)
        self.rank, self.world = int(rank), int(world)
        sr = feature_extractor.sampling_rate
        if stride_length_s is None:
            stride_length_s = chunk_length_s / 6
        if isinstance(stride_length_s, (int, float)):
            stride_length_s = [stride_length_s, stride_length_s]
        self.chunk_len = int(round(chunk_length_s * sr))
        self.stride_left = int(round(stride_length_s[0] * sr))
        self.stride_right = int(round(stride_length_s[1] * sr))
        if self.chunk_len > feature_extractor.n_samples:
            raise ValueError("chunk_length_s exceeds the 30 s receptive field of the Whisper encoder")
        if self.chunk_len < self.stride_left + self.stride_right:
            raise ValueError("Chunk length must be superior to stride length")
        d = model.dims
        self.eos = eos_token_id
        self.first_special = first_special_id if first_special_id is not None else \
            (eos_token_id if eos_token_id is not None else d.decoder_start_token_id - 1)
        dev = model.ops.device
        self.dev = dev
        self.prompt = torch.as_tensor([d.decoder_start_token_id] if prompt_ids is None else list(prompt_ids),
                                      dtype=torch.long, device=dev)
        self.max_new = int(max_new_tokens)
        self.return_timestamps = bool(return_timestamps)
        rules = None
        if self.return_timestamps:
            if no_timestamps_token_id is None or eos_token_id is None:
                raise ValueError("return_timestamps=True needs no_timestamps_token_id and eos_token_id")
            self.timestamp_begin = int(no_timestamps_token_id) + 1
            self.time_precision = feature_extractor.chunk_length / d.max_src          # 30 s / 1500 positions = 0.02 s
            # every id the tokenizer lists as special and that is not a timestamp: by default the ids from EOS up to
            # <|notimestamps|> (Whisper's vocabulary keeps them contiguous)
            # (the default covers the special-token block [eos, timestamp_begin); a pad id below eos -- the collator-style
            # 50256 with the multilingual eos 50257 -- is NOT in it and would end up in the segments as text, so the pad,
            # start and prompt ids this decoder knows are always added; a tokenizer's all_special_ids can be passed)
            self.special_ids = set(range(int(eos_token_id), self.timestamp_begin)) if special_ids is None else set(special_ids)
            self.special_ids |= {int(eos_token_id), int(d.pad_token_id), int(d.decoder_start_token_id)}
            # prompt ids: only the control tokens (>= the first special id).  A prompt that carries previous-text conditioning
            # (<|startofprev|> + text ids + <|startoftranscript|> ...) holds ordinary vocabulary ids too; adding those would
            # drop every later occurrence of the same words from the stitched segments.
            self.special_ids |= {int(t) for t in self.prompt.tolist() if int(t) >= int(self.first_special)}
            rules = dict(begin_index=len(self.prompt), no_timestamps_token_id=int(no_timestamps_token_id),
                         max_initial_timestamp_index=max_initial_timestamp_index)
        self.decoder = GreedyDecoder(model.engine, self.B, len(self.prompt) + self.max_new, eos_token_id=eos_token_id,
                                     suppress_tokens=suppress_tokens, begin_suppress_tokens=begin_suppress_tokens,
                                     use_graphs=use_graphs, timestamp_rules=rules)
        self._wave = torch.zeros((self.B, feature_extractor.n_samples), dtype=torch.float32, device=dev)
        # overlap=True (GPU only): the encoder of batch i+1 runs on one HIP stream while the token loop of batch i runs on
        # another.  A batch of 16 windows is ~41 ms of encoder (MFMA-bound, persistent GEMM grids on every CU) followed by ~32 ms
        # of token steps (128 x 18 dependent few-microsecond launches that leave the matrix pipe idle): back to back they add up,
        # side by side the token steps fit under the encoder.  The persistent GEMM grids hold a CU for a whole launch, so
        # `decode_cus` CUs are kept out of their grids (dw_debug_set key 9) for the token-step kernels to land on at once.
        self.overlap = bool(overlap) and torch.device(dev).type == "cuda"
        self.decode_cus = int(decode_cus)

    def plan(self, lengths):
        """[(utterance, start, length)] for all windows of all utterances, in pipeline order."""
        jobs = []
        for u, n in enumerate(lengths):
            for start, length, sl, sr, _ in chunk_spans(int(n), self.chunk_len, self.stride_left, self.stride_right):
                jobs.append((u, start, length, sl, sr))
        return jobs

    def __call__(self, audios, gather=False, group=None):
        """With world > 1 each rank transcribes a contiguous shard of the utterances (whole utterances: stitching is
        per utterance, there is no data-path collective); the others come back as None unless `gather=True`, which
        exchanges the stitched id lists in rank order (the reference's pad_across_processes + gather_for_metrics,
        run_distillation.py:1527, 1695-1697)."""
        model = self.model
        model._sync_shadow()
        n_all = len(audios)
        per = (n_all + self.world - 1) // self.world
        lo, hi = min(self.rank * per, n_all), min((self.rank + 1) * per, n_all)
        audios = [torch.as_tensor(np.asarray(a, dtype=np.float32) if not torch.is_tensor(a) else a,
                                  dtype=torch.float32).reshape(-1).to(self.dev) for a in audios[lo:hi]]
        mine = self._transcribe(audios)
        if self.world == 1:
            return mine
        if gather:
            from .gather import gather_token_lists
            rows = gather_token_lists(mine, self.eos if self.eos is not None else 0, self.dev, group=group)
            if self.return_timestamps:
                raise NotImplementedError("gather=True exchanges plain id lists; gather timestamped segments per rank")
            if len(rows) != n_all:
                raise RuntimeError(f"gathered {len(rows)} transcripts for {n_all} utterances: ranks disagree on the input")
            return rows
        return [None] * lo + mine + [None] * (n_all - hi)

    def _encode_batch(self, audios, batch):
        model = self.model
        self._wave.zero_()
        for r, (u, start, length, _, _) in enumerate(batch):
            self._wave[r, :length].copy_(audios[u][start:start + length])
        feats = model.ops.logmel(self._wave, self.fe._filt)
        enc, _ = model.engine.encode(feats, save=False)
        return enc

    def _decoded_batches(self, audios, jobs, prompt):
        """Yields (batch, ids as a numpy array) in order.  overlap: two-stage pipeline over two HIP streams (see __init__)."""
        batches = [jobs[b0:b0 + self.B] for b0 in range(0, len(jobs), self.B)]
        yield from two_stage_pipeline(self, self.model.ops, self.dev, batches, lambda batch: self._encode_batch(audios, batch),
                                      lambda enc: self.decoder.run(enc, prompt, self.max_new), self.overlap, self.decode_cus)

    def _transcribe(self, audios):
        jobs = self.plan([a.numel() for a in audios])
        per_utt = [[] for _ in audios]
        prompt = self.prompt[None, :].expand(self.B, -1).contiguous()
        for batch, ids in self._decoded_batches(audios, jobs, prompt):
            sr_hz = float(self.fe.sampling_rate)
            for r, (u, _, length, sl, sr) in enumerate(batch):
                row = ids[r, self.prompt.numel():]
                if self.return_timestamps:
                    # the pipeline hands `_decode_asr` the whole generated row (EOS / padding are special ids) and the
                    # window's (length, left stride, right stride) in seconds
                    per_utt[u].append({"tokens": [int(x) for x in row], "stride": (length / sr_hz, sl / sr_hz, sr / sr_hz)})
                    continue
                text = [int(x) for x in row if x < self.first_special]
                if text:                                   # (_decode_asr only keeps windows that produced text tokens)
                    per_utt[u].append(text)
        if self.return_timestamps:
            return [stitch_timestamped(w, self.timestamp_begin, self.special_ids, time_precision=self.time_precision)
                    for w in per_utt]
        return [merge_sequences(seqs) for seqs in per_utt]
