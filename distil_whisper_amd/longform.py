"""Chunked long-form transcription scheduler (BASELINE config 5).

Reference behaviour: run_eval.py:566-576, 726-786 drives the `transformers` ASR pipeline with `chunk_length_s`:
  * the audio is cut by `chunk_iter` (TF:pipelines/automatic_speech_recognition.py:61-84) into windows of
    `chunk_len` samples that advance by `chunk_len - stride_left - stride_right` (stride = chunk/6 by default,
    TF:...automatic_speech_recognition.py `preprocess`), every window is padded to 30 s by the feature extractor;
  * the windows of all utterances are batched through encoder + greedy generate;
  * per utterance, the text tokens of consecutive windows are stitched by the sliding longest-common-sequence rule of
    `_find_longest_common_sequence` (TF:models/whisper/tokenization_whisper.py, called from `_decode_asr` when no
    timestamps are requested).
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


class LongFormTranscriber:
    """audio (list of 1-D float tensors/arrays at 16 kHz, any length) -> list of stitched text-token id lists."""

    def __init__(self, model, feature_extractor, batch_size=16, chunk_length_s=30.0, stride_length_s=None,
                 max_new_tokens=128, prompt_ids=None, eos_token_id=None, first_special_id=None, suppress_tokens=None,
                 begin_suppress_tokens=None, use_graphs=None, rank=0, world=1):
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
        self.decoder = GreedyDecoder(model.engine, self.B, len(self.prompt) + self.max_new, eos_token_id=eos_token_id,
                                     suppress_tokens=suppress_tokens, begin_suppress_tokens=begin_suppress_tokens,
                                     use_graphs=use_graphs)
        self._wave = torch.zeros((self.B, feature_extractor.n_samples), dtype=torch.float32, device=dev)

    def plan(self, lengths):
        """[(utterance, start, length)] for all windows of all utterances, in pipeline order."""
        jobs = []
        for u, n in enumerate(lengths):
            for start, length, _, _, _ in chunk_spans(int(n), self.chunk_len, self.stride_left, self.stride_right):
                jobs.append((u, start, length))
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
            if len(rows) != n_all:
                raise RuntimeError(f"gathered {len(rows)} transcripts for {n_all} utterances: ranks disagree on the input")
            return rows
        return [None] * lo + mine + [None] * (n_all - hi)

    def _transcribe(self, audios):
        model = self.model
        jobs = self.plan([a.numel() for a in audios])
        per_utt = [[] for _ in audios]
        prompt = self.prompt[None, :].expand(self.B, -1).contiguous()
        for b0 in range(0, len(jobs), self.B):
            batch = jobs[b0:b0 + self.B]
            self._wave.zero_()
            for r, (u, start, length) in enumerate(batch):
                self._wave[r, :length].copy_(audios[u][start:start + length])
            feats = model.ops.logmel(self._wave, self.fe._filt)
            enc, _ = model.engine.encode(feats, save=False)
            ids = self.decoder.run(enc, prompt, self.max_new).cpu().numpy()
            for r, (u, _, _) in enumerate(batch):
                row = ids[r, self.prompt.numel():]
                text = [int(x) for x in row if x < self.first_special]
                if text:                                   # (_decode_asr only keeps windows that produced text tokens)
                    per_utt[u].append(text)
        return [merge_sequences(seqs) for seqs in per_utt]
