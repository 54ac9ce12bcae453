"""Pseudo-labelling throughput path: teacher-only batched generate over 30 s packs.

Reference: run_pseudo_labelling.py:632-673 (`concatenate_dataset`: consecutive samples of one speaker are appended
until the next one would exceed `max_input_length` = 30 s; `condition_on_prev` marks a pack that continues the
previous pack's speaker) and 861-996 (the teacher's `generate` over the packed batches under data parallelism, one
process per GPU, each on its own shard).  Here the packs are gathered on the GPU straight into the log-mel kernel's
[B, 480000] buffer and decoded by decoding.GreedyDecoder; sharding across ranks is by pack index (no collective).
"""
import numpy as np
import torch

from .decoding import GreedyDecoder


def pack_plan(lengths, speaker_ids=None, max_input_length=480000):
    """Greedy packing rule of `concatenate_dataset`: returns (packs, condition_on_prev) where packs[i] is the list of
    sample indices concatenated into pack i.  A sample joins the current pack iff it has the pack's speaker and the
    summed length stays <= max_input_length; otherwise it opens a new pack, flagged 1 when the speaker continues."""
    n = len(lengths)
    if n == 0:
        return [], []
    spk = [None] * n if speaker_ids is None else list(speaker_ids)
    packs, cond = [[0]], [0]
    cur_len, cur_spk = int(lengths[0]), spk[0]
    for i in range(1, n):
        same = spk[i] == cur_spk
        if same and int(lengths[i]) + cur_len <= max_input_length:
            packs[-1].append(i)
            cur_len += int(lengths[i])
        else:
            packs.append([i])
            cond.append(1 if same else 0)
            cur_len, cur_spk = int(lengths[i]), spk[i]
    return packs, cond


def shard(items, rank, world):
    """Contiguous per-rank shard of the pack list (what `accelerator.prepare(dataloader)` does for the reference)."""
    per = (len(items) + world - 1) // world
    return items[rank * per:(rank + 1) * per]


class PseudoLabeller:
    """audios (list of 1-D 16 kHz waveforms, each <= 30 s) -> (token id lists per pack, packs, condition_on_prev)."""

    def __init__(self, model, feature_extractor, batch_size=16, max_new_tokens=255, prompt_ids=None, eos_token_id=None,
                 timestamp_rules=None, use_graphs=None, rank=0, world=1, suppress_tokens=None,
                 begin_suppress_tokens=None, num_beams=1, overlap=False, decode_cus=64):
        self.model, self.fe = model, feature_extractor
        self.B, self.max_new = int(batch_size), int(max_new_tokens)
        self.rank, self.world = rank, world
        self.num_beams = int(num_beams)      # generation_num_beams of run_pseudo_labelling.py:835-843
        self._beam_kw = dict(suppress_tokens=suppress_tokens, begin_suppress_tokens=begin_suppress_tokens)
        d = model.dims
        dev = model.ops.device
        self.dev = dev
        self.prompt = torch.as_tensor([d.decoder_start_token_id] if prompt_ids is None else list(prompt_ids),
                                      dtype=torch.long, device=dev)
        if timestamp_rules is not None:
            timestamp_rules = dict(timestamp_rules, begin_index=len(self.prompt))
        self._ts_rules = timestamp_rules
        self.eos = eos_token_id
        self.decoder = GreedyDecoder(model.engine, self.B, len(self.prompt) + self.max_new, eos_token_id=eos_token_id,
                                     suppress_tokens=suppress_tokens, begin_suppress_tokens=begin_suppress_tokens,
                                     use_graphs=use_graphs, timestamp_rules=timestamp_rules)
        self._wave = torch.zeros((self.B, feature_extractor.n_samples), dtype=torch.float32, device=dev)
        # overlap=True: the encoder of the next batch of packs beside the token loop of the current one (longform.two_stage_pipeline;
        # plain greedy decoding only -- the seek loop and beam search encode / decode in their own order)
        self.overlap = bool(overlap) and torch.device(dev).type == "cuda"
        self.decode_cus = int(decode_cus)

    def __call__(self, audios, speaker_ids=None, gather=False, group=None):
        """`gather=True`: every rank returns the labels of ALL packs (rank-ordered exchange of the id lists, the
        reference's pad_across_processes + gather_for_metrics, run_pseudo_labelling.py:893-895); otherwise packs of
        other ranks' shards come back as None."""
        model = self.model
        model._sync_shadow()
        audios = [torch.as_tensor(np.asarray(a, dtype=np.float32) if not torch.is_tensor(a) else a,
                                  dtype=torch.float32).reshape(-1).to(self.dev) for a in audios]
        packs, cond = pack_plan([a.numel() for a in audios], speaker_ids, self.fe.n_samples)
        mine = shard(list(range(len(packs))), self.rank, self.world)
        prompt = self.prompt[None, :].expand(self.B, -1).contiguous()
        out = {}
        def wave_to_features(batch):
            self._wave.zero_()
            for r, pi in enumerate(batch):
                pos = 0
                for si in packs[pi]:
                    n = audios[si].numel()
                    self._wave[r, pos:pos + n].copy_(audios[si])
                    pos += n
            return model.ops.logmel(self._wave, self.fe._filt)

        batches = [mine[b0:b0 + self.B] for b0 in range(0, len(mine), self.B)]
        plain = not (self._ts_rules is not None and self.num_beams == 1) and self.num_beams == 1
        if plain:
            from .longform import two_stage_pipeline
            for batch, ids in two_stage_pipeline(self, model.ops, self.dev, batches,
                                                 lambda b: model.engine.encode(wave_to_features(b), save=False)[0],
                                                 lambda enc: self.decoder.run(enc, prompt, self.max_new), self.overlap,
                                                 self.decode_cus):
                for r, pi in enumerate(batch):
                    row = ids[r, self.prompt.numel():].tolist()
                    if self.eos is not None and self.eos in row:
                        row = row[:row.index(self.eos)]
                    out[pi] = [int(x) for x in row]
            batches = []
        for batch in batches:
            feats = wave_to_features(batch)
            if self._ts_rules is not None and self.num_beams == 1:
                # generate(..., return_timestamps=True) of the reference is a seek loop (TF:784-903): a pack is decoded
                # in as many passes as its predicted end-of-segment timestamps require
                nb = len(batch)
                segs = model.seek_decode(feats[:nb], [feats.shape[-1]] * nb, [self.prompt.tolist()] * nb,
                                         lambda P: (self.max_new, 0), self.eos,
                                         self.eos, self._ts_rules["no_timestamps_token_id"],
                                         self._ts_rules.get("max_initial_timestamp_index"), **self._beam_kw)
                for r, pi in enumerate(batch):
                    out[pi] = [int(t) for sg in segs[r] for t in sg["tokens"]]
                continue
            enc, _ = model.engine.encode(feats, save=False)
            from .decoding import beam_search_decode
            ids = beam_search_decode(model.engine, enc, prompt, self.max_new, self.num_beams, self.eos,
                                     timestamp_rules=self._ts_rules, **self._beam_kw).cpu().numpy()
            for r, pi in enumerate(batch):
                row = ids[r, self.prompt.numel():].tolist()
                if self.eos is not None and self.eos in row:
                    row = row[:row.index(self.eos)]
                out[pi] = [int(x) for x in row]
        if gather and self.world > 1:
            from .gather import gather_token_lists
            fill = self.eos if self.eos is not None else 0
            rows = gather_token_lists([out[i] for i in mine], fill, self.dev, group=group)
            if len(rows) != len(packs):
                raise RuntimeError(f"gathered {len(rows)} label rows for {len(packs)} packs: ranks disagree on the plan")
            return rows, packs, cond
        return [out.get(i) for i in range(len(packs))], packs, cond
