"""Batched greedy KV-cache decoding on the engine, with the per-token launch sequence captured in HIP graphs.

Reference behaviour: `model.generate(...)` as called by run_distillation.py:1524-1528 (`generate_step`) and
run_eval.py:739 / 806-844 (`benchmark_gen`: fixed number of new tokens on random encoder input) with greedy search
(`num_beams=1`, `top_k=0`): TF:generation_whisper.py:383 + the cache branches of TF:modeling_whisper.py:312-335, with
the `SuppressTokensLogitsProcessor` / `SuppressTokensAtBeginLogitsProcessor` steps (TF:generation_whisper.py:1774-1812).

One decoding step of the 2-layer student is ~30 kernel launches of a few microseconds each with M = batch rows, i.e.
launch-latency bound when driven from Python.  Every token position t has fixed shapes (cache length t+1, position
embedding row t), so the step for position t is captured once into a HIP graph and replayed for every later batch of
chunks; all tensors the graphs touch (current ids, token matrix, done flags, K/V caches) are allocated once.
"""
import torch


def apply_timestamp_rules(scores, tokens, n, begin_index, no_timestamps_token_id, eos_token_id,
                          max_initial_timestamp_index=None, return_rule_margin=False):
    """The decoding rules of `WhisperTimeStampLogitsProcessor` (TF:generation/logits_process.py; installed by
    TF:generation_whisper.py:1774-1812 when `return_timestamps=True`, run_eval.py:690-739) on a whole batch without
    host round trips: scores f32 [B, V] (modified copy returned), tokens int64 [B, >= n] of which the first n are
    the sequence so far (prompt included), begin_index = length of the forced prefix.
      * <|notimestamps|> is never sampled;
      * timestamps come in pairs: after text+timestamp only a timestamp or EOS may follow, after two timestamps only
        text; a timestamp may not be smaller than the last one emitted (nor repeat it when it closed a pair);
      * the first sampled token must be a timestamp, at most `max_initial_timestamp_index` steps in;
      * if the timestamps together are more probable than the most probable text token, a timestamp is sampled."""
    sc = scores.clone()
    B, V = sc.shape
    tb = no_timestamps_token_id + 1
    neg = float("-inf")
    col = torch.arange(V, device=sc.device)[None, :]
    sc[:, no_timestamps_token_id] = neg
    L = n - begin_index
    if L >= 1:
        last_ts = tokens[:, n - 1] >= tb
        pen_ts = tokens[:, n - 2] >= tb if L >= 2 else torch.ones_like(last_ts)
        sc = sc.masked_fill((last_ts & pen_ts)[:, None] & (col >= tb), neg)
        sc = sc.masked_fill((last_ts & ~pen_ts)[:, None] & (col < eos_token_id), neg)
        seq = tokens[:, begin_index:n]
        is_ts = seq >= tb
        any_ts = is_ts.any(1)
        pos = (is_ts.long() * torch.arange(1, L + 1, device=sc.device)[None, :]).max(1).values   # 1-based, 0 = none
        last_val = seq.gather(1, (pos - 1).clamp(min=0)[:, None])[:, 0]
        ts_last = torch.where(last_ts & ~pen_ts, last_val, last_val + 1)
        sc = sc.masked_fill(any_ts[:, None] & (col >= tb) & (col < ts_last[:, None]), neg)
    if L == 0:
        sc = sc.masked_fill(col < tb, neg)
        if max_initial_timestamp_index is not None:
            sc = sc.masked_fill(col > tb + max_initial_timestamp_index, neg)
    lp = torch.log_softmax(sc.float(), dim=-1)
    ts_lp = lp[:, tb:].logsumexp(-1)
    text_max = lp[:, :tb].max(-1).values
    out = sc.masked_fill((ts_lp > text_max)[:, None] & (col < tb), neg)
    if return_rule_margin:          # distance of the probability-mass decision from its threshold (fixture generator)
        return out, (ts_lp - text_max).abs()
    return out


def apply_repetition_penalty(scores, input_ids, penalty):
    """`RepetitionPenaltyLogitsProcessor` (TF:generation/logits_process.py): the scores of every token already in the row's
    sequence (prompt included) are divided by `penalty` when positive and multiplied when negative."""
    sc = torch.gather(scores, 1, input_ids)
    sc = torch.where(sc < 0, sc * penalty, sc / penalty)
    return scores.scatter(1, input_ids, sc)


def apply_no_repeat_ngram(scores, input_ids, n):
    """`NoRepeatNGramLogitsProcessor`: a token that would complete an n-gram the row's sequence already contains is banned.
    Vectorised: every window of n-1 tokens equal to the sequence's last n-1 tokens bans the token that followed it."""
    B, L = input_ids.shape
    if n <= 0 or L + 1 < n:
        return scores
    if n == 1:
        return scores.scatter(1, input_ids, float("-inf"))
    tail = input_ids[:, L - (n - 1):]                                     # [B, n-1]
    win = input_ids.unfold(1, n - 1, 1)[:, : L - (n - 1)]                  # [B, L-n+1, n-1]: windows that HAVE a follower
    hit = (win == tail[:, None, :]).all(-1)                               # [B, L-n+1]
    follow = input_ids[:, n - 1:]                                          # [B, L-n+1]
    # (several windows may name the same token: any hit bans it)
    ban_any = torch.zeros_like(scores, dtype=torch.int32).scatter_add_(1, follow, hit.to(torch.int32)) > 0
    return scores.masked_fill(ban_any, float("-inf"))


def warp_and_sample(scores, temperature=None, top_k=None, top_p=None, generator=None):
    """The sampling tail of `GenerationMixin._sample`: TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper (in
    that order, TF `_get_logits_processor`), softmax, one `torch.multinomial` draw per row."""
    sc = scores
    if temperature is not None and float(temperature) != 1.0:
        sc = sc / float(temperature)
    if top_k is not None and int(top_k) > 0:
        k = min(int(top_k), sc.shape[-1])
        kth = torch.topk(sc, k)[0][..., -1, None]
        sc = sc.masked_fill(sc < kth, float("-inf"))
    if top_p is not None and float(top_p) < 1.0:
        srt, idx = torch.sort(sc, descending=False)
        cum = srt.softmax(-1).cumsum(-1)
        remove = cum <= (1.0 - float(top_p))
        remove[..., -1:] = False                                          # min_tokens_to_keep = 1
        sc = sc.masked_fill(remove.scatter(1, idx, remove), float("-inf"))
    probs = torch.softmax(sc, dim=-1)
    return torch.multinomial(probs, num_samples=1, generator=generator)[:, 0]


class GreedyDecoder:
    def __init__(self, engine, batch, max_len, eos_token_id=None, suppress_tokens=None, begin_suppress_tokens=None,
                 use_graphs=None, check_every=16, timestamp_rules=None, pad_token_id=None, soft=None):
        self.eng, self.B, self.max_len = engine, int(batch), int(max_len)
        d = engine.dims
        if self.max_len > d.max_tgt:
            raise ValueError(f"max_len = {max_len} exceeds max_target_positions = {d.max_tgt}")
        dev = engine.ops.device
        self.dev = dev
        self.eos = eos_token_id
        self.use_graphs = (torch.device(dev).type == "cuda") if use_graphs is None else bool(use_graphs)
        self.check_every = int(check_every)
        self.tokens = torch.zeros((self.B, self.max_len), dtype=torch.long, device=dev)
        self.cur = torch.zeros((self.B, 1), dtype=torch.long, device=dev)
        self.done = torch.zeros((self.B,), dtype=torch.bool, device=dev)

        def mask(ids):
            if ids is None or len(ids) == 0:
                return None
            m = torch.zeros((d.vocab,), dtype=torch.uint8, device=dev)
            m[torch.as_tensor(list(ids), dtype=torch.long, device=dev)] = 1
            return m
        self.suppress = mask(suppress_tokens)
        self.begin_suppress = mask(begin_suppress_tokens)
        # dict(begin_index=, no_timestamps_token_id=, max_initial_timestamp_index=): WhisperTimeStampLogitsProcessor
        self.timestamp_rules = timestamp_rules
        if timestamp_rules is not None and eos_token_id is None:
            raise ValueError("timestamp rules need eos_token_id")
        self.fill = -1 if eos_token_id is None else (eos_token_id if pad_token_id is None else pad_token_id)
        # soft = dict(do_sample=, temperature=, top_k=, top_p=, repetition_penalty=, no_repeat_ngram_size=, generator=):
        # history-dependent processors and sampling of `GenerationMixin` (TF `_get_logits_processor` order: repetition
        # penalty, no-repeat n-gram, min-new-tokens, Whisper's suppress / begin-suppress / timestamp rules, then the
        # temperature / top-k / top-p warpers and the multinomial draw).  The selection then runs as torch ops on the
        # step's logits instead of dw_greedy_select, eagerly (no HIP-graph replay: the history grows every step).
        self.soft = soft
        if soft is not None:
            self.use_graphs = False
        self.cache = None
        self.graphs = {}
        self.pool = None
        self._warm = False

    # mode 0: position t+1 is still inside the prompt (teacher forcing); 1: first generated token; 2: later tokens
    def _step(self, t, mode, no_eos=False):
        eng, d = self.eng, self.eng.dims
        self.cache["t"] = t
        logits = eng.decode_step(self.cur, self.cache)
        r = self.timestamp_rules
        if self.soft is not None and mode != 0:
            self._select_soft(logits, t + 1, mode, no_eos)
            return
        # logits processors of the reference (min-new-tokens, begin-suppress, suppress, timestamp rules), argmax and the
        # EOS bookkeeping in one launch (csrc/decode.hip); the next token lands in tokens[:, t+1] and in cur
        eng.ops.greedy_select(
            logits, d.vocab, self.tokens, t + 1, self.cur, suppress=self.suppress, begin_suppress=self.begin_suppress,
            first=(mode == 1), no_eos=no_eos, forced=(mode == 0),
            ts_begin=-1 if r is None else r["no_timestamps_token_id"] + 1,
            max_initial=-1 if (r is None or r.get("max_initial_timestamp_index") is None)
            else r["max_initial_timestamp_index"],
            begin_index=1 if r is None else r["begin_index"], eos=-1 if self.eos is None else self.eos,
            fill=self.fill, done=self.done)

    def _select_soft(self, logits, n, mode, no_eos):
        """Token n of every row from `logits` with the history-dependent processors / sampling of `self.soft`."""
        d, so, r = self.eng.dims, self.soft, self.timestamp_rules
        B = self.B
        neg = float("-inf")
        sc = logits[:B, :d.vocab].float()
        hist = self.tokens[:, :n]
        rp = so.get("repetition_penalty")
        if rp is not None and float(rp) != 1.0:
            sc = apply_repetition_penalty(sc, hist, float(rp))
        if so.get("no_repeat_ngram_size"):
            sc = apply_no_repeat_ngram(sc, hist, int(so["no_repeat_ngram_size"]))
        if no_eos and self.eos is not None:
            sc[:, self.eos] = neg
        if self.suppress is not None:
            sc = sc.masked_fill(self.suppress[:d.vocab].bool()[None, :], neg)
        if mode == 1 and self.begin_suppress is not None:
            sc = sc.masked_fill(self.begin_suppress[:d.vocab].bool()[None, :], neg)
        if r is not None:
            sc = apply_timestamp_rules(sc, self.tokens, n, int(r["begin_index"]), int(r["no_timestamps_token_id"]), self.eos,
                                       r.get("max_initial_timestamp_index"))
        if so.get("do_sample"):
            nxt = warp_and_sample(sc, so.get("temperature"), so.get("top_k"), so.get("top_p"), so.get("generator"))
        else:
            nxt = sc.argmax(-1)
        if self.eos is not None:
            nxt = torch.where(self.done, torch.full_like(nxt, self.fill), nxt)
            self.done.logical_or_(nxt == self.eos)
        self.tokens[:, n].copy_(nxt)
        self.cur.copy_(nxt.view(B, 1))

    def _run_step(self, t, mode, no_eos=False):
        if not self.use_graphs:
            self._step(t, mode, no_eos)
            return
        g = self.graphs.get((t, mode, no_eos))
        if g is None:
            if not self._warm:
                # one eager step first: lazy initialisation inside torch must not happen under stream capture
                keep = (self.cur.clone(), self.tokens.clone(), self.done.clone())
                self._step(t, mode, no_eos)
                self.cur.copy_(keep[0]); self.tokens.copy_(keep[1]); self.done.copy_(keep[2])
                torch.cuda.synchronize(self.dev)
                self.pool = torch.cuda.graph_pool_handle()
                self._warm = True
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=self.pool):
                self._step(t, mode, no_eos)
            self.graphs[(t, mode, no_eos)] = g
        g.replay()

    def run(self, enc_out, prompt_ids, max_new_tokens, min_new_tokens=0):
        """enc_out: encoder output of `batch` chunks (engine layout, rows = batch * max_source_positions);
        prompt_ids: int64 [batch, P] forced decoder prefix (P >= 1; position 0 = <|startoftranscript|>).
        Returns int64 [batch, P + n] with n <= max_new_tokens (stops early once every row has produced EOS)."""
        B, P = prompt_ids.shape
        if B != self.B:
            raise ValueError(f"decoder built for batch {self.B}, got {B}")
        total = P + int(max_new_tokens)
        if total > self.max_len:
            raise ValueError(f"prompt + max_new_tokens = {total} exceeds the decoder's max_len = {self.max_len}")
        self.cache = self.eng.decode_init(enc_out, B, self.max_len, cache=self.cache)
        self.tokens.zero_()
        self.tokens[:, :P].copy_(prompt_ids)
        self.done.zero_()
        self.cur.copy_(self.tokens[:, 0:1])
        n = P
        for t in range(total - 1):
            mode = 0 if t + 1 < P else (1 if t + 1 == P else 2)
            self._run_step(t, mode, bool(mode) and (t + 1 - P) < int(min_new_tokens))
            n = t + 2
            if self.eos is not None and mode and (t + 2 - P) % self.check_every == 0 and bool(self.done.all()):
                break
        return self.tokens[:, :n].clone()


def assisted_greedy_decode(target, assistant, enc_target, enc_assistant, prompt_ids, max_new_tokens,
                           num_assistant_tokens=5, eos_token_id=None, suppress_tokens=None, min_new_tokens=0,
                           pad_token_id=None, use_cache=True, timestamp_rules=None):
    """Speculative (assisted) greedy decoding: the small `assistant` engine drafts `num_assistant_tokens` tokens, the
    `target` engine scores all of them in ONE decoder pass and keeps the longest draft prefix that equals its own
    greedy choices plus its next token -- the output is token-for-token what target-only greedy decoding produces.

    Reference: run_eval.py:578-599, 706-707 (`assistant_model` of `generate`; the distilled student drafts for the
    teacher and shares its encoder output) and flax/run_speculative_decoding.py:76-107.  target / assistant:
    WhisperEngine; enc_*: their encoder outputs in engine layout (the same tensor when the encoders are shared);
    prompt_ids int64 [B, P].  With a batch the accepted length is the minimum over the rows that are still running
    (every emitted token is still each row's own greedy token).  suppress_tokens / min_new_tokens are the target's
    logits rules (SuppressTokensLogitsProcessor, MinNewTokensLengthLogitsProcessor); finished rows are filled with
    pad_token_id.  timestamp_rules (dict(begin_index, no_timestamps_token_id, max_initial_timestamp_index)): the
    WhisperTimeStampLogitsProcessor rules on every scored position, for the drafts and for the verification alike (the
    reference hands the same processors to the candidate generator): `return_timestamps=True` with an assistant.

    use_cache: both models keep KV caches.  The target verifies with `decode_multi` -- only the tokens it has not
    consumed yet (last accepted + drafts) go through the decoder, against the cached keys/values with the
    bottom-right aligned causal mask -- and rejected positions are dropped by rolling the cache position back; the
    assistant catches up on the accepted tokens the same way and drafts one token per step.  use_cache=False re-decodes
    the whole prefix every time (cross-check).  Returns (ids [B, P + n], drafted, accepted)."""
    dt, da = target.dims, assistant.dims
    B, P0 = prompt_ids.shape
    ids = prompt_ids.clone()
    dev = ids.device
    total = P0 + int(max_new_tokens)
    if total > min(dt.max_tgt, da.max_tgt):
        raise ValueError(f"prompt + max_new_tokens = {total} exceeds max_target_positions")
    done = torch.zeros(B, dtype=torch.bool, device=dev)
    drafted = accepted = 0
    fill = eos_token_id if pad_token_id is None else pad_token_id
    sup = {}

    def sup_mask(V):
        if not suppress_tokens:
            return None
        if V not in sup:
            m = torch.zeros((V,), dtype=torch.float32, device=dev)
            m[torch.as_tensor([t for t in suppress_tokens if t < V], dtype=torch.long, device=dev)] = float("-inf")
            sup[V] = m
        return sup[V]

    def pick(logits, d, first_pos, hist=None):
        """greedy tokens from scores [B, n, V] whose row j predicts the token at sequence index first_pos + j (hist: the
        sequence those positions continue, at least first_pos + n - 1 tokens: history of the timestamp rules)"""
        sc = logits.float()
        m = sup_mask(d.vocab)
        if m is not None:
            sc = sc + m
        if eos_token_id is not None and min_new_tokens > 0 and eos_token_id < d.vocab:
            gen_idx = torch.arange(first_pos - P0, first_pos - P0 + sc.shape[1], device=dev)
            sc = sc.clone()
            sc[:, :, eos_token_id] = torch.where((gen_idx < min_new_tokens)[None, :], float("-inf"),
                                                 sc[:, :, eos_token_id])
        if timestamp_rules is not None:
            tr = timestamp_rules
            sc = torch.stack([apply_timestamp_rules(sc[:, j], hist, first_pos + j, int(tr["begin_index"]),
                                                    int(tr["no_timestamps_token_id"]), eos_token_id,
                                                    tr.get("max_initial_timestamp_index"))
                              for j in range(sc.shape[1])], 1)
        return sc.argmax(-1)

    def scores_nocache(eng, d, seq, enc, first):
        T = seq.shape[1]
        logits, _ = eng.decode(seq.contiguous(), enc, save=False)
        return logits[: B * T, : d.vocab].view(B, T, -1)[:, first:]

    ct = ca = None
    if use_cache:
        ct = target.decode_init(enc_target, B, total)
        ca = assistant.decode_init(enc_assistant, B, total)

    def scores_cached(eng, d, cache, seq):
        """feed the tokens of seq the cache has not consumed; scores [B, n, V] for those positions"""
        new = seq[:, cache["t"]:].contiguous()
        n = new.shape[1]
        logits = eng.decode_multi(new, cache)
        return logits[: B * n, : d.vocab].view(B, n, -1)

    while ids.shape[1] < total and not (eos_token_id is not None and bool(done.all())):
        L = ids.shape[1]
        k = min(int(num_assistant_tokens), total - L - 1)
        draft = ids
        for j in range(k):                                   # the assistant drafts k tokens greedily
            if use_cache:
                sc = scores_cached(assistant, da, ca, draft)[:, -1:]
            else:
                sc = scores_nocache(assistant, da, draft, enc_assistant, draft.shape[1] - 1)
            nxt = pick(sc, da, draft.shape[1], draft)[:, -1]
            draft = torch.cat([draft, nxt[:, None]], 1)
        if use_cache:
            sc = scores_cached(target, dt, ct, draft)[:, -(k + 1):]
        else:
            sc = scores_nocache(target, dt, draft, enc_target, L - 1)
        own = pick(sc, dt, L, draft)                         # [B, k + 1]: target's choice after each prefix
        if k > 0:
            agree = (own[:, :k] == draft[:, L:]) | done[:, None]
            n_ok = int(agree.long().cumprod(1).sum(1).min().item())
        else:
            n_ok = 0
        drafted += k
        accepted += n_ok
        new = own[:, : n_ok + 1]                             # accepted draft tokens (== own) + the target's next token
        if eos_token_id is not None:
            for j in range(new.shape[1]):
                col = torch.where(done, torch.full_like(new[:, j], fill), new[:, j])
                new[:, j] = col
                done = done | (col == eos_token_id)
        ids = torch.cat([ids, new], 1)
        if use_cache:
            # positions L .. L+n_ok-1 hold accepted drafts (their K/V are valid); everything later is dropped
            ct["t"] = min(ct["t"], L + n_ok)
            ca["t"] = min(ca["t"], L + n_ok)
    return ids, drafted, accepted


def beam_search_decode(engine, enc_out, prompt_ids, max_new_tokens, num_beams, eos_token_id, pad_token_id=None,
                       suppress_tokens=None, begin_suppress_tokens=None, min_new_tokens=0, length_penalty=1.0,
                       early_stopping=False, timestamp_rules=None):
    """Beam search over the KV-cache decoder: `generate(num_beams=k)` of the reference (run_eval.py:143, 693;
    run_distillation.py:1428-1436; TF:generation/utils.py `_beam_search`, the vectorised v5 algorithm) -- per step the
    log-softmax of every live beam plus its running score, the top 2k continuations per utterance, the k best open ones
    carried on (their K/V cache rows gathered in place), finished ones merged into the k best finished hypotheses
    under the length penalty, and the early-stopping heuristic of `early_stopping` in {False, True, "never"}.
    The decoder passes are the engine's cached passes over B * k rows (prompt prefill in one multi-token pass, then
    token steps); the bookkeeping is index arithmetic on [B, 2k] tensors on the device.
    prompt_ids int64 [B, P] -> sequences int64 [B, P + n] (best finished hypothesis per row, padded with pad_token_id)."""
    dev = prompt_ids.device
    d = engine.dims
    B, P = prompt_ids.shape
    nb, V = int(num_beams), d.vocab
    max_length = P + int(max_new_tokens)
    if eos_token_id is None:
        raise ValueError("beam search needs eos_token_id")
    eos = int(eos_token_id)
    fill = int(pad_token_id) if pad_token_id is not None else eos
    keep = 2 * nb                                       # (number of EOS ids + 1) * num_beams continuations per utterance
    neg = -1.0e9

    def mask_of(ids):
        m = torch.zeros(V, dtype=torch.bool, device=dev)
        if ids:
            m[torch.as_tensor(list(ids), dtype=torch.long, device=dev)] = True
        return m
    sup, bsup = mask_of(suppress_tokens), mask_of(begin_suppress_tokens)

    def gather(t, idx):                                  # t [B, n, ...], idx [B, m] -> [B, m, ...]
        ix = idx
        while ix.dim() < t.dim():
            ix = ix.unsqueeze(-1)
        return torch.gather(t, 1, ix.expand(*idx.shape, *t.shape[2:]))

    Lk, D = d.max_src, d.d_model
    enc_rep = enc_out[:B * Lk].view(B, Lk, D).repeat_interleave(nb, 0).reshape(B * nb * Lk, D).contiguous()
    cache = engine.decode_init(enc_rep, B * nb, max_length)
    running = torch.full((B, nb, max_length), fill, dtype=torch.long, device=dev)
    running[:, :, :P] = prompt_ids[:, None, :]
    sequences = running.clone()
    run_scores = torch.zeros((B, nb), dtype=torch.float32, device=dev)
    run_scores[:, 1:] = neg
    beam_scores = torch.full((B, nb), neg, dtype=torch.float32, device=dev)
    finished = torch.zeros((B, nb), dtype=torch.bool, device=dev)
    lengths = torch.zeros((B, nb), dtype=torch.long, device=dev)          # generated tokens of the finished hypotheses
    unsat = torch.ones((B, 1), dtype=torch.bool, device=dev)
    top_mask = torch.cat([torch.ones(nb, dtype=torch.bool, device=dev), torch.zeros(keep - nb, dtype=torch.bool, device=dev)])
    cur = P
    logits = engine.decode_multi(running[:, :, :P].reshape(B * nb, P), cache).view(B * nb, P, -1)[:, -1, :V]
    while True:
        lp = torch.log_softmax(logits.float(), dim=-1)
        flat = running[:, :, :cur].reshape(B * nb, cur)
        if cur - P < int(min_new_tokens):
            lp[:, eos] = float("-inf")
        if cur == P and bool(bsup.any()):
            lp = lp.masked_fill(bsup[None, :], float("-inf"))
        if bool(sup.any()):
            lp = lp.masked_fill(sup[None, :], float("-inf"))
        if timestamp_rules is not None:
            lp = apply_timestamp_rules(lp, flat, cur, timestamp_rules["begin_index"],
                                       timestamp_rules["no_timestamps_token_id"], eos,
                                       timestamp_rules.get("max_initial_timestamp_index"))
        acc = (lp.view(B, nb, V) + run_scores[:, :, None]).reshape(B, nb * V)
        top_lp, top_ix = torch.topk(acc, k=keep)
        src_beam, tok = top_ix // V, top_ix % V
        top_seq = gather(running, src_beam)
        top_seq[:, :, cur] = tok
        hits = (tok == eos) | (cur + 1 >= max_length)
        # open beams carried to the next step
        open_lp = top_lp + hits.float() * neg
        nxt = torch.topk(open_lp, k=nb)[1]
        running = gather(top_seq, nxt)
        run_scores = gather(open_lp, nxt)
        src_rows = (gather(src_beam, nxt) + torch.arange(B, device=dev)[:, None] * nb).reshape(-1)
        # finished hypotheses: only the best num_beams continuations may finish
        just = hits & top_mask[None, :]
        fin_lp = top_lp / float((cur + 1 - P) ** length_penalty)
        fin_lp = fin_lp + (finished.all(-1, keepdim=True) & (early_stopping is True)).float() * neg
        fin_lp = fin_lp + (~unsat).float() * neg
        fin_lp = fin_lp + (~just).float() * neg
        m_seq = torch.cat([sequences, top_seq], 1)
        m_sc = torch.cat([beam_scores, fin_lp], 1)
        m_fin = torch.cat([finished, just], 1)
        m_len = torch.cat([lengths, torch.full((B, keep), cur + 1 - P, dtype=torch.long, device=dev)], 1)
        best = torch.topk(m_sc, k=nb)[1]
        sequences, beam_scores, finished, lengths = gather(m_seq, best), gather(m_sc, best), gather(m_fin, best), gather(m_len, best)
        # the carried beams' K/V rows (positions < cur are live: the next pass appends position cur)
        for kvc in cache["self"]:
            v = kvc.view(B * nb, max_length, -1)
            v[:, :cur].copy_(v[src_rows, :cur])
        cur += 1
        # early-stopping heuristic and loop condition (TF `_check_early_stop_heuristic`, `_beam_search_has_unfinished_sequences`)
        hyp_len = (max_length - P) if (early_stopping == "never" and length_penalty > 0.0) else (cur - P)
        best_running = run_scores[:, :1] / float(hyp_len ** length_penalty)
        worst_fin = torch.where(finished, beam_scores.min(1, keepdim=True)[0], torch.full_like(beam_scores, neg))
        unsat = unsat & (best_running > worst_fin).any(-1, keepdim=True)
        go_on = bool(unsat.any()) and not (bool(finished.all()) and early_stopping is True) and not bool(hits.all())
        if not go_on:
            break
        logits = engine.decode_step(running[:, :, cur - 1].reshape(B * nb, 1).contiguous(), cache)[:, :V]
    out_len = P + int(lengths[:, 0].max().item())
    return sequences[:, 0, :out_len].contiguous()
