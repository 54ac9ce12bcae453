"""Distillation step on the MI355X engine: teacher forward + student forward/backward + gradient all-reduce + global
norm clip + AdamW -- the hot loop of run_distillation.py:1465-1495 and 1606-1614, one process per GPU.

Semantics kept from the reference (PyTorch path):
  * loss = 0.8 * CE + kl_weight * KL * T^2, CE = token mean over labels != -100, KL = sum / count(labels >= 0)
    (run_distillation.py:1453-1462, 1486-1493), each computed PER RANK; ranks are then averaged by the gradient
    all-reduce exactly like DDP does -- not the Flax path's global-token normalisation;
  * shared-encoder mode (`--freeze_encoder` with an identical teacher encoder, run_distillation.py:1046-1049,
    1473-1478): one encoder forward, teacher decoder inputs rebuilt from the labels by shift_tokens_right;
  * AdamW with two parameter groups (decay / no decay for biases and LayerNorm, 1386-1407), clip_grad_norm_(1.0).
Data parallelism: the flat fp32 gradient buffer is all-reduced with RCCL (torch.distributed "nccl" backend on ROCm)
in layer-ordered buckets on a side stream while the backward of earlier layers is still running.
"""
import numbers
import os
import queue
import sys
import threading
import time

import torch

from .engine import LiveRows, ParamStore, WhisperDims, WhisperEngine


def shift_tokens_right(labels, pad_token_id, decoder_start_token_id):
    """TF:modeling_whisper.py:68-81 (used by the shared-encoder teacher call, run_distillation.py:1478)."""
    out = labels.new_zeros(labels.shape)
    out[:, 1:] = labels[:, :-1]
    out[:, 0] = decoder_start_token_id
    return out.masked_fill(out == -100, pad_token_id)


class BucketWatchdog(threading.Thread):
    """A collective that never completes (a rank that died, a wedged xGMI link, mismatched bucket sizes) shows up as a
    silent hang at the next synchronisation, minutes later and far from its cause.  Every gradient bucket therefore leaves
    an event behind itself; this daemon thread polls them (`Event.query()`, no synchronisation, nothing on the enqueuing
    thread's path) and, when one is still pending `timeout_s` after it was issued, prints ONE line naming the step, the
    bucket and its element range and ends the process (exit code 124) -- the other ranks' own watchdogs then do the same.
    `on_timeout` replaces the exit (tests)."""

    def __init__(self, timeout_s, rank=0, on_timeout=None, poll_s=0.05, device=None, first_grace_s=None):
        super().__init__(daemon=True, name="grad-bucket-watchdog")
        self.timeout_s, self.rank, self.on_timeout, self.poll_s = float(timeout_s), rank, on_timeout, poll_s
        self.q = queue.Queue()
        self.fired = None
        # the device whose events this thread queries: a new thread's current device is 0, and Event.query() from a thread
        # that never set its device would run against (and create a context on) GPU 0 on every rank with local_rank != 0
        self.device = device
        # the very first bucket of a run also pays RCCL's lazy communicator set-up (rings, IPC handles): slow, not stuck
        self.first_grace_s = 3.0 * self.timeout_s if first_grace_s is None else float(first_grace_s)
        self._seen = 0
        self.errors = 0

    def submit(self, label, done):
        """done: a zero-argument callable that is True once the bucket has completed (Event.query for device buckets)."""
        self.q.put((time.monotonic(), label, done))

    def _done(self, done, label):
        """A failing poll must not end the thread silently (the watchdog would be off without anyone noticing): the error is
        logged once per bucket and the bucket is treated as still pending, so the deadline keeps applying."""
        try:
            return bool(done())
        except Exception as e:      # noqa: BLE001 -- anything the runtime raises from Event.query()
            self.errors += 1
            if self.errors <= 3:
                sys.stderr.write(f"[rank {self.rank}] gradient all-reduce watchdog: polling {label} raised "
                                 f"{type(e).__name__}: {e}\n")
                sys.stderr.flush()
            return False

    def run(self):
        if self.device is not None and torch.cuda.is_available() and torch.device(self.device).type == "cuda":
            torch.cuda.set_device(self.device)
        while True:
            item = self.q.get()
            if item is None:
                return
            t0, label, done = item
            limit = self.first_grace_s if self._seen == 0 else self.timeout_s
            self._seen += 1
            while not self._done(done, label):
                if time.monotonic() - t0 > limit:
                    msg = (f"[rank {self.rank}] gradient all-reduce watchdog: {label} still pending {limit:.0f} s after it "
                           f"was issued -- a peer rank is gone or the collective is wedged; ending this rank")
                    self.fired = msg
                    sys.stderr.write(msg + "\n")
                    sys.stderr.flush()
                    if self.on_timeout is not None:
                        self.on_timeout(msg)
                        break
                    os._exit(124)
                time.sleep(self.poll_s)

    def stop(self):
        self.q.put(None)


class GradReducer:
    """Bucketed all-reduce (sum) of ranges of the flat gradient buffer, issued on a communication stream as soon as a
    range is final.  With world_size 1 (or no process group) it is a no-op.  `watchdog_s` > 0: see BucketWatchdog."""

    def __init__(self, flat, group=None, bucket_bytes=256 << 20, always_reduce=False, watchdog_s=0.0, on_timeout=None):
        import torch.distributed as dist
        self.flat, self.group, self.dist = flat, group, dist
        self.initialized = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.initialized else 1
        # always_reduce: issue the collectives even with one rank (exercises the stream hand-off on a single GPU)
        self.active = self.world > 1 or (always_reduce and self.initialized)
        self.bucket_elems = bucket_bytes // 4
        self.pending_lo = self.pending_hi = None
        self.handles = []
        self.cuda = flat.is_cuda
        self.stream = torch.cuda.Stream(device=flat.device) if (self.cuda and self.active) else None
        self.also_wait = []
        self.labels = []
        self.step = 0
        self.watchdog = None
        self._bytes = self._buckets = 0                  # of the step being issued
        self.last_bytes = self.last_buckets = 0          # of the last completed step (what bench.py reports)
        if self.active and watchdog_s and watchdog_s > 0:
            rank = dist.get_rank(group) if self.initialized else 0
            self.watchdog = BucketWatchdog(watchdog_s, rank, on_timeout, device=flat.device if flat.is_cuda else None)
            self.watchdog.start()

    def _launch(self, lo, hi):
        view = self.flat[lo:hi]
        self._bytes += (hi - lo) * 4
        self._buckets += 1
        self.labels.append(f"step {self.step} bucket {len(self.handles)}: elements [{lo}, {hi}) of the flat gradient "
                           f"({(hi - lo) * 4 / 2**20:.0f} MiB)")
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream(self.flat.device))
            for w in self.also_wait:      # streams that also write gradients (the engine's weight-gradient stream)
                self.stream.wait_stream(w)
            with torch.cuda.stream(self.stream):
                self.handles.append(self.dist.all_reduce(view, op=self.dist.ReduceOp.SUM, group=self.group,
                                                         async_op=True))
        else:
            self.handles.append(self.dist.all_reduce(view, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))

    def ready(self, lo, hi):
        """Gradients in [lo, hi) are final.  Ranges must arrive adjacent and descending (backward order)."""
        if not self.active or hi <= lo:
            return
        if self.pending_lo is not None and hi == self.pending_lo:
            self.pending_lo = lo
        else:
            self.flush()
            self.pending_lo, self.pending_hi = lo, hi
        if self.pending_hi - self.pending_lo >= self.bucket_elems:
            self.flush()

    def reduce_small(self, t):
        """Sum a small device tensor over the ranks on the communication stream (joined by wait())."""
        if not self.active:
            return
        self.labels.append(f"step {self.step} small tensor {tuple(t.shape)} (optimizer-skip gate)")
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream(self.flat.device))
            with torch.cuda.stream(self.stream):
                self.handles.append(self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self.handles.append(self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))

    def flush(self):
        if self.pending_lo is not None:
            self._launch(self.pending_lo, self.pending_hi)
            self.pending_lo = self.pending_hi = None

    def wait(self):
        self.flush()
        # the handles complete in issue order on the communication stream: h.wait() makes that stream wait for the
        # collective (it does not block the host for device tensors); an event behind each one tells the watchdog WHICH
        # bucket is the first that never finished
        watch = self.watchdog is not None and self.stream is not None and not torch.cuda.is_current_stream_capturing()
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                for h, label in zip(self.handles, self.labels):
                    h.wait()
                    if watch:
                        ev = torch.cuda.Event()
                        ev.record(self.stream)
                        self.watchdog.submit(label, ev.query)
        else:
            for h in self.handles:
                h.wait()
        self.handles, self.labels = [], []
        self.step += 1
        self.last_bytes, self.last_buckets, self._bytes, self._buckets = self._bytes, self._buckets, 0, 0
        if self.stream is not None:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.stream)


def _as_int(v):
    """A host integer in any of its spellings (int, numpy integer, 0-d array / tensor) -> int; everything else unchanged."""
    if isinstance(v, numbers.Integral):
        return int(v)
    if v is not None and hasattr(v, "ndim") and v.ndim == 0:
        return int(v.item())
    return v


def trim_dead_positions(decoder_input_ids, labels, valid_len):
    """Decoder positions behind the LAST labelled position are dead: the decoder is causal, so they reach no labelled
    position; the cross-entropy ignores label -100 and the KL term is masked by `labels >= 0`
    (run_distillation.py:1453-1462, 1486-1493), so they add nothing to the loss and their rows of every gradient GEMM are
    zero.  The reference still computes them (its collator pads every batch to max_label_length = 448,
    run_distillation.py:405-478).  `valid_len` is HOST data, known from the label lengths before the batch goes to the
    device (collator.DataCollatorSpeechSeq2SeqWithPadding.report_valid_len):
      * an int (1 + the index of the last label != -100 over the whole batch): the decoders, the LM heads and the loss
        run over the first `valid_len` positions of every sequence;
      * a sequence of B ints (the same per sequence): additionally the frozen teacher's decoder, both LM heads and the
        loss run over the live rows of each sequence only (engine.LiveRows; the student's layers keep the trimmed
        rectangle -- their backward does);
      * None keeps every position.
    Same loss, same gradients in all three.  Returns (decoder_input_ids, labels, per-sequence lengths or None)."""
    if valid_len is None:
        return decoder_input_ids, labels, None
    T = decoder_input_ids.shape[1]
    lens = None
    valid_len = _as_int(valid_len)
    if not isinstance(valid_len, int):
        lens = [max(1, min(T, int(x))) for x in valid_len]
        if len(lens) != decoder_input_ids.shape[0]:
            raise ValueError("valid_len: one length per sequence of the batch (or one int for the batch)")
        valid_len = max(lens)
    Te = max(1, min(T, int(valid_len)))
    if Te < T:
        decoder_input_ids, labels = decoder_input_ids[:, :Te], labels[:, :Te]
    return decoder_input_ids, labels, lens


class DistillationTrainer:
    overwrite_wgrad = True    # the split-K combine of the layers' weight gradients stores into the cleared buffer instead of adding to it

    def __init__(self, ops, student_sd, student_dims, teacher_sd, teacher_dims, *, temperature=2.0, kl_weight=1.0,
                 lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0, freeze_encoder=False,
                 share_encoder=False, freeze_embed_positions=False, process_group=None, mel_filters=None,
                 overlap_teacher=False, overlap_wgrad=False, pad_teacher_rows=False, bucket_bytes=256 << 20,
                 always_reduce=False, comm_watchdog_s=0.0):
        self.ops = ops
        self.sdims, self.tdims = WhisperDims.from_any(student_dims), WhisperDims.from_any(teacher_dims)
        frozen = []
        if freeze_encoder:
            frozen.append("model.encoder.")
        if freeze_embed_positions:
            frozen.append("model.decoder.embed_positions.")
        self.student_store = ParamStore(ops, self.sdims, student_sd, trainable=True, frozen_prefixes=tuple(frozen))
        self.teacher_store = ParamStore(ops, self.tdims, teacher_sd, trainable=False, round_bf16=True)
        self.student = WhisperEngine(ops, self.student_store, torch.float32)
        self.teacher = WhisperEngine(ops, self.teacher_store, ops.lowp)
        self.teacher.pad_gemm_rows = bool(pad_teacher_rows)   # decoder GEMMs of the frozen teacher over M padded to 320 rows
        self.temperature, self.kl_weight = temperature, kl_weight
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.freeze_encoder, self.share_encoder = freeze_encoder, share_encoder
        if share_encoder and not freeze_encoder:
            raise ValueError("share_encoder requires freeze_encoder (run_distillation.py:1046-1049)")
        st = self.student_store
        # optimizer scalars live on the device (lr, step count, betas: include/dwamd.h dw_adam_tick), so that the whole
        # step is a fixed launch sequence (HIP-graph capturable) and an all-ignored batch can be skipped without a host sync
        self._adam = ops.adam_state(lr, betas[0], betas[1], 0)
        self._lr_dev = lr
        self._gate = None            # f32[1] on the device: labels counted in the micro-batches of the current step
        self._graph = None           # the captured whole-step HIP graph in use (train_step_graphed)
        self._graphs, self._graph_inputs = {}, None   # every captured plan by (shapes, live decoder positions); shared inputs
        self.reducer = GradReducer(st.G, process_group, bucket_bytes, always_reduce, watchdog_s=comm_watchdog_s) \
            if st.G is not None else None
        self.world = self.reducer.world if self.reducer else 1
        self.mel_filters = mel_filters
        # optional: the frozen teacher forward is independent of the student forward until the loss and can run on a
        # second HIP stream so that its kernels fill the tails of the student's (measured +0.6 % on 1 GPU; off by default
        # so that per-kernel profiles of the step stay one-kernel-at-a-time)
        self.overlap_teacher = overlap_teacher and self.student_store.P.is_cuda
        self._tstream = torch.cuda.Stream(device=self.student_store.P.device) if self.overlap_teacher else None
        # optional: the weight-gradient GEMMs of the student's backward on a second stream (engine._wgrad): they are off
        # the critical path, and a persistent GEMM of one stream takes the CUs the other stream's kernel leaves idle in
        # its last round of tiles
        self.set_overlap_wgrad(overlap_wgrad)
        self._sumsq = ops.zeros((1,), torch.float32)
        self._accum = 1            # micro-batches summed in the gradient buffer of the current optimizer step
        self._last_gm = 1.0        # gradient multiplier the last optimizer step applied (1 / (world * accum))
        self.segments = st.adam_segments(weight_decay)

    def set_overlap_wgrad(self, on):
        on = bool(on) and self.student_store.P.is_cuda
        ws = torch.cuda.Stream(device=self.student_store.P.device) if on else None
        self.student.join_wgrad_stream()
        self.student.wgrad_stream = ws
        if self.reducer is not None:
            self.reducer.also_wait = [ws] if ws is not None else []

    # ------------------------------------------------------------------------------------------------------------
    def features(self, audio):
        """fp32 waveforms [B, 480000] on the device -> log-mel input_features [B, n_mels, 3000]."""
        return self.ops.logmel(audio, self.mel_filters)

    pack_live_rows_below = 0.9      # per-sequence lengths are used when the live rows are less than this share of B x T

    def _live_rows(self, lens, B, Te, device):
        if lens is None or sum(lens) >= self.pack_live_rows_below * B * Te:
            return None
        return LiveRows.build(lens, Te, device)

    def forward_backward(self, input_features, decoder_input_ids, labels, zero_grad=True, sync_grads=True, valid_len=None,
                         _live=None):
        """One micro-batch: returns losses fp32[4] = (ce, kl, loss, n_valid) on the device (no host sync).
        zero_grad=False accumulates onto the gradients already in the flat buffer and sync_grads=False skips the
        all-reduce (gradient accumulation, `accelerator.accumulate` / DDP no_sync of run_distillation.py:1607).
        `valid_len`: see trim_dead_positions."""
        ops, S, T = self.ops, self.student, self.teacher
        decoder_input_ids, labels, lens = trim_dead_positions(decoder_input_ids, labels, valid_len)
        B, Td = decoder_input_ids.shape
        live = _live if _live is not None else self._live_rows(lens, B, Td, decoder_input_ids.device)
        input_features = input_features.to(torch.float32).contiguous()
        decoder_input_ids = decoder_input_ids.contiguous()
        labels_flat = labels.reshape(-1).contiguous()
        if live is not None:
            labels_flat = labels_flat.index_select(0, live.idx)      # labels of the packed rows
        if self.overlap_teacher and self._tstream is None:
            self._tstream = torch.cuda.Stream(device=self.student_store.P.device)
        side = self._tstream if (self.overlap_teacher and not self.share_encoder) else None
        if side is not None:
            main = torch.cuda.current_stream(input_features.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                enc_t, _ = T.encode(input_features, save=False)
                logits_t, _ = T.decode(decoder_input_ids, enc_t, save=False, live=live)
                del enc_t
            # (no record_stream: the inputs are the caller's and stay alive over the call, the side stream is joined
            # below, and logits_t -- a block of the side stream's pool -- is next written by the teacher forward of the
            # following step, which starts with side.wait_stream(main), i.e. after the loss kernel that reads it)
        enc_s, ectx = S.encode(input_features, save=not self.freeze_encoder)
        logits_s, dctx = S.decode(decoder_input_ids, enc_s, save=True, live=live)
        if side is not None:
            main.wait_stream(side)
        elif self.share_encoder:
            t_ids = shift_tokens_right(labels, self.tdims.pad_token_id, self.tdims.decoder_start_token_id)
            logits_t, _ = T.decode(t_ids, enc_s, save=False, live=live)
        else:
            enc_t, _ = T.encode(input_features, save=False)
            logits_t, _ = T.decode(decoder_input_ids, enc_t, save=False, live=live)
            del enc_t
        R = B * Td if live is None else live.n
        losses = ops.distill_loss(logits_s[:R], logits_t[:R], labels_flat, self.sdims.vocab, self.temperature, 0.8,
                                  self.kl_weight, 1.0, True)
        del logits_t
        self._gate = losses[3:4] if (zero_grad or self._gate is None) else self._gate + losses[3:4]
        if zero_grad:
            S.zero_small_grads(skip_weights=self.overwrite_wgrad)
        st = self.student_store
        dp = self.reducer is not None and self.reducer.active and sync_grads
        S.wgrad_overwrite = bool(zero_grad) and self.overwrite_wgrad   # (the buffer was just cleared: the layers' weight gradients are stored, not added)
        try:
            denc = S.backward_decoder(dctx, logits_s, want_denc=not self.freeze_encoder)
            del logits_s, dctx
            if dp:
                self.reducer.ready(st.train_start if self.freeze_encoder else st.dec_start, st.train_end)
            if not self.freeze_encoder:
                S.backward_encoder(ectx, denc, on_ready=self.reducer.ready if dp else None)
        finally:
            S.wgrad_overwrite = False
        S.join_wgrad_stream()
        return losses

    def eval_step(self, input_features, decoder_input_ids, labels, valid_len=None):
        """Forward-only CE + KL of the reference's `eval_step` (run_distillation.py:1498-1522: both models in eval
        mode under no_grad, the same loss mix, "temperature is always 1 for eval"): no activation is kept, nothing is
        written to the gradient buffer, the student logits are left intact.  Returns losses fp32[4] = (ce, kl, loss,
        n_valid) on the device."""
        ops, S, T = self.ops, self.student, self.teacher
        decoder_input_ids, labels, lens = trim_dead_positions(decoder_input_ids, labels, valid_len)
        B, Td = decoder_input_ids.shape
        live = self._live_rows(lens, B, Td, decoder_input_ids.device)
        input_features = input_features.to(torch.float32).contiguous()
        decoder_input_ids = decoder_input_ids.contiguous()
        enc_s, _ = S.encode(input_features, save=False)
        logits_s, _ = S.decode(decoder_input_ids, enc_s, save=False, live=live)
        if self.share_encoder:
            t_ids = shift_tokens_right(labels, self.tdims.pad_token_id, self.tdims.decoder_start_token_id)
            logits_t, _ = T.decode(t_ids, enc_s, save=False, live=live)
        else:
            enc_t, _ = T.encode(input_features, save=False)
            logits_t, _ = T.decode(decoder_input_ids, enc_t, save=False, live=live)
        R = B * Td if live is None else live.n
        labels_flat = labels.reshape(-1).contiguous()
        if live is not None:
            labels_flat = labels_flat.index_select(0, live.idx)
        return ops.distill_loss(logits_s[:R], logits_t[:R], labels_flat, self.sdims.vocab,
                                1.0, 0.8, self.kl_weight, 1.0, False)

    def set_lr(self, lr):
        """Learning rate of the next optimizer step(s) (the reference steps `get_scheduler`'s LambdaLR once per step,
        run_distillation.py:1410-1415, 1613): written to the device-resident optimizer state when it changed."""
        if lr is not None and lr != self._lr_dev:
            self._adam[0:1].fill_(lr)
            self._lr_dev = lr

    @property
    def step_count(self):
        """Optimizer steps applied so far (device-resident: a step on a batch without labels is skipped there)."""
        return int(self._adam[1].item())

    @step_count.setter
    def step_count(self, n):
        self._adam[1:2].fill_(float(n))

    def optimizer_step(self, lr=None, _write_lr=True):
        """clip_grad_norm_ + AdamW over the flat buffers; refreshes the bf16 shadow weights (fused) and the packed
        conv weights.  The step is skipped ON THE DEVICE when the micro-batches of this step held no label at all
        (the reference's loss is 0/0 = NaN there and poisons every parameter, run_distillation.py:1486-1493; with zero
        gradients AdamW would still move the weights by momentum and weight decay)."""
        ops, st = self.ops, self.student_store
        if self.reducer is not None:
            if self.reducer.active and self._gate is not None:
                # The gradients every rank applies are the reduced ones, so the skip decision has to be global too: a rank
                # whose own shard held no label would otherwise freeze its replica (parameters, moments, step count) while
                # the others step -- replicas diverge silently.  The label counts are summed with the gradient buckets.
                self._gate = self._gate.clone()
                self.reducer.reduce_small(self._gate)
            self.reducer.wait()
        if _write_lr:
            self.set_lr(self.lr if lr is None else lr)
        lo, hi = st.train_start, st.train_end
        gm = 1.0 / (self.world * self._accum)
        self._last_gm = gm
        self._sumsq.zero_()
        ops.sumsq(st.G[lo:hi], self._sumsq)
        ops.adam_tick(self._adam, self._gate)
        for a, b, wd in self.segments:
            ops.adamw_dev(st.P[a:b], st.G[a:b], st.M[a:b], st.V[a:b], st.S[a:b], self._sumsq, self.max_grad_norm, gm,
                          self._adam, self.eps, wd)
        if not self.freeze_encoder:
            st.repack_conv()
        self._gate = None

    def train_step(self, input_features, decoder_input_ids, labels, lr=None, valid_len=None):
        losses = self.forward_backward(input_features, decoder_input_ids, labels, valid_len=valid_len)
        self.optimizer_step(lr)
        return losses

    # ---- the whole step as ONE captured HIP graph --------------------------------------------------------------
    def train_step_graphed(self, inputs, decoder_input_ids, labels, lr=None, eager_steps=2, valid_len=None):
        """`train_step` (preceded by the log-mel front end when `inputs` are waveforms [B, n_samples] instead of
        features [B, n_mels, 3000]) replayed from one HIP graph.  The shapes of a step are static, so its ~2 700
        launches on three streams (main, frozen teacher, weight gradients), every buffer address and the cross-stream
        dependencies are planned ONCE: the first `eager_steps` calls run eagerly on the capture stream (first-use
        initialisation of the library and the engine), the next call captures the step -- torch's allocator serves the
        capture from a private pool, which is the static activation arena of this (B, T): lifetimes are the Python
        lifetimes of one step, no allocator call and no host-side stream bookkeeping remain afterwards -- and every
        later call is three small device copies into the static input buffers plus one graph launch.  Every call
        performs exactly one optimizer step; the returned losses tensor is static (overwritten by the next call).
        `valid_len` (trim_dead_positions) is part of the plan: one graph per (live decoder positions, packed rows), at most
        `max_graphs` of them, all replaying out of the SAME pool (they never run concurrently) and the same static input
        buffers.  Data-parallel runs keep the eager path (`train_step`): the bucketed RCCL all-reduce is issued between
        the backward's layers from the host."""
        if self.reducer is not None and self.reducer.active:
            raise RuntimeError("train_step_graphed: data-parallel steps run eagerly (RCCL buckets are issued from the host)")
        dev = self.student_store.P.device
        B, T = decoder_input_ids.shape
        lens = None
        valid_len = _as_int(valid_len)
        if valid_len is not None and not isinstance(valid_len, int):
            lens = [max(1, min(T, int(x))) for x in valid_len]
            valid_len = max(lens)
        Te = T if valid_len is None else max(1, min(T, int(valid_len)))
        # Plans are keyed by QUANTISED sizes: the live positions rounded up to a multiple of `plan_pos_quantum`, the packed
        # rows to a multiple of `plan_row_quantum` (the list is filled up with dead rows of the rectangle: label -100, no
        # contribution).  With the label lengths of real batches nearly every batch has its own (max, sum); exact keys
        # would mean a capture per batch and constant eviction among the `max_graphs` plans.
        qp, qr = self.plan_pos_quantum, self.plan_row_quantum
        Te = min(T, -(-Te // qp) * qp) if (qp > 1 and valid_len is not None) else Te
        if lens is not None and sum(lens) >= self.pack_live_rows_below * B * Te:
            lens = None
        Rc = sum(lens) if lens is not None else 0           # packed rows (0: the trimmed rectangle)
        if lens is not None and qr > 1:
            Rc = min(B * Te, -(-Rc // qr) * qr)
            if Rc >= self.pack_live_rows_below * B * Te:        # (filled up, the list is nearly the rectangle: keep the rectangle)
                lens, Rc = None, 0
        in_key = (tuple(inputs.shape), inputs.dtype, tuple(decoder_input_ids.shape), self.overlap_teacher,
                  self.student.wgrad_stream is not None,
                  # (a captured plan has these baked into its launches)
                  self.temperature, self.kl_weight, self.max_grad_norm, self.eps, self.weight_decay, self._accum)
        key = in_key + (Te, Rc)
        ins = self._graph_inputs
        if ins is None or ins["key"] != in_key:
            self._graphs.clear()
            self._graph = None
            ins = self._graph_inputs = {"key": in_key, "stream": torch.cuda.Stream(device=dev), "pool": None,
                                        "x": torch.empty_like(inputs), "ids": torch.empty_like(decoder_input_ids),
                                        "labels": torch.empty_like(labels),
                                        "live_idx": torch.zeros(B * T, dtype=torch.int32, device=dev),
                                        # ragged-batch table of the packed rows (LiveRows.seq_table): B sequences + filler entries
                                        "seq_start": torch.zeros(B + max(1, qr) + 1, dtype=torch.int32, device=dev),
                                        "seq_len": torch.zeros(B + max(1, qr) + 1, dtype=torch.int32, device=dev)}
        g = self._graph
        if g is None or g["key"] != key:
            g = self._graphs.get(key)
            if g is None:
                while len(self._graphs) >= self.max_graphs:
                    self._graphs.pop(next(iter(self._graphs)))
                g = self._graphs[key] = {"key": key, "calls": 0, "graph": None, "losses": None}
            self._graph = g
        self.set_lr(self.lr if lr is None else lr)
        ins["x"].copy_(inputs)
        ins["ids"].copy_(decoder_input_ids)
        ins["labels"].copy_(labels)
        live = None
        if lens is not None:             # the row list of this batch goes into the plan's static index buffer
            ins["live_idx"][:Rc].copy_(LiveRows.host_index(lens, Te, fill_to=Rc))
            E = B + -(-max(1, qr) // Te)           # table entries of this plan (fixed: the launch's sequence count is captured)
            st, ln = LiveRows.seq_table(lens, Te, fill_to=Rc, entries=E)
            ins["seq_start"][:E].copy_(st)
            ins["seq_len"][:E].copy_(ln)
            live = LiveRows(ins["live_idx"][:Rc], Rc, B, Te, ins["seq_start"][:E], ins["seq_len"][:E], Te)

        def body():
            feats = self.features(ins["x"]) if ins["x"].dim() == 2 else ins["x"]
            losses = self.forward_backward(feats, ins["ids"], ins["labels"], valid_len=Te, _live=live)
            self.optimizer_step(_write_lr=False)
            return losses

        if g["graph"] is not None:
            g["graph"].replay()
        elif g["calls"] < eager_steps:
            cur, cs = torch.cuda.current_stream(dev), ins["stream"]
            cs.wait_stream(cur)
            with torch.cuda.stream(cs):
                g["losses"] = body()
            cur.wait_stream(cs)
        else:
            if self.ops.profile is not None:
                raise RuntimeError("train_step_graphed: per-launch profiling events cannot be captured")
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, pool=ins["pool"], stream=ins["stream"]):
                g["losses"] = body()
            if ins["pool"] is None:
                ins["pool"] = graph.pool()
            g["graph"] = graph
            graph.replay()
        g["calls"] += 1
        return g["losses"]

    max_graphs = 8    # captured steps kept (one per number of live decoder positions; the oldest plan is dropped)
    plan_pos_quantum = 32     # live decoder positions of a plan: multiples of this
    plan_row_quantum = 320    # packed rows of a plan: multiples of this (one row tile of the 320-row GEMM kernel)

    def drop_graph(self):
        """Forget every captured step (their shared memory pool is released with them)."""
        self._graph = None
        self._graphs.clear()
        self._graph_inputs = None

    def train_step_accumulated(self, micro_batches, lr=None):
        """Gradient accumulation over a list of (input_features, decoder_input_ids, labels): gradients are summed in
        the flat buffer, all-reduced once (with the last micro-batch) and averaged over the micro-batches in the fused
        AdamW, like accelerate's `accumulate` context (loss / gradient_accumulation_steps)."""
        n = len(micro_batches)
        out = []
        self._accum = n
        try:
            for i, mb in enumerate(micro_batches):      # (input_features, decoder_input_ids, labels[, valid_len])
                out.append(self.forward_backward(mb[0], mb[1], mb[2], zero_grad=(i == 0), sync_grads=(i == n - 1),
                                                 valid_len=mb[3] if len(mb) > 3 else None))
            self.optimizer_step(lr)
        finally:
            self._accum = 1
        return torch.stack(out).nanmean(0)     # (a micro-batch without labels reports NaN losses; it adds no gradient)

    def grad_norm(self):
        """Global gradient norm the last optimizer step clipped (what `accelerator.clip_grad_norm_` returns,
        run_distillation.py:1611): norm of the rank- and micro-batch-averaged gradient."""
        return torch.sqrt(self._sumsq[0]) * self._last_gm

    # ---- checkpoint / resume (the reference: accelerator.save_state / load_state, run_distillation.py:1638-1650) ----
    def state_dict(self):
        """Everything a resumed run needs to continue bit-exactly: fp32 master weights, Adam moments (HF parameter
        names), step count and hyper-parameters."""
        st = self.student_store
        names = [n for n in st.real_names() if st.is_trainable(n)]

        def view(buf, name):
            o, shape, _ = st.entries[name]
            n = 1
            for d in shape:
                n *= d
            return buf[o:o + n].view(shape).detach().clone()
        return {"model": st.state_dict(), "exp_avg": {n: view(st.M, n) for n in names},
                "exp_avg_sq": {n: view(st.V, n) for n in names}, "step": self.step_count,
                "hyper": {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
                          "max_grad_norm": self.max_grad_norm, "temperature": self.temperature,
                          "kl_weight": self.kl_weight}}

    def load_state_dict(self, state):
        st = self.student_store
        st.load_state_dict(state["model"])
        for key, buf in (("exp_avg", st.M), ("exp_avg_sq", st.V)):
            for n, t in state[key].items():
                o, shape, _ = st.entries[n]
                buf[o:o + t.numel()].view(shape).copy_(t.to(buf.device))
        h = state.get("hyper", {})
        self.lr, self.eps = h.get("lr", self.lr), h.get("eps", self.eps)
        self.betas = tuple(h.get("betas", self.betas))
        self.temperature, self.kl_weight = h.get("temperature", self.temperature), h.get("kl_weight", self.kl_weight)
        self.max_grad_norm = h.get("max_grad_norm", self.max_grad_norm)
        self.weight_decay = h.get("weight_decay", self.weight_decay)
        self.segments = st.adam_segments(self.weight_decay)       # per-range weight decay follows the restored value
        self._adam = self.ops.adam_state(self.lr, self.betas[0], self.betas[1], int(state["step"]))
        self._lr_dev = self.lr
        self.drop_graph()         # (a captured step has the old hyper-parameters baked in: re-planned on next use)
