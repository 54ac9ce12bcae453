"""Host engine of the MI355X Whisper distillation path: parameter store, hand-written forward and backward of the
Whisper encoder-decoder over the `ops` interface (distil_whisper_amd.ops_hip.HipOps -> libdwamd.so HIP kernels).

What it replaces in the reference: WhisperForConditionalGeneration.forward + autograd backward
(TF:modeling_whisper.py:592-646 encoder, 690-795 decoder, 994-1099 LM head) as driven by `train_step`
(run_distillation.py:1465-1495).  The dtype flow reproduces the reference under `--dtype bfloat16`
(SURVEY.md section 8 a'): student = fp32 master weights, bf16 GEMM/attention operands, fp32 residual stream and LayerNorm;
teacher = bf16 weights and bf16 residual stream.  Activations kept for backward are exactly the ones autograd would
keep (no recompute); weight gradients are produced in fp32 straight into one flat gradient buffer so that the
data-parallel all-reduce and the fused clip+AdamW run over contiguous memory.

No autograd, no nn.Module here: modeling.py wraps this engine behind the reference's module surface.
"""
from dataclasses import dataclass

import ctypes as C

import torch


@dataclass
class WhisperDims:
    d_model: int
    heads: int
    ffn: int
    enc_layers: int
    dec_layers: int
    vocab: int
    n_mels: int
    max_src: int = 1500
    max_tgt: int = 448
    pad_token_id: int = 50256
    decoder_start_token_id: int = 50257

    @staticmethod
    def from_any(c):
        """Accepts a transformers WhisperConfig, an oracle OracleConfig or a WhisperDims."""
        if isinstance(c, WhisperDims):
            return c
        if hasattr(c, "encoder_attention_heads"):
            return WhisperDims(c.d_model, c.encoder_attention_heads, c.encoder_ffn_dim, c.encoder_layers,
                               c.decoder_layers, c.vocab_size, c.num_mel_bins, c.max_source_positions,
                               c.max_target_positions, c.pad_token_id, c.decoder_start_token_id)
        return WhisperDims(c.d_model, c.heads, c.ffn, c.enc_layers, c.dec_layers, c.vocab, c.n_mels, c.max_src,
                           c.max_tgt, c.pad_token_id, c.decoder_start_token_id)


def _rup(x, m):
    return (x + m - 1) // m * m


def layer_names(prefix, cross):
    blocks = ["self_attn", "encoder_attn"] if cross else ["self_attn"]
    out = []
    for b in blocks:
        out += [(f"{prefix}.{b}.q_proj.weight", "w"), (f"{prefix}.{b}.k_proj.weight", "w"),
                (f"{prefix}.{b}.v_proj.weight", "w"), (f"{prefix}.{b}.q_proj.bias", "b"),
                (f"{prefix}.{b}.k_proj.bias", "zero"), (f"{prefix}.{b}.v_proj.bias", "b"),
                (f"{prefix}.{b}.out_proj.weight", "w"), (f"{prefix}.{b}.out_proj.bias", "b"),
                (f"{prefix}.{b}_layer_norm.weight", "ln"), (f"{prefix}.{b}_layer_norm.bias", "ln")]
    out += [(f"{prefix}.fc1.weight", "w"), (f"{prefix}.fc1.bias", "b"), (f"{prefix}.fc2.weight", "w"),
            (f"{prefix}.fc2.bias", "b"), (f"{prefix}.final_layer_norm.weight", "ln"),
            (f"{prefix}.final_layer_norm.bias", "ln")]
    return out


class ParamStore:
    """All parameters of one model in flat device buffers.

    p      fp32 master weights (HF names -> views; `k_proj.bias` are zero dummies that keep q/k/v biases contiguous)
    s      bf16 shadow of p with identical offsets (GEMM operands); q/k/v weights are adjacent, so the fused
           [3D, D] QKV weight is a plain view
    g,m,v  fp32 gradient and Adam moments (trainable models only), same offsets
    Frozen tensors come first, trainable ones after, so gradient all-reduce / clip / AdamW see one contiguous range.
    """

    def __init__(self, ops, dims: WhisperDims, state_dict, trainable=False, frozen_prefixes=(), round_bf16=False):
        self.ops, self.dims, self.trainable = ops, dims, trainable
        D, Fd = dims.d_model, dims.ffn
        spec = [("model.encoder.conv1.weight", (D, dims.n_mels, 3), "conv"), ("model.encoder.conv1.bias", (D,), "b"),
                ("model.encoder.conv2.weight", (D, D, 3), "conv"), ("model.encoder.conv2.bias", (D,), "b"),
                ("model.encoder.embed_positions.weight", (dims.max_src, D), "frozen")]

        def shapes(name, kind):
            if kind == "w":
                if name.endswith("fc1.weight"):
                    return (Fd, D)
                if name.endswith("fc2.weight"):
                    return (D, Fd)
                return (D, D)
            if name.endswith("fc1.bias"):
                return (Fd,)
            return (D,)

        for i in range(dims.enc_layers):
            spec += [(n, shapes(n, k), k) for n, k in layer_names(f"model.encoder.layers.{i}", False)]
        spec += [("model.encoder.layer_norm.weight", (D,), "ln"), ("model.encoder.layer_norm.bias", (D,), "ln"),
                 ("model.decoder.embed_positions.weight", (dims.max_tgt, D), "emb")]
        for i in range(dims.dec_layers):
            spec += [(n, shapes(n, k), k) for n, k in layer_names(f"model.decoder.layers.{i}", True)]
        spec += [("model.decoder.layer_norm.weight", (D,), "ln"), ("model.decoder.layer_norm.bias", (D,), "ln"),
                 ("model.decoder.embed_tokens.weight", (dims.vocab, D), "emb")]

        def is_frozen(name, kind):
            if kind == "frozen" or not trainable:
                return True
            return any(name.startswith(pf) for pf in frozen_prefixes)

        ordered = [e for e in spec if is_frozen(e[0], e[2])] + [e for e in spec if not is_frozen(e[0], e[2])]
        self.entries, off = {}, 0
        self.train_start = None
        for name, shape, kind in ordered:
            if self.train_start is None and not is_frozen(name, kind):
                self.train_start = off
            n = 1
            for d in shape:
                n *= d
            self.entries[name] = (off, shape, kind)
            off += _rup(n, 64)
        self.train_end = off
        if self.train_start is None:
            self.train_start = off
        dec_offs = [o for n, (o, _, _) in self.entries.items() if n.startswith("model.decoder.") and o >= self.train_start]
        self.dec_start = min(dec_offs) if dec_offs else self.train_end  # first trainable decoder-side element
        total = off + 64 * max(D, 64)  # slack: the LM-head dX GEMM over-reads up to 63 rows past embed_tokens
        self.total = total
        self.P = ops.zeros((total,), torch.float32)
        self.S = ops.zeros((total,), ops.lowp)
        self.G = self.M = self.V = None
        if trainable and self.train_end > self.train_start:
            self.G = ops.zeros((total,), torch.float32)
            self.M = ops.zeros((total,), torch.float32)
            self.V = ops.zeros((total,), torch.float32)
        self.p, self.s, self.g = {}, {}, {}
        for name, (o, shape, kind) in self.entries.items():
            n = 1
            for d in shape:
                n *= d
            self.p[name] = self.P[o:o + n].view(shape)
            self.s[name] = self.S[o:o + n].view(shape)
            if self.G is not None and o >= self.train_start:
                self.g[name] = self.G[o:o + n].view(shape)
        # Ranges of G that are ACCUMULATED into (atomics / += : biases, LayerNorm parameters, embeddings, the convolution
        # weights' unpack) as opposed to the layers' weight matrices, which their weight-gradient GEMMs can store outright
        # (engine.wgrad_overwrite): merged into contiguous views, cleared by one multi-tensor fill instead of a 3 GB one.
        self.small_grad_views = []
        if self.G is not None:
            rng = []
            for name, (o, shape, kind) in self.entries.items():
                if o < self.train_start or kind == "w":
                    continue
                n = 1
                for d in shape:
                    n *= d
                if rng and rng[-1][1] == o:
                    rng[-1][1] = o + _rup(n, 64)
                else:
                    rng.append([o, o + _rup(n, 64)])
            self.small_grad_views = [self.G[a:b] for a, b in rng]
        self.kpad1 = _rup(3 * dims.n_mels, 64)
        self.conv1_packed = ops.zeros((D, self.kpad1), ops.lowp)
        self.conv2_packed = ops.zeros((D, 3 * D), ops.lowp)
        self.load_state_dict(state_dict, round_bf16=round_bf16)

    # ------------------------------------------------------------------------------------------------------------
    def real_names(self):
        return [n for n, (_, _, k) in self.entries.items() if k != "zero"]

    def load_state_dict(self, sd, round_bf16=False):
        dev = self.P.device
        for name in self.real_names():
            src = sd[name].detach().to(device=dev, dtype=torch.float32)
            if round_bf16:
                src = src.to(torch.bfloat16).to(torch.float32)
            self.p[name].copy_(src)
        self.refresh_shadow()

    def state_dict(self):
        sd = {n: self.p[n].detach().clone() for n in self.real_names()}
        sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
        return sd

    def refresh_shadow(self):
        """bf16 shadow <- master (whole buffer) and conv weights -> GEMM layout."""
        self.ops.cast_bf16(self.P, out=self.S)
        self.repack_conv()

    def repack_conv(self):
        self.ops.pack_conv_weight(self.p["model.encoder.conv1.weight"], self.kpad1, out=self.conv1_packed)
        self.ops.pack_conv_weight(self.p["model.encoder.conv2.weight"], 3 * self.dims.d_model, out=self.conv2_packed)

    def is_trainable(self, name):
        return self.G is not None and self.entries[name][0] >= self.train_start

    # fused views ------------------------------------------------------------------------------------------------
    def attn_views(self, prefix, cross_kv=False):
        """Views for one attention block: fused QKV (self) or Q + fused KV (cross)."""
        D = self.dims.d_model
        oq = self.entries[f"{prefix}.q_proj.weight"][0]
        ob = self.entries[f"{prefix}.q_proj.bias"][0]
        assert self.entries[f"{prefix}.k_proj.weight"][0] == oq + D * D
        assert self.entries[f"{prefix}.v_proj.bias"][0] == ob + 2 * D
        v = {"wqkv": self.S[oq:oq + 3 * D * D].view(3 * D, D), "bqkv": self.P[ob:ob + 3 * D],
             "wo": self.s[f"{prefix}.out_proj.weight"], "bo": self.p[f"{prefix}.out_proj.bias"]}
        if self.is_trainable(f"{prefix}.q_proj.weight"):
            v["g_wqkv"] = self.G[oq:oq + 3 * D * D].view(3 * D, D)
            v["g_bqkv"] = self.G[ob:ob + 3 * D]
            v["g_wo"] = self.g[f"{prefix}.out_proj.weight"]
            v["g_bo"] = self.g[f"{prefix}.out_proj.bias"]
        return v

    def adam_segments(self, weight_decay, decay_filter=None):
        """Contiguous [start, end, wd) ranges of the trainable region (run_distillation.py:1386-1407: no decay for
        LayerNorm parameters and biases)."""
        segs = []
        for name, (o, shape, kind) in self.entries.items():
            if o < self.train_start:
                continue
            n = 1
            for d in shape:
                n *= d
            wd = weight_decay if kind in ("w", "conv", "emb") else 0.0
            end = o + _rup(n, 64)
            if segs and segs[-1][2] == wd and segs[-1][1] == o:
                segs[-1] = (segs[-1][0], end, wd)
            else:
                segs.append((o, end, wd))
        return segs


class LiveRows:
    """The live rows of a [B, T] decoder batch: position t of sequence b is live when t < lens[b] (lens[b] = 1 + the
    last labelled position of the sequence; everything behind it reaches no label through the causal decoder).  `idx`
    (int32 on the device) lists their row numbers b*T + t, sequence by sequence.  Row-local work (projections,
    LayerNorm, MLP, LM head, loss) runs over the n packed rows; attention gets the (batch, position) layout back through
    one scatter / gather pair (ops.scatter_rows / gather_rows)."""

    def __init__(self, idx, n, B, T, seq_start=None, seq_len=None, max_q=0, attn_flops=(0.0, 0.0)):
        self.idx, self.n, self.B, self.T = idx, int(n), int(B), int(T)
        # ragged-batch table of the packed rows (int32 on the device, LiveRows.seq_table): with it attention reads the packed rows
        # in place (ops.attn_fwd_varlen) instead of going through the (batch, position) rectangle
        self.seq_start, self.seq_len, self.max_q = seq_start, seq_len, int(max_q)
        self.attn_flops = attn_flops          # (sum of len^2, sum of len) over the sequences: flop counts of the per-launch profile

    @staticmethod
    def seq_table(lens, T, fill_to=None, entries=None):
        """(starts, lengths) of the packed rows host_index(lens, T, fill_to) lists: one entry per sequence, then the filler rows cut
        into pseudo-sequences of at most T rows (their attention results are never used, they only have to be finite), then
        zero-length entries up to `entries` (a captured plan's table has a fixed size: B + ceil(row quantum / T))."""
        lens = [max(1, min(int(T), int(x))) for x in lens]
        n_live = sum(lens)
        fill = max(0, int(fill_to) - n_live) if fill_to is not None else 0
        lens = lens + [min(T, fill - o) for o in range(0, fill, T)]
        if entries is not None:
            if len(lens) > entries:
                raise ValueError("LiveRows.seq_table: more filler sequences than the table holds")
            lens = lens + [0] * (entries - len(lens))
        ln = torch.tensor(lens, dtype=torch.int32)
        st = torch.cumsum(ln, 0, dtype=torch.int32) - ln
        return st, ln

    @staticmethod
    def host_index(lens, T, fill_to=None):
        """Row numbers of the live positions, sequence by sequence.  fill_to: pad the list up to that many rows with
        DEAD rows of the rectangle (distinct positions behind their sequence's last label: label -100, their own causal
        context only) -- captured plans are keyed by a quantised row count (DistillationTrainer.plan_row_quantum)."""
        lens = [max(1, min(int(T), int(x))) for x in lens]
        idx = torch.cat([b * T + torch.arange(n, dtype=torch.int32) for b, n in enumerate(lens)])
        if fill_to is not None and fill_to > idx.numel():
            need = int(fill_to) - idx.numel()
            dead = torch.cat([b * T + torch.arange(n, T, dtype=torch.int32) for b, n in enumerate(lens)])
            if dead.numel() < need:
                raise ValueError("LiveRows.host_index: not enough dead rows to fill the list")
            idx = torch.cat([idx, dead[:need]])
        return idx

    @classmethod
    def build(cls, lens, T, device):
        h = cls.host_index(lens, T)
        st, ln = cls.seq_table(lens, T)
        lf = ln.double()
        return cls(h.to(device), h.numel(), len(lens), T, st.to(device), ln.to(device), int(ln.max()),
                   attn_flops=(float((lf * lf).sum()), float(lf.sum())))


class WhisperEngine:
    """Forward/backward of one Whisper model over a ParamStore.  `stream` is the residual-stream dtype: fp32 for the
    student (autocast keeps residual adds and LayerNorm in fp32), the low-precision dtype for the bf16 teacher."""

    def __init__(self, ops, store: ParamStore, stream_dtype=torch.float32):
        self.ops, self.st, self.dims = ops, store, store.dims
        self.stream = stream_dtype
        self.lowp = ops.lowp
        self.ldv = _rup(self.dims.vocab, 64)
        self.wgrad_stream = None   # torch.cuda.Stream: weight-gradient GEMMs / bias column sums of the backward go there
        self._wgrad_held, self._wgrad_fences = [], []
        # Forward-only decoder passes (the frozen teacher) may run their GEMMs over a row count padded to a multiple of
        # 320 when that costs <= 1/32 extra rows: M = 32 x 447 = 14304 -> 14400 = 45 row tiles of the 320 x 256 GEMM
        # kernel, ONE round of 225 workgroups for N = 1280 where the 128-tile kernel needs 4.4 rounds of 1120.  Pad rows
        # hold garbage; every operation between the embedding and the logits is row-local (GEMM rows, LayerNorm rows,
        # attention indexed by (batch, position) over the first B*T rows), so they never reach a valid row.
        self.pad_gemm_rows = False
        self.pad_gemm_rows_min = 2560       # smallest B*T worth padding
        self.pad_gemm_rows_slack = 1 / 32   # most extra rows accepted, as a fraction of B*T
        self.pad_lm_rows = True             # training passes: zero pad rows behind hf / logits for the LM-head backward

    # ---- helpers -------------------------------------------------------------------------------------------------
    # Row pitch of the wide activation buffers (round 5, tools/gemm_stride_probe.py): a [48000 x 5120] bf16 operand has 10 240-byte
    # rows, and a GEMM tile reads 128 bytes of each of 256-320 consecutive rows -- addresses 10 KiB apart fall on two of an XCD's
    # sixteen L2 channels.  With 64 more elements per row (128 bytes) the rows of a tile spread over all channels: fc2 forward
    # 1 117-1 126 -> 1 269-1 323 TFLOP/s (+14..17 %), the fc1 weight gradient +10 %, dX of fc1 +4 %.  The pad columns are never
    # read or written (every consumer takes the row pitch: GEMM operands / outputs, stored gelu').
    ffn_row_pad = 64
    # ... and of the d_model-wide bf16 buffers inside the layers (LayerNorm outputs, the fused QKV / Q / KV projections and their
    # gradients, attention outputs, the bf16 copy of the residual gradient): 2 560 / 7 680-byte rows fall on 8 of the 16 channels --
    # QKV forward +4 %, fc1 forward +2 %, the fc1 weight gradient (h is its second operand) +10 % in the probe.  The encoder's
    # and decoder's final LayerNorm outputs (handed to callers / the LM head) and the convolution stem keep dense rows.
    row_pad = 64
    # ... and of the residual stream the out-proj / fc2 epilogues read and write (fp32 for the student: 5 120-byte rows, four channels
    # per 32-row slab of a wave; 128 bytes = 32 floats / 64 bf16 of pad)
    stream_row_pad = 128
    # ... and of the dX outputs the backward hands to LayerNorm / attention (dh, dO)
    dx_row_pad = 64

    def act(self, rows, cols, dtype=None, zero_pad=True, pad=0):
        """Activation buffer with rows padded to a multiple of 64 and the pad rows zeroed: the weight-gradient GEMMs
        contract over the token dimension in K-steps of 64 and must see zeros there.  zero_pad=False (forward-only
        passes: nothing contracts over the rows) leaves the pad rows as they are -- a decoder pass of the frozen
        teacher was ~350 five-microsecond fill launches otherwise.  pad > 0: a [rows, cols] view of a buffer whose rows
        are cols + pad elements apart (see ffn_row_pad)."""
        dtype = self.lowp if dtype is None else dtype
        rp = _rup(rows, 64)
        t = self.ops.empty((rp, cols + pad), dtype)
        if pad:
            t = t[:, :cols]
        if rp > rows and zero_pad:
            t[rows:].zero_()
        return t

    def _gemm_rows(self, R, save):
        if save or not self.pad_gemm_rows or R < self.pad_gemm_rows_min:
            return R
        Rg = _rup(R, 320)
        return Rg if Rg - R <= R * self.pad_gemm_rows_slack else R

    def _ln(self, name, x, R, save, rows_alloc=None, zero_pad=None, pad=0):
        y = self.act(R if rows_alloc is None else rows_alloc, x.shape[1], zero_pad=save if zero_pad is None else zero_pad, pad=pad)
        _, mu, rs = self.ops.layernorm_fwd(x[:R] if x.shape[0] != R else x, self.st.p[f"{name}.weight"],
                                           self.st.p[f"{name}.bias"], 1e-5, save_stats=save, out=y[:R])
        return y, mu, rs

    def _ln_bwd(self, name, dy, x, mu, rs, dres, R, emit=False, colsum_to=None):
        """LayerNorm backward into the fp32 residual-gradient stream.  With emit=True the kernel also writes the
        low-precision copy of the updated stream (what the next residual branch's GEMMs consume) and adds its column
        sums to `colsum_to` (the bias gradient of that branch's output projection).  Returns (dres, dy_next)."""
        if self.st.is_trainable(f"{name}.weight"):
            dg, db = self.st.g[f"{name}.weight"], self.st.g[f"{name}.bias"]
        else:
            dg, db = self._scratch_vec(x.shape[1]), self._scratch_vec(x.shape[1], 1)
        nxt = self.act(R, x.shape[1], pad=self.row_pad) if emit else None
        dres = self.ops.layernorm_bwd(dy[:R], x[:R] if x.shape[0] != R else x, mu, rs, self.st.p[f"{name}.weight"],
                                      dres, dg, db, out_lowp=nxt[:R] if emit else None,
                                      colsum=colsum_to if emit else None)
        return dres, nxt

    def _train_scatter_buf(self, slot, rows, cols):
        """(batch, position)-layout buffers of the PACKED TRAINING pass (pack_train_layers): one per (layer, operand) slot -- the
        backward reads them -- zero-initialised once and only ever written with projected / computed rows, so rows that are dead
        in this step hold finite values of an earlier one (all the attention kernels need of them: their dO is exactly zero)."""
        if not hasattr(self, "_tsb"):
            self._tsb = {}
        # (never freed or replaced: a captured step has the addresses baked in -- a larger plan ADDS a buffer, like _scatter_buf)
        fits = [b for b in self._tsb.get((slot, cols), []) if b.shape[0] >= rows]
        if fits:
            return min(fits, key=lambda b: b.shape[0])
        buf = self.ops.zeros((_rup(rows, 64), cols), self.lowp)
        self._tsb.setdefault((slot, cols), []).append(buf)
        return buf

    def _scatter_buf(self, rows, cols):
        """(batch, position)-layout target of the packed pass's scatter: zero-initialised once, afterwards it only ever
        receives projected rows, so its dead rows stay finite.  Buffers are never freed or replaced while the engine lives
        -- a captured step (train_step_graphed) has their addresses baked in -- a request is served by the smallest
        existing buffer of that width with enough rows, a larger one is added when none has."""
        if not hasattr(self, "_sb"):
            self._sb = {}
        fits = [b for b in self._sb.get(cols, []) if b.shape[0] >= rows]
        if fits:
            return min(fits, key=lambda b: b.shape[0])
        buf = self.ops.zeros((_rup(rows, 64), cols), self.lowp)
        self._sb.setdefault(cols, []).append(buf)
        return buf

    def _scratch_vec(self, n, slot=0):
        key = (n, slot)
        if not hasattr(self, "_sv"):
            self._sv = {}
        if key not in self._sv:
            self._sv[key] = self.ops.zeros((n,), torch.float32)
        return self._sv[key]

    def _wgrad(self, dy, x, gout, gbias, R, bias_cols=None):
        """gout (+)= dy^T . x over the (padded) token dimension; gbias (+)= column sums of dy."""
        ws = self.wgrad_stream
        if ws is not None:
            # Weight gradients are off the critical path of the backward (nothing below reads them): issued on a second
            # stream, their persistent GEMMs take the CUs the dX chain's kernels leave idle in their last tile round
            # (and vice versa).  Ordering: the side stream waits for everything the main stream has enqueued so far
            # (dy and x are complete); the caller joins the streams before anyone reads the gradients
            # (join_wgrad_stream).  Lifetime: dy and x belong to the main stream's allocator pool, which hands a freed
            # block to the next main-stream kernel at once -- so this engine keeps them referenced until the main
            # stream has waited for an event recorded behind their last side-stream reader (_wgrad_fence).  No
            # `record_stream`: that defers the reuse of a block until the host observes the side stream's progress, and
            # with the host a few steps ahead of the device every step's activations (94 GiB) came from fresh
            # hipMalloc'ed blocks until the device was full (round 2: 287 GiB reserved, one 4.6 s allocator retry).
            main = torch.cuda.current_stream(dy.device)
            ws.wait_stream(main)
            with torch.cuda.stream(ws):
                self._wgrad_issue(dy, x, gout, gbias, R, bias_cols)
            self._wgrad_held += [dy, x]
        else:
            self._wgrad_issue(dy, x, gout, gbias, R, bias_cols)

    pack_train_layers = True  # training passes with per-sequence label lengths: the decoder layers' row-local work (GEMMs, LayerNorm, their
                              # backward) over the packed live rows, attention through the rectangle (False: the layers keep the rectangle)
    varlen_attention = True   # packed (live-row) forward passes: attention over ragged batches in place (False: scatter -> rectangle -> gather)
    wgrad_overwrite = False   # set by the trainer around a backward that follows zero_small_grads(): every weight matrix handed to
                              # _wgrad gets exactly one contribution per backward, so the split-K combine may store instead of add
    wgrad_lag = 2     # layers of weight-gradient work the side stream may be behind before the main stream waits for it

    def _wgrad_fence(self):
        """Layer boundary of the backward: mark the side stream's position; tensors handed to it `wgrad_lag` boundaries
        ago are released after the main stream has waited for that mark."""
        ws = self.wgrad_stream
        if ws is None or not (self._wgrad_held or self._wgrad_fences):
            return
        ev = torch.cuda.Event()
        ev.record(ws)
        self._wgrad_fences.append((ev, self._wgrad_held))
        self._wgrad_held = []
        while len(self._wgrad_fences) > self.wgrad_lag:
            ev, held = self._wgrad_fences.pop(0)
            torch.cuda.current_stream(self.st.P.device).wait_event(ev)
            held.clear()

    def _wgrad_issue(self, dy, x, gout, gbias, R, bias_cols):
        if gout is not None:  # the gradient buffer was zeroed (or holds earlier micro-batches): always accumulate
            self.ops.gemm(dy, x, trans_a=True, trans_b=True, out_dtype=torch.float32, out=gout, atomic_acc=True,
                          overwrite=self.wgrad_overwrite)
        if gbias is not None:
            if bias_cols is None:
                self.ops.colsum(dy[:R], gbias, accumulate=True)
            else:
                for lo, hi in bias_cols:
                    self.ops.colsum(dy[:R, lo:hi], gbias[lo:hi], accumulate=True)

    def join_wgrad_stream(self):
        """The main stream waits for every weight-gradient kernel issued so far (before the optimizer, the gradient
        all-reduce of a finished range, or any other reader of the gradient buffer)."""
        if self.wgrad_stream is not None:
            torch.cuda.current_stream(self.st.P.device).wait_stream(self.wgrad_stream)
        self._wgrad_fences.clear()
        self._wgrad_held = []

    # ---- encoder -------------------------------------------------------------------------------------------------
    def encode(self, mel, save=False):
        """WhisperEncoder.forward (TF:modeling_whisper.py:592-646).  mel fp32 [B, n_mels, 3000] -> (enc_out
        low-precision [rows_padded, D], ctx)."""
        ops, st, d = self.ops, self.st, self.dims
        B, _, T = mel.shape
        D, H = d.d_model, d.heads
        R1, R = B * T, B * T // 2
        L = T // 2
        ctx = {"B": B, "T": T, "R": R, "layers": []} if save else None
        xcol1 = self.act(R1, st.kpad1, zero_pad=save)
        ops.im2col_mel(mel, st.kpad1, out=xcol1[:R1])
        a1 = self.act(R1, D, zero_pad=save)
        _, z1 = ops.gemm(xcol1[:R1], st.conv1_packed, bias=st.p["model.encoder.conv1.bias"], act=1, want_z=True,
                         out=a1[:R1])
        xcol2 = self.act(R, 3 * D, zero_pad=save)
        ops.im2col_s2(a1[:R1], B, T, out=xcol2[:R])
        x = ops.empty((R, D), self.stream)
        _, z2 = ops.gemm(xcol2[:R], st.conv2_packed, bias=st.p["model.encoder.conv2.bias"], act=1, want_z=True,
                         residual=st.p["model.encoder.embed_positions.weight"], r_row_mod=L, round_res=True, out=x)
        if save:
            ctx.update(xcol1=xcol1, z1=z1, xcol2=xcol2, z2=z2)
        else:
            del xcol1, a1, z1, xcol2, z2
        for i in range(d.enc_layers):
            x, lc = self._layer_fwd(f"model.encoder.layers.{i}", x, B, L, None, 0, False, save)
            if save:
                ctx["layers"].append(lc)
        # (pad rows zeroed also in a forward-only pass: with a frozen encoder its output is still an operand of the
        # decoder's k_proj / v_proj weight-gradient GEMMs, which contract over the padded rows -- 0 x NaN from stale memory)
        y, mu, rs = self._ln("model.encoder.layer_norm", x, R, save, zero_pad=True)
        if save:
            ctx.update(x_final=x, mu=mu, rs=rs, enc_out=y)
        return y, ctx

    def _layer_fwd(self, p, x, B, L, enc_out, Lk, causal, save, Rg=None, live=None):
        """One pre-LN transformer layer (TF:modeling_whisper.py:379-413 encoder, 448-505 decoder).  Rg >= the valid
        rows: rows the projections run over (`pad_gemm_rows`; x then has Rg rows); attention and LayerNorm see the valid
        rows.  live (forward-only passes): x holds the packed live rows (LiveRows); attention runs in the (batch,
        position) layout between a scatter and a gather -- its dead rows hold stale memory, which only ever reaches
        dead rows (queries are independent, the causal mask hides later keys, encoder keys are all live)."""
        ops, st, d = self.ops, self.st, self.dims
        D, H, Rp = d.d_model, d.heads, B * L
        R = Rp if live is None else live.n          # valid rows of x
        Rg = R if Rg is None else Rg
        assert Rg == R or not save
        lc = {} if save else None
        rect = {}                              # (packed training pass: the rectangular q / k / v / o of this layer's attentions)

        def attend(q_src, k, v, Lkv, is_causal, cols):
            """q_src [>= R, cols] with the queries in its first D columns -> o [Rg, D]"""
            o = self.act(Rg, D, zero_pad=save, pad=self.row_pad)
            if live is None:
                _, lse = ops.attn_fwd(q_src[:R, :D], k, v, B, H, L, Lkv, is_causal, 0.125, out=o[:R])
                return o, lse
            if save:
                # Packed TRAINING pass (pack_train_layers): the row-local work of the layer runs over the live rows, attention --
                # whose backward kernels want the (batch, position) rectangle -- between a scatter and a gather, on buffers this
                # layer keeps for its backward.  Dead rows of the rectangle: finite values (see _train_scatter_buf).
                tag = "x" if k is not None else "s"
                qp = self._train_scatter_buf((p, tag, "q"), Rp, cols)
                ops.scatter_rows(q_src[:R], live.idx, qp)
                op = self._train_scatter_buf((p, tag, "o"), Rp, D)
                kk, vv = (qp[:Rp, D:2 * D], qp[:Rp, 2 * D:]) if k is None else (k, v)
                _, lse = ops.attn_fwd(qp[:Rp, :D], kk, vv, B, H, L, Lkv, is_causal, 0.125, out=op[:Rp])
                ops.gather_rows(op, live.idx, o)
                rect[tag] = (qp, op)
                return o, lse
            if live.seq_start is not None and self.varlen_attention:
                # ragged batches: the kernel walks the packed rows of every sequence in place (same arithmetic per query row as over
                # the rectangle: bit-identical live rows; no scatter / gather launches, no dead query rows)
                if k is None:
                    ops.attn_fwd_varlen(q_src[:R, :D], q_src[:R, D:2 * D], q_src[:R, 2 * D:], H, live.max_q, live.seq_start,
                                        live.seq_len, is_causal, 0.125, o[:R], self_attention=True,
                                        flops=256.0 * H * live.attn_flops[0])
                else:
                    ops.attn_fwd_varlen(q_src[:R, :D], k, v, H, live.max_q, live.seq_start, live.seq_len, is_causal, 0.125, o[:R],
                                        Lk=Lkv, kv_batches=B, self_attention=False, flops=256.0 * H * Lkv * live.attn_flops[1])
                return o, None
            # Dead rows must hold FINITE values: a live query's masked keys (later positions inside its 64-key tile) enter
            # the P.V product with probability exactly 0, and 0 x NaN from stale memory would poison the live row.  The
            # scatter target is a buffer of this engine that starts zeroed and only ever receives projected rows.
            qp = self._scatter_buf(Rp, cols)
            ops.scatter_rows(q_src[:R], live.idx, qp)
            op = self.act(Rp, D, zero_pad=False)
            kk, vv = (qp[:Rp, D:2 * D], qp[:Rp, 2 * D:]) if k is None else (k, v)
            ops.attn_fwd(qp[:Rp, :D], kk, vv, B, H, L, Lkv, is_causal, 0.125, out=op[:Rp])
            ops.gather_rows(op, live.idx, o)
            return o, None
        # --- self attention
        av = st.attn_views(f"{p}.self_attn")
        h, mu, rs = self._ln(f"{p}.self_attn_layer_norm", x, R, save, Rg, pad=self.row_pad)
        qkv = self.act(Rg, 3 * D, zero_pad=save, pad=self.row_pad)
        ops.gemm(h[:Rg], av["wqkv"], bias=av["bqkv"], out=qkv[:Rg])
        if live is None:
            o, lse = attend(qkv, qkv[:R, D:2 * D], qkv[:R, 2 * D:], L, causal, 3 * D)
        else:
            o, lse = attend(qkv, None, None, L, causal, 3 * D)
        xp = self.stream_row_pad // (4 if self.stream == torch.float32 else 2)
        x1 = ops.gemm(o[:Rg], av["wo"], bias=av["bo"], residual=x, round_res=True, out_dtype=self.stream, out_row_pad=xp)
        if save:
            lc.update(x0=x, mu0=mu, rs0=rs, h0=h, qkv=qkv, o0=o, lse0=lse)
            if live is not None:
                lc.update(qkv_r=rect["s"][0], o0_r=rect["s"][1])
        x = x1
        # --- cross attention (decoder only)
        if enc_out is not None:
            cv = st.attn_views(f"{p}.encoder_attn")
            Re = B * Lk
            h, mu, rs = self._ln(f"{p}.encoder_attn_layer_norm", x, R, save, Rg, pad=self.row_pad)
            q = self.act(Rg, D, zero_pad=save, pad=self.row_pad)
            ops.gemm(h[:Rg], cv["wqkv"][:D], bias=cv["bqkv"][:D], out=q[:Rg])
            kv = self.act(Re, 2 * D, zero_pad=save, pad=self.row_pad)
            ops.gemm(enc_out[:Re], cv["wqkv"][D:], bias=cv["bqkv"][D:], out=kv[:Re])
            o, lse = attend(q, kv[:Re, :D], kv[:Re, D:], Lk, False, D)
            x1 = ops.gemm(o[:Rg], cv["wo"], bias=cv["bo"], residual=x, round_res=True, out_dtype=self.stream, out_row_pad=xp)
            if save:
                lc.update(x1=x, mu1=mu, rs1=rs, h1=h, q1=q, kv1=kv, o1=o, lse1=lse)
                if live is not None:
                    lc.update(q1_r=rect["x"][0], o1_r=rect["x"][1])
            x = x1
        # --- feed forward
        h, mu, rs = self._ln(f"{p}.final_layer_norm", x, R, save, Rg, pad=self.row_pad)
        a = self.act(Rg, d.ffn, zero_pad=save, pad=self.ffn_row_pad)
        res = ops.gemm(h[:Rg], st.s[f"{p}.fc1.weight"], bias=st.p[f"{p}.fc1.bias"], act=1,
                       want_z=("grad" if self.ffn_keeps_gelu_grad else True) if save else False, out=a[:Rg],
                       z_row_pad=self.ffn_row_pad)
        z = res[1] if save else None
        x2 = ops.gemm(a[:Rg], st.s[f"{p}.fc2.weight"], bias=st.p[f"{p}.fc2.bias"], residual=x, round_res=True,
                      out_dtype=self.stream, out_row_pad=xp)
        if save:
            lc.update(x2=x, mu2=mu, rs2=rs, h2=h, a=a, z=z)
        return x2, lc

    # ---- decoder -------------------------------------------------------------------------------------------------
    def decode(self, ids, enc_out, save=False, live=None):
        """WhisperDecoder.forward + tied LM head (TF:modeling_whisper.py:690-795, 965, 1080).  ids int64 [B, T];
        enc_out low-precision [>= B*Lk, D].  Returns (logits low-precision [rows, ldv] with V valid columns, ctx); rows =
        B*T in (batch, position) order, or -- with `live` (LiveRows) -- its n live rows in packed order:
          * forward-only pass (the frozen teacher): the whole decoder runs over the packed rows (_layer_fwd);
          * training pass: the LM head and with it dE = dlogits^T . hf and dhf = dlogits . E run over the live rows only; with
            pack_train_layers (default) so does every row-local operation of the layers, forward and backward -- attention alone
            goes through the (batch, position) rectangle its backward kernels want (_layer_fwd / _layer_bwd); otherwise the layers
            keep the rectangle."""
        ops, st, d = self.ops, self.st, self.dims
        B, T = ids.shape
        R, Lk = B * T, d.max_src
        assert live is None or (live.B == B and live.T == T)
        ctx = {"B": B, "T": T, "R": R, "ids": ids, "layers": [], "enc_out": enc_out} if save else None
        packed = live is not None and (not save or self.pack_train_layers)
        if packed and not save and getattr(self, "_sb", None):
            # The dead rows of the scatter targets must be finite (see _layer_fwd.attend).  They only ever receive
            # projected rows, but one overflowing row of one bad batch would stay in a dead slot for the rest of the run
            # and reach later batches through 0 x NaN: one fill per buffer and pass (two fills of a 32-layer pass).
            for bufs in self._sb.values():
                for b_ in bufs:
                    b_.zero_()
        Rv = live.n if packed else R                 # rows the layers run over
        self._last_decode_rows = Rv                  # (tests)
        Rg = self._gemm_rows(Rv, save)
        if self.stream == torch.float32:
            x = ops.embed_fwd(ids, st.p["model.decoder.embed_tokens.weight"],
                              st.p["model.decoder.embed_positions.weight"], torch.float32, rows_alloc=R if packed else Rg)
        else:
            x = ops.embed_fwd(ids, st.s["model.decoder.embed_tokens.weight"],
                              st.s["model.decoder.embed_positions.weight"], self.lowp, rows_alloc=R if packed else Rg)
        if packed:
            xp, x = x, ops.empty((Rg, d.d_model), self.stream)
            ops.gather_rows(xp, live.idx, x)
            del xp
        for i in range(d.dec_layers):
            x, lc = self._layer_fwd(f"model.decoder.layers.{i}", x, B, T, enc_out, Lk, True, save, Rg,
                                    live if packed else None)
            if save:
                ctx["layers"].append(lc)
        # Training pass: the rows of hf / logits are padded with ZERO rows to a multiple of 320 (same rule as
        # pad_gemm_rows) so that the backward's dhf = dlogits . E (M = rows, N = D, K = padded vocabulary) is one round
        # of 320-row tiles instead of two rounds of 256-tiles; zero rows add nothing to dE = dlogits^T . hf.
        Rh = live.n if live is not None else R       # rows of the LM head
        Rl = Rh
        if save and self.pad_lm_rows and Rh >= self.pad_gemm_rows_min:
            Rl = _rup(Rh, 320)
            Rl = Rl if Rl - Rh <= Rh * self.pad_gemm_rows_slack else Rh
        hf, mu, rs = self._ln("model.decoder.layer_norm", x, Rv, save, max(Rl, Rv))
        if live is not None and save and not packed: # (batch, position) rows of the final LayerNorm -> live rows
            hfp, hf = hf, self.act(Rl, d.d_model)
            ops.gather_rows(hfp, live.idx, hf)
            del hfp
        logits = self.act(Rl, self.ldv, zero_pad=save)
        if Rl > Rh:
            hf[Rh:Rl].zero_()
            logits[Rh:Rl].zero_()
        # N = padded vocabulary (multiple of 64): the rows of the shadow buffer behind E are finite parameters / zero
        # slack, the resulting pad columns are never read as logits (the loss kernel stops at V and zeroes them)
        eo = st.entries["model.decoder.embed_tokens.weight"][0]
        e_pad = st.S[eo:eo + self.ldv * d.d_model].view(self.ldv, d.d_model)
        ops.gemm(hf[:Rh], e_pad, out=logits[:Rh])
        if save:
            ctx.update(x_final=x, mu=mu, rs=rs, hf=hf, lm_rows=Rl, live=live, packed=packed, Rv=Rv)
        return logits, ctx

    # ---- incremental decoding with a KV cache (TF:modeling_whisper.py:312-335, EncoderDecoderCache) -----------------
    def decode_init(self, enc_out, B, max_len, cache=None):
        """Per decoder layer: the static cross-attention K/V projected once from the encoder output, and an empty
        self-attention K/V cache laid out [B][max_len][2D] that the attention kernel reads in place (batch pitch =
        max_len rows).  Passing a `cache` made by an earlier call with the same (B, max_len) refills its buffers in
        place (their addresses are baked into captured HIP graphs, see decoding.GreedyDecoder)."""
        ops, st, d = self.ops, self.st, self.dims
        D, Re = d.d_model, B * d.max_src
        if cache is None:
            cache = {"B": B, "max_len": max_len, "t": 0, "cross": [None] * d.dec_layers,
                     "self": [ops.zeros((B * max_len, 2 * D), self.lowp) for _ in range(d.dec_layers)]}
        assert cache["B"] == B and cache["max_len"] == max_len
        cache["t"] = 0
        for i in range(d.dec_layers):
            cv = st.attn_views(f"model.decoder.layers.{i}.encoder_attn")
            cache["cross"][i] = ops.gemm(enc_out[:Re], cv["wqkv"][D:], bias=cv["bqkv"][D:], out=cache["cross"][i],
                                         out_row_pad=self.kv_row_pad)
        return cache

    # ---- single-call decoder pass (C entry dw_decode_step) ---------------------------------------------------------
    # the FFN's saved by-product is gelu'(z) in fp16 (the backward epilogue multiplies) instead of z in bf16 (it
    # evaluates erf/exp again): same bytes, no transcendentals in the dX GEMM of fc2
    ffn_keeps_gelu_grad = True
    fuse_fc1_bias_grad = True   # fc1.bias gradient from the epilogue of the dX GEMM of fc2 (DwGemm.colsum_out)
    # q_proj / v_proj bias gradients from the attention-backward kernels (dw_attn_bwd_ex): correct and tested, but OFF --
    # 64 cross-lane reductions per wave at the end of every attention-backward workgroup cost more (+4 ms per step) than
    # the two column-sum launches per layer they replace (profiles/r3_gemm_epilogue.md section 5)
    fuse_attn_bias_grad = False
    use_c_decode = True     # HIP path: one library call per decoder pass instead of ~30 per-kernel calls
    # rows of the static cross-attention K | V padded by 128 bytes: a head's 128-byte pieces of consecutive 5 120-byte rows fall on
    # four of an XCD's sixteen L2 channels (tools/decode_stride_probe.py: 29.8 -> 26.4 us per layer at batch 16)
    kv_row_pad = 64

    def _decode_desc(self, cache, n):
        """DwDecodeStep for `n` new positions per row over `cache` (built once per (cache, n): weights, caches and the
        workspace it points at are owned by this engine / the cache dict and stay alive with them)."""
        key = ("desc", n)
        if key in cache:
            return cache[key]
        from .ops_hip import DwDecodeStep, DwDecoderLayer, DW_BF16, DW_F32
        ops, st, d = self.ops, self.st, self.dims
        B, D, F = cache["B"], d.d_model, d.ffn
        rows = B * n
        f32 = self.stream == torch.float32
        tab = st.p if f32 else st.s
        ws = {"x": ops.empty((rows, D), self.stream), "h": ops.empty((rows, D), self.lowp),
              "qkv": ops.empty((rows, 3 * D), self.lowp), "o": ops.empty((rows, D), self.lowp),
              "a": ops.empty((rows, F), self.lowp), "logits": ops.empty((rows, self.ldv), self.lowp)}
        layers = (DwDecoderLayer * d.dec_layers)()
        keep = [ws]
        for i in range(d.dec_layers):
            p = f"model.decoder.layers.{i}"
            av, cv = st.attn_views(f"{p}.self_attn"), st.attn_views(f"{p}.encoder_attn")
            L = layers[i]
            L.ln1_g, L.ln1_b = st.p[f"{p}.self_attn_layer_norm.weight"].data_ptr(), st.p[f"{p}.self_attn_layer_norm.bias"].data_ptr()
            L.wqkv, L.bqkv = av["wqkv"].data_ptr(), av["bqkv"].data_ptr()
            L.wo, L.bo = av["wo"].data_ptr(), av["bo"].data_ptr()
            L.ln2_g, L.ln2_b = st.p[f"{p}.encoder_attn_layer_norm.weight"].data_ptr(), st.p[f"{p}.encoder_attn_layer_norm.bias"].data_ptr()
            L.wq, L.bq = cv["wqkv"][:D].data_ptr(), cv["bqkv"][:D].data_ptr()
            L.wo2, L.bo2 = cv["wo"].data_ptr(), cv["bo"].data_ptr()
            L.ln3_g, L.ln3_b = st.p[f"{p}.final_layer_norm.weight"].data_ptr(), st.p[f"{p}.final_layer_norm.bias"].data_ptr()
            L.w1, L.b1 = st.s[f"{p}.fc1.weight"].data_ptr(), st.p[f"{p}.fc1.bias"].data_ptr()
            L.w2, L.b2 = st.s[f"{p}.fc2.weight"].data_ptr(), st.p[f"{p}.fc2.bias"].data_ptr()
            L.self_kv, L.cross_kv = cache["self"][i].data_ptr(), cache["cross"][i].data_ptr()
            keep += [av, cv]
        eo = st.entries["model.decoder.embed_tokens.weight"][0]
        desc = DwDecodeStep()
        desc.batch, desc.n_new, desc.d_model, desc.heads, desc.ffn, desc.n_layers = B, n, D, d.heads, F, d.dec_layers
        desc.src_len, desc.max_len, desc.vocab, desc.ldv = d.max_src, cache["max_len"], d.vocab, self.ldv
        desc.stream_dtype = DW_F32 if f32 else DW_BF16
        desc.cross_kv_ld = cache["cross"][0].stride(0)
        assert all(c.stride(0) == desc.cross_kv_ld for c in cache["cross"])
        desc.tok_emb = tab["model.decoder.embed_tokens.weight"].data_ptr()
        desc.pos_emb = tab["model.decoder.embed_positions.weight"].data_ptr()
        desc.lnf_g, desc.lnf_b = st.p["model.decoder.layer_norm.weight"].data_ptr(), st.p["model.decoder.layer_norm.bias"].data_ptr()
        desc.lm_head = st.S[eo:eo + self.ldv * D].data_ptr()
        desc.layers = C.cast(layers, C.c_void_p)
        for k, v in ws.items():
            setattr(desc, k, v.data_ptr())
        cache[key] = (desc, ws, layers, keep)
        return cache[key]

    def _decode_pass_c(self, ids, cache, n):
        desc, ws, _, _ = self._decode_desc(cache, n)
        ids = ids.contiguous()
        desc.ids, desc.t = ids.data_ptr(), cache["t"]
        self.ops.decode_pass(desc)
        cache["t"] += n
        return ws["logits"]     # (workspace of the cache: valid until the next pass with the same n)

    def _decoder_layers_cached(self, x, cache, n):
        """The decoder layers of one cached pass over `n` new positions per row (x: [B*n, D] embedded inputs).  With
        few rows (B*n <= 32 / 64) the LayerNorm in front of each projection is computed inside the weight-streaming
        GEMM and the new K/V go straight into the cache (ops.gemm ln= / kv_append=); otherwise they are separate
        launches.  This is the per-kernel PYTHON form of the pass (`use_c_decode = False`; the CPU restatement of the
        tests runs it).  `dw_decode_step` computes the same pass in fewer launches: since round 5 it runs LayerNorm + q
        projection + cross-attention in ONE kernel (`attn_decode_proj_kernel`, dw_debug_set key 7, default 4), whose fp32
        summation order differs from the GEMV's -- q may differ by one bf16 ulp between the two paths.  Tests that
        compare them either pin `dw_debug_set(7, 12)` (the unfused sequence: exactly this one) or state a tolerance."""
        ops, st, d = self.ops, self.st, self.dims
        B, t, ML = cache["B"], cache["t"], cache["max_len"]
        D, H, Lk = d.d_model, d.heads, d.max_src
        rows = B * n
        fuse_ln, fuse_kv = rows <= 32 and D <= 1280, rows <= 64
        causal = 2 if n > 1 else False
        for i in range(d.dec_layers):
            p = f"model.decoder.layers.{i}"
            av = st.attn_views(f"{p}.self_attn")
            ln1 = (st.p[f"{p}.self_attn_layer_norm.weight"], st.p[f"{p}.self_attn_layer_norm.bias"], 1e-5)
            kvc = cache["self"][i]
            h = x if fuse_ln else ops.layernorm_fwd(x, *ln1, save_stats=False)[0]
            qkv = ops.gemm(h, av["wqkv"], bias=av["bqkv"], ln=ln1[:2] if fuse_ln else None,
                           kv_append=(kvc, D, n, ML, t) if fuse_kv else None)
            if not fuse_kv:
                kvc.view(B, ML, 2 * D)[:, t:t + n].copy_(qkv[:, D:].view(B, n, 2 * D))   # append this pass's K/V
            o, _ = ops.attn_fwd(qkv[:, :D], kvc[:, :D], kvc[:, D:], B, H, n, t + n, causal, 0.125, kv_batch_rows=ML)
            x = ops.gemm(o, av["wo"], bias=av["bo"], residual=x, round_res=True, out_dtype=self.stream)
            cv = st.attn_views(f"{p}.encoder_attn")
            ln2 = (st.p[f"{p}.encoder_attn_layer_norm.weight"], st.p[f"{p}.encoder_attn_layer_norm.bias"], 1e-5)
            h = x if fuse_ln else ops.layernorm_fwd(x, *ln2, save_stats=False)[0]
            q = ops.gemm(h, cv["wqkv"][:D], bias=cv["bqkv"][:D], ln=ln2[:2] if fuse_ln else None)
            kv = cache["cross"][i]
            o, _ = ops.attn_fwd(q, kv[:, :D], kv[:, D:], B, H, n, Lk, False, 0.125)
            x = ops.gemm(o, cv["wo"], bias=cv["bo"], residual=x, round_res=True, out_dtype=self.stream)
            ln3 = (st.p[f"{p}.final_layer_norm.weight"], st.p[f"{p}.final_layer_norm.bias"], 1e-5)
            h = x if fuse_ln else ops.layernorm_fwd(x, *ln3, save_stats=False)[0]
            a = ops.gemm(h, st.s[f"{p}.fc1.weight"], bias=st.p[f"{p}.fc1.bias"], act=1, ln=ln3[:2] if fuse_ln else None)
            x = ops.gemm(a, st.s[f"{p}.fc2.weight"], bias=st.p[f"{p}.fc2.bias"], residual=x, round_res=True,
                         out_dtype=self.stream)
        return x

    def decode_step(self, ids_t, cache):
        """One greedy-decoding step: ids_t int64 [B, 1] at position cache["t"] -> logits low-precision [B, ldv].
        ALIASING: on the HIP path (`use_c_decode`) the returned tensor is the per-(cache, n) workspace of
        dw_decode_step: it is overwritten by the next pass over the same cache with the same number of new positions
        (its address is what captured HIP graphs replay into).  Consume or clone it before the next step."""
        ops, st, d = self.ops, self.st, self.dims
        B, t, ML = cache["B"], cache["t"], cache["max_len"]
        D, H, Lk = d.d_model, d.heads, d.max_src
        assert ids_t.shape == (B, 1) and t < ML and t < d.max_tgt
        if self.use_c_decode and hasattr(ops, "decode_pass"):
            return self._decode_pass_c(ids_t, cache, 1)
        f32 = self.stream == torch.float32
        tok = (st.p if f32 else st.s)["model.decoder.embed_tokens.weight"]
        pos = (st.p if f32 else st.s)["model.decoder.embed_positions.weight"][t:t + 1]
        x = ops.embed_fwd(ids_t.contiguous(), tok, pos, torch.float32 if f32 else self.lowp)
        x = self._decoder_layers_cached(x, cache, 1)
        hf, _, _ = ops.layernorm_fwd(x, st.p["model.decoder.layer_norm.weight"], st.p["model.decoder.layer_norm.bias"],
                                     1e-5, save_stats=False)
        eo = st.entries["model.decoder.embed_tokens.weight"][0]
        e_pad = st.S[eo:eo + self.ldv * D].view(self.ldv, D)
        logits = ops.gemm(hf, e_pad)
        cache["t"] = t + 1
        return logits

    def decode_multi(self, ids, cache):
        """n >= 1 new tokens per row at positions cache["t"] .. cache["t"]+n-1 against the KV cache in ONE decoder pass
        (ids int64 [B, n]) -> logits low-precision [B*n, ldv] (row b*n + j = position t+j of row b).  The new keys and
        values are appended first and the self-attention runs with the bottom-right aligned causal mask (query j sees
        keys <= t + j): the cached multi-token verify step of speculative decoding (run_eval.py:578-599) and the
        prompt prefill.  cache["t"] advances by n; callers roll it back to drop rejected positions.  The returned logits
        alias the cache's workspace for this n on the HIP path (see decode_step)."""
        ops, st, d = self.ops, self.st, self.dims
        B, t, ML = cache["B"], cache["t"], cache["max_len"]
        n = ids.shape[1]
        D, H, Lk = d.d_model, d.heads, d.max_src
        assert ids.shape[0] == B and t + n <= ML and t + n <= d.max_tgt
        if self.use_c_decode and hasattr(ops, "decode_pass"):
            return self._decode_pass_c(ids, cache, n)
        f32 = self.stream == torch.float32
        tok = (st.p if f32 else st.s)["model.decoder.embed_tokens.weight"]
        pos = (st.p if f32 else st.s)["model.decoder.embed_positions.weight"][t:t + n]
        x = ops.embed_fwd(ids.contiguous(), tok, pos.contiguous(), torch.float32 if f32 else self.lowp)
        x = self._decoder_layers_cached(x, cache, n)
        hf, _, _ = ops.layernorm_fwd(x, st.p["model.decoder.layer_norm.weight"], st.p["model.decoder.layer_norm.bias"],
                                     1e-5, save_stats=False)
        eo = st.entries["model.decoder.embed_tokens.weight"][0]
        e_pad = st.S[eo:eo + self.ldv * D].view(self.ldv, D)
        logits = ops.gemm(hf, e_pad)
        cache["t"] = t + n
        return logits

    # ---- backward --------------------------------------------------------------------------------------------------
    def _bias_grad(self, name):
        return self.st.g[name] if self.st.is_trainable(name) else None

    def _layer_bwd(self, p, lc, dres, dy, B, L, Lk, causal, denc, emit_last, colsum_last, live=None):
        """Backward of _layer_fwd.  dres: fp32 [R, D] gradient w.r.t. the layer output (updated in place to the
        gradient w.r.t. the layer input); dy: its low-precision copy, whose column sums were already added to this
        layer's fc2.bias gradient by the kernel that produced it.  denc: fp32 [Re, D] accumulator for the encoder
        output gradient.  Returns (dres, dy_for_the_layer_below)."""
        ops, st, d = self.ops, self.st, self.dims
        D, H, Rp = d.d_model, d.heads, B * L
        R = Rp if live is None else live.n      # live (packed training pass, _layer_fwd): row-local work over the live rows, attention
                                                # over the layer's rectangular buffers between a scatter and a gather
        tr = st.is_trainable(f"{p}.fc1.weight")
        cross = "x1" in lc

        def to_rect(t, cols):
            """packed rows -> a ZEROED (batch, position) rectangle (a row that is dead in this step must contribute nothing)"""
            r = ops.zeros((_rup(Rp, 64), cols), t.dtype)
            ops.scatter_rows(t[:R], live.idx, r)
            return r
        # --- feed forward
        dz = self.act(R, d.ffn, pad=self.ffn_row_pad)
        # (fc1.bias gradient = column sums of dz: accumulated by this GEMM's epilogue, no separate pass over dz)
        fuse_cs = tr and self.fuse_fc1_bias_grad and R > 64
        ops.gemm(dy[:R], st.s[f"{p}.fc2.weight"], trans_b=True, zgrad=lc["z"], out=dz[:R],
                 colsum=st.g[f"{p}.fc1.bias"] if fuse_cs else None)
        if tr:
            self._wgrad(dy, lc["a"], st.g[f"{p}.fc2.weight"], None, R)
            self._wgrad(dz, lc["h2"], st.g[f"{p}.fc1.weight"], None if fuse_cs else st.g[f"{p}.fc1.bias"], R)
        dh = ops.gemm(dz[:R], st.s[f"{p}.fc1.weight"], trans_b=True, out_row_pad=self.dx_row_pad)
        nb = f"{p}.encoder_attn.out_proj.bias" if cross else f"{p}.self_attn.out_proj.bias"
        dres, dy = self._ln_bwd(f"{p}.final_layer_norm", dh, lc["x2"], lc["mu2"], lc["rs2"], dres, R, emit=True,
                                colsum_to=self._bias_grad(nb))
        del dz, dh
        # --- cross attention
        if cross:
            cv = st.attn_views(f"{p}.encoder_attn")
            Re = B * Lk
            do = ops.gemm(dy[:R], cv["wo"], trans_b=True, out_row_pad=self.dx_row_pad)
            dq = self.act(R, D, pad=self.row_pad)
            dkv = self.act(Re, 2 * D, pad=self.row_pad)
            fb = tr and self.fuse_attn_bias_grad      # q / v bias gradients from the attention-backward kernels
            if live is None:
                ops.attn_bwd(lc["q1"][:R], lc["kv1"][:Re, :D], lc["kv1"][:Re, D:], lc["o1"][:R], do, lc["lse1"], B, H, L,
                             Lk, False, 0.125, dq=dq[:R], dk=dkv[:Re, :D], dv=dkv[:Re, D:],
                             dq_colsum=cv["g_bqkv"][:D] if fb else None, dv_colsum=cv["g_bqkv"][2 * D:] if fb else None)
            else:
                do_r = to_rect(do, D)
                dq_r = self.act(Rp, D, zero_pad=False)
                ops.attn_bwd(lc["q1_r"][:Rp], lc["kv1"][:Re, :D], lc["kv1"][:Re, D:], lc["o1_r"][:Rp], do_r[:Rp], lc["lse1"], B, H, L,
                             Lk, False, 0.125, dq=dq_r[:Rp], dk=dkv[:Re, :D], dv=dkv[:Re, D:],
                             dq_colsum=cv["g_bqkv"][:D] if fb else None, dv_colsum=cv["g_bqkv"][2 * D:] if fb else None)
                ops.gather_rows(dq_r, live.idx, dq)
                del do_r, dq_r
            if tr:
                self._wgrad(dy, lc["o1"], cv["g_wo"], None, R)
                self._wgrad(dq, lc["h1"], cv["g_wqkv"][:D], None if fb else cv["g_bqkv"][:D], R)
                self._wgrad(dkv, lc["enc_out"], cv["g_wqkv"][D:], None if fb else cv["g_bqkv"][D:], Re, bias_cols=[(D, 2 * D)])
            if denc is not None:
                ops.gemm(dkv[:Re], cv["wqkv"][D:], trans_b=True, residual=denc, round_res=True,
                         out_dtype=torch.float32, out=denc)
            dh = ops.gemm(dq[:R], cv["wqkv"][:D], trans_b=True, out_row_pad=self.dx_row_pad)
            dres, dy = self._ln_bwd(f"{p}.encoder_attn_layer_norm", dh, lc["x1"], lc["mu1"], lc["rs1"], dres, R,
                                    emit=True, colsum_to=self._bias_grad(f"{p}.self_attn.out_proj.bias"))
            del do, dq, dkv, dh
        # --- self attention
        av = st.attn_views(f"{p}.self_attn")
        do = ops.gemm(dy[:R], av["wo"], trans_b=True, out_row_pad=self.dx_row_pad)
        dqkv = self.act(R, 3 * D, pad=self.row_pad)
        qkv = lc["qkv"]
        fb = tr and self.fuse_attn_bias_grad
        if live is None:
            ops.attn_bwd(qkv[:R, :D], qkv[:R, D:2 * D], qkv[:R, 2 * D:], lc["o0"][:R], do, lc["lse0"], B, H, L, L, causal,
                         0.125, dq=dqkv[:R, :D], dk=dqkv[:R, D:2 * D], dv=dqkv[:R, 2 * D:],
                         dq_colsum=av["g_bqkv"][:D] if fb else None, dv_colsum=av["g_bqkv"][2 * D:] if fb else None)
        else:
            qr = lc["qkv_r"]
            do_r = to_rect(do, D)
            dqkv_r = self.act(Rp, 3 * D, zero_pad=False)
            ops.attn_bwd(qr[:Rp, :D], qr[:Rp, D:2 * D], qr[:Rp, 2 * D:], lc["o0_r"][:Rp], do_r[:Rp], lc["lse0"], B, H, L, L, causal,
                         0.125, dq=dqkv_r[:Rp, :D], dk=dqkv_r[:Rp, D:2 * D], dv=dqkv_r[:Rp, 2 * D:],
                         dq_colsum=av["g_bqkv"][:D] if fb else None, dv_colsum=av["g_bqkv"][2 * D:] if fb else None)
            ops.gather_rows(dqkv_r, live.idx, dqkv)
            del do_r, dqkv_r
        if tr:
            self._wgrad(dy, lc["o0"], av["g_wo"], None, R)
            self._wgrad(dqkv, lc["h0"], av["g_wqkv"], None if fb else av["g_bqkv"], R, bias_cols=[(0, D), (2 * D, 3 * D)])
        dh = ops.gemm(dqkv[:R], av["wqkv"], trans_b=True, out_row_pad=self.dx_row_pad)
        return self._ln_bwd(f"{p}.self_attn_layer_norm", dh, lc["x0"], lc["mu0"], lc["rs0"], dres, R, emit=emit_last,
                            colsum_to=colsum_last)

    def backward_decoder(self, ctx, dlogits, want_denc=True, accumulate=False):
        """dlogits: low-precision [>=R, ldv] (pad columns zero).  Accumulates parameter gradients into the store and
        returns the fp32 gradient w.r.t. the encoder output ([B*max_src, D]) or None."""
        ops, st, d = self.ops, self.st, self.dims
        self._accumulate = accumulate
        B, T, R, D = ctx["B"], ctx["T"], ctx["R"], d.d_model
        Lk = d.max_src
        emb = "model.decoder.embed_tokens.weight"
        tr_emb = st.is_trainable(emb)
        if tr_emb:
            # tied head: dE = dlogits^T . hf  (rows beyond V of the padded dlogits are not stored: m = V)
            # (first writer of dE in a backward: after zero_small_grads() it may store; the embedding backward adds to it later)
            ops.gemm(dlogits[:, :d.vocab], ctx["hf"], trans_a=True, trans_b=True, out_dtype=torch.float32,
                     out=st.g[emb], atomic_acc=True, overwrite=self.wgrad_overwrite)
        # dhf = dlogits . E : contraction over the padded vocabulary (pad columns of dlogits are zero; the rows of
        # the shadow buffer behind E are finite parameters / zero slack)
        eo = st.entries[emb][0]
        e_pad = st.S[eo:eo + self.ldv * D].view(self.ldv, D)
        Rl = ctx.get("lm_rows", R)
        assert dlogits.shape[0] >= Rl
        dh = ops.gemm(dlogits[:Rl], e_pad, trans_b=True)      # rows R..Rl of dlogits are zero (decode, pad_lm_rows)
        live = ctx.get("live")
        packed = bool(ctx.get("packed")) and live is not None
        Rv = ctx["Rv"] if packed else R                       # rows the layers' backward runs over
        if live is not None and not packed:                   # packed LM-head rows -> (batch, position); dead rows: zero
            dhp = self.act(R, D)
            dhp[:R].zero_()
            ops.scatter_rows(dh, live.idx, dhp)
            dh = dhp
        nl = d.dec_layers
        dres, dy = self._ln_bwd("model.decoder.layer_norm", dh, ctx["x_final"], ctx["mu"], ctx["rs"], None, Rv,
                                emit=True, colsum_to=self._bias_grad(f"model.decoder.layers.{nl - 1}.fc2.bias"))
        denc = ops.zeros((B * Lk, D), torch.float32) if want_denc else None
        for i in reversed(range(nl)):
            lc = ctx["layers"][i]
            lc["enc_out"] = ctx["enc_out"]
            below = self._bias_grad(f"model.decoder.layers.{i - 1}.fc2.bias") if i > 0 else None
            dres, dy = self._layer_bwd(f"model.decoder.layers.{i}", lc, dres, dy, B, T, Lk, True, denc, i > 0, below,
                                       live=live if packed else None)
            ctx["layers"][i] = None
            self._wgrad_fence()
        if tr_emb or st.is_trainable("model.decoder.embed_positions.weight"):
            dtok = st.g[emb] if tr_emb else self._scratch_tok()
            dpos = st.g.get("model.decoder.embed_positions.weight")
            if packed:                                        # live rows -> (batch, position); dead rows: zero
                dres_r = ops.zeros((R, D), dres.dtype)
                ops.scatter_rows(dres[:Rv], live.idx, dres_r)
                dres = dres_r
            ops.embed_bwd(dres, ctx["ids"], dtok, dpos)
        return denc

    def _scratch_tok(self):
        if not hasattr(self, "_stok"):
            self._stok = self.ops.zeros((self.dims.vocab, self.dims.d_model), torch.float32)
        return self._stok

    def param_range(self, prefix):
        """[lo, hi) of the flat buffers covered by the trainable parameters whose name starts with `prefix`."""
        st = self.st
        offs = [(o, o + _rup(int(torch.tensor(shape).prod()), 64)) for n, (o, shape, _) in st.entries.items()
                if n.startswith(prefix) and o >= st.train_start]
        if not offs:
            return None
        return min(a for a, _ in offs), max(b for _, b in offs)

    def backward_encoder(self, ctx, denc, accumulate=False, on_ready=None):
        """denc: fp32 [R, D] gradient w.r.t. the encoder output (encoder_last_hidden_state).  on_ready(lo, hi) is
        called as soon as the gradients of a flat-buffer range are final (layer by layer, top down) so that the
        data-parallel all-reduce of that range can start while the layers below are still in their backward."""
        ops, st, d = self.ops, self.st, self.dims
        self._accumulate = accumulate
        B, T, R, D = ctx["B"], ctx["T"], ctx["R"], d.d_model
        L, R1 = T // 2, B * T
        dyf = self.act(R, D)
        ops.cast_bf16(denc, out=dyf[:R])
        nl = d.enc_layers
        dres, dy = self._ln_bwd("model.encoder.layer_norm", dyf, ctx["x_final"], ctx["mu"], ctx["rs"], None, R,
                                emit=True, colsum_to=self._bias_grad(f"model.encoder.layers.{nl - 1}.fc2.bias"))
        for i in reversed(range(nl)):
            below = self._bias_grad(f"model.encoder.layers.{i - 1}.fc2.bias") if i > 0 else None
            dres, dy = self._layer_bwd(f"model.encoder.layers.{i}", ctx["layers"][i], dres, dy, B, L, 0, False, None,
                                       i > 0, below)
            ctx["layers"][i] = None
            self._wgrad_fence()
            if on_ready is not None:
                # everything above layer i-1's last parameter is final (fc2.bias of layer i-1 still receives the
                # column sums emitted by this layer's last LayerNorm backward, so the cut is at layer i's first entry)
                rng = self.param_range(f"model.encoder.layers.{i}.")
                if rng is not None:
                    hi = self._enc_ready_hi if hasattr(self, "_enc_ready_hi") and self._enc_ready_hi else \
                        self.param_range("model.encoder.layer_norm.")[1]
                    on_ready(rng[0], hi)
                    self._enc_ready_hi = rng[0]
        # conv stem: x0 = gelu(conv2(a1)) + pos ; a1 = gelu(conv1(mel))
        dz2 = self.act(R, D)
        ops.gelu_bwd(dres, ctx["z2"], out=dz2[:R])
        gw2 = ops.zeros((D, 3 * D), torch.float32)
        ops.gemm(dz2, ctx["xcol2"], trans_a=True, trans_b=True, out_dtype=torch.float32, out=gw2, atomic_acc=True)
        ops.unpack_conv_grad(gw2, st.g["model.encoder.conv2.weight"], True)
        ops.colsum(dz2[:R], st.g["model.encoder.conv2.bias"], accumulate=True)
        dxcol2 = ops.gemm(dz2[:R], st.conv2_packed, trans_b=True)
        dz1 = self.act(R1, D)
        ops.col2im_s2_gelu_bwd(dxcol2, ctx["z1"], B, T, out=dz1[:R1])
        gw1 = ops.zeros((D, st.kpad1), torch.float32)
        ops.gemm(dz1, ctx["xcol1"], trans_a=True, trans_b=True, out_dtype=torch.float32, out=gw1, atomic_acc=True)
        ops.unpack_conv_grad(gw1, st.g["model.encoder.conv1.weight"], True)
        ops.colsum(dz1[:R1], st.g["model.encoder.conv1.bias"], accumulate=True)
        if on_ready is not None:
            hi = self._enc_ready_hi if getattr(self, "_enc_ready_hi", None) else st.dec_start
            on_ready(st.train_start, hi)
        self._enc_ready_hi = None

    def zero_small_grads(self, skip_weights=False):
        """Bias / LayerNorm / embedding gradients are accumulated with atomics: zero the gradient buffer's trainable
        range before a (non-accumulating) backward.  skip_weights: the caller runs the backward with wgrad_overwrite, so
        the layers' weight matrices (97 % of the range) are stored by their GEMMs and only the accumulated ranges are
        cleared (ParamStore.small_grad_views, one multi-tensor fill)."""
        st = self.st
        if st.G is None:
            return
        if skip_weights and st.small_grad_views:
            torch._foreach_zero_(st.small_grad_views)
        else:
            st.G[st.train_start:st.train_end].zero_()
