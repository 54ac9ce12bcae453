"""Drop-in module surface of the reference path over the MI355X engine.

The reference scripts touch exactly two `transformers` classes on the hot path:
  * `WhisperFeatureExtractor.__call__(audio, sampling_rate=...) -> .input_features`
    (run_distillation.py:1176,1234; run_eval.py:628-642; TF:feature_extraction_whisper.py:193-346);
  * `WhisperForConditionalGeneration.forward(input_features=, decoder_input_ids=, labels=)` or
    `forward(encoder_outputs=, labels=)` -> `.loss`, `.logits`, `.encoder_last_hidden_state`, followed by
    `loss.backward()` (run_distillation.py:1472-1484, 1609; TF:modeling_whisper.py:994-1099).
The classes below keep those names, argument meanings, parameter names/shapes (HF `state_dict` keys, tied
`proj_out`), `nn.LayerNorm` instance types (the weight-decay grouping of run_distillation.py:1386-1391 depends on
them) and error behaviour, but every tensor operation runs in the HIP kernels through `WhisperEngine`.
Parameters are views into the engine's flat fp32 master buffer, so `optimizer.step()`, `save_pretrained`-style
`state_dict()`, `load_state_dict()` and DistributedDataParallel all work on them unchanged.
"""
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .engine import ParamStore, WhisperDims, WhisperEngine
from .lazy_logits import LazyLogits, LazyState, _CEFn, _FusedFn, lazy_backward
from .student_init import mel_filter_bank, random_state_dict


def _default_ops(device):
    from .ops_hip import HipOps  # raises when the HIP library or the GPU is missing: there is no CPU fallback
    return HipOps(device)


@dataclass
class Seq2SeqLMOutput:
    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    encoder_last_hidden_state: Optional[torch.Tensor] = None
    # handles for the fused loss kernels (not part of the reference surface): the engine's low-precision logits
    # [rows >= B*T, padded vocabulary] behind `.logits`, and the model that produced them
    _logits_lowp: Optional[torch.Tensor] = None
    _model: Optional[object] = None
    _rows: Optional[object] = None      # _RowSel of a forward called with valid_len (rows of _logits_lowp), else None
    _lazy: Optional[object] = None      # lazy_logits.LazyState shared by `.logits`, `.loss` and the engine node's backward


@dataclass
class BaseModelOutput:
    last_hidden_state: torch.Tensor = None

    def __getitem__(self, i):
        return (self.last_hidden_state,)[i]


class BatchFeature(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class WhisperFeatureExtractor:
    """Log-mel front end with the call surface of TF:feature_extraction_whisper.py:193-346, computed on the GPU."""

    def __init__(self, feature_size=80, sampling_rate=16000, hop_length=160, chunk_length=30, n_fft=400,
                 padding_value=0.0, ops=None, device="cuda:0"):
        if n_fft != 400 or hop_length != 160:
            raise ValueError("the HIP log-mel kernel implements Whisper's n_fft=400 / hop_length=160")
        self.feature_size, self.sampling_rate = feature_size, sampling_rate
        self.hop_length, self.chunk_length, self.n_fft = hop_length, chunk_length, n_fft
        self.n_samples = chunk_length * sampling_rate
        self.nb_max_frames = self.n_samples // hop_length
        self.padding_value = padding_value
        self.mel_filters = mel_filter_bank(feature_size)  # [201, n_mels] float64, as the reference attribute
        self.ops = ops if ops is not None else _default_ops(device)
        self._filt = torch.tensor(self.mel_filters, dtype=torch.float32, device=self.ops.device).contiguous()

    def __call__(self, raw_speech, truncation=True, pad_to_multiple_of=None, return_tensors=None,
                 return_attention_mask=None, padding="max_length", max_length=None, sampling_rate=None,
                 do_normalize=None, device=None, **kwargs):
        if sampling_rate is not None and sampling_rate != self.sampling_rate:
            raise ValueError(
                f"The model corresponding to this feature extractor: {self.__class__.__name__} was trained using a "
                f"sampling rate of {self.sampling_rate}. Please make sure that the provided `raw_speech` input was "
                f"sampled with {self.sampling_rate} and not {sampling_rate}.")
        if isinstance(raw_speech, (np.ndarray, torch.Tensor)) and getattr(raw_speech, "ndim", 1) == 1:
            raw_speech = [raw_speech]
        n = max_length if max_length else self.n_samples
        if n % self.hop_length:
            raise ValueError("max_length must be a multiple of hop_length")
        batch = torch.full((len(raw_speech), n), float(self.padding_value), dtype=torch.float32)
        mask = torch.zeros((len(raw_speech), n), dtype=torch.int32)
        for i, w in enumerate(raw_speech):
            w = torch.as_tensor(np.asarray(w, dtype=np.float32)).reshape(-1)
            if w.numel() > n:
                if not truncation:
                    raise ValueError("clips longer than max_length need truncation=True (chunk long audio first)")
                w = w[:n]
            batch[i, : w.numel()] = w
            mask[i, : w.numel()] = 1
        feats = self.ops.logmel(batch.to(self.ops.device), self._filt)
        out = BatchFeature()
        if return_tensors == "pt":
            out["input_features"] = feats
        elif return_tensors == "np" or return_tensors is None:
            arr = feats.cpu().numpy()
            out["input_features"] = arr if return_tensors == "np" else [a for a in arr]
        else:
            raise ValueError(f"unsupported return_tensors={return_tensors}")
        if return_attention_mask:
            out["attention_mask"] = mask[:, :: self.hop_length]
        return out


    def pad(self, processed_features, padding=True, max_length=None, truncation=False, pad_to_multiple_of=None,
            return_attention_mask=None, return_tensors=None):
        """`SequenceFeatureExtractor.pad` as the reference's collator calls it (run_distillation.py:447-451:
        `pad({"input_features": [M x 3000 arrays]}, padding="max_length", return_tensors="pt")`).  The base class
        pads along the FIRST axis of every item (the mel-bin axis for Whisper features, which is equal for all items),
        so the call (`padding="longest"` at run_distillation.py:1421) amounts to stacking the already fixed-length features into one float32 batch; items whose first
        axes differ are padded with `padding_value` up to the longest (or `max_length`)."""
        if isinstance(processed_features, (list, tuple)):
            processed_features = {k: [f[k] for f in processed_features] for k in processed_features[0]}
        if "input_features" not in processed_features:
            raise ValueError("You should supply an instance of `BatchFeature` or list of `BatchFeature` to this "
                             f"method that includes input_features, but you provided {list(processed_features.keys())}")
        items = [torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x).to(torch.float32)
                 for x in processed_features["input_features"]]
        out = BatchFeature()
        if len(items) == 0:
            out["input_features"] = []
            return out
        do_pad = padding not in (False, "do_not_pad")
        if padding == "max_length" and max_length is None:
            raise ValueError("When setting ``padding=max_length``, make sure that max_length is defined")
        n0 = max(x.shape[0] for x in items)
        if do_pad and padding == "max_length" and max_length is not None:
            n0 = max_length
        if do_pad and pad_to_multiple_of:
            n0 = -(-n0 // pad_to_multiple_of) * pad_to_multiple_of
        rows, masks = [], []
        for x in items:
            if truncation and max_length is not None and x.shape[0] > max_length:
                x = x[:max_length]
            m = torch.ones(x.shape[0], dtype=torch.int32)
            if do_pad and x.shape[0] < n0:
                fill = torch.full((n0 - x.shape[0],) + tuple(x.shape[1:]), float(self.padding_value))
                x = torch.cat([x, fill], 0)
                m = torch.cat([m, torch.zeros(n0 - m.numel(), dtype=torch.int32)])
            rows.append(x)
            masks.append(m)
        if return_tensors in ("pt", "np"):
            if any(r.shape != rows[0].shape for r in rows):
                raise ValueError("Unable to convert output 'input_features' to tensor: the items have different "
                                 "shapes. Use padding=True to ensure all outputs have the same length.")
            batch = torch.stack(rows)
            out["input_features"] = batch if return_tensors == "pt" else batch.numpy()
            if return_attention_mask:
                am = torch.stack(masks)
                out["attention_mask"] = am if return_tensors == "pt" else am.numpy()
        elif return_tensors is None:
            out["input_features"] = [r.numpy() for r in rows]
            if return_attention_mask:
                out["attention_mask"] = [m.numpy() for m in masks]
        else:
            raise ValueError(f"unsupported return_tensors={return_tensors}")
        return out

    # -- checkpoint directory round trip (run_distillation.py:966, 1071, 1641) ------------------------------------------
    def to_dict(self):
        return {"feature_extractor_type": "WhisperFeatureExtractor", "feature_size": self.feature_size,
                "sampling_rate": self.sampling_rate, "hop_length": self.hop_length, "chunk_length": self.chunk_length,
                "n_fft": self.n_fft, "padding_value": self.padding_value, "n_samples": self.n_samples,
                "nb_max_frames": self.nb_max_frames, "padding_side": "right", "return_attention_mask": False}

    def save_pretrained(self, save_directory, **kwargs):
        import json
        import os
        os.makedirs(save_directory, exist_ok=True)
        with open(os.path.join(save_directory, "preprocessor_config.json"), "w") as f:
            json.dump(self.to_dict(), f, indent=2, sort_keys=True)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, ops=None, device="cuda:0", **kwargs):
        import json
        import os
        path = os.path.join(str(pretrained_model_name_or_path), "preprocessor_config.json")
        if not os.path.exists(path):
            raise OSError(f"{path} not found (no hub access in this environment: pass a local checkpoint directory)")
        with open(path) as f:
            d = json.load(f)
        keys = ("feature_size", "sampling_rate", "hop_length", "chunk_length", "n_fft", "padding_value")
        return cls(**{k: d[k] for k in keys if k in d}, ops=ops, device=device)


class _EncoderLayer(nn.Module):
    def __init__(self, D, Fd):
        super().__init__()
        self.self_attn = _Attention(D)
        self.self_attn_layer_norm = nn.LayerNorm(D, device="meta")
        self.fc1 = nn.Linear(D, Fd, device="meta")
        self.fc2 = nn.Linear(Fd, D, device="meta")
        self.final_layer_norm = nn.LayerNorm(D, device="meta")


class _DecoderLayer(nn.Module):
    def __init__(self, D, Fd):
        super().__init__()
        self.self_attn = _Attention(D)
        self.self_attn_layer_norm = nn.LayerNorm(D, device="meta")
        self.encoder_attn = _Attention(D)
        self.encoder_attn_layer_norm = nn.LayerNorm(D, device="meta")
        self.fc1 = nn.Linear(D, Fd, device="meta")
        self.fc2 = nn.Linear(Fd, D, device="meta")
        self.final_layer_norm = nn.LayerNorm(D, device="meta")


class _Attention(nn.Module):
    def __init__(self, D):
        super().__init__()
        self.k_proj = nn.Linear(D, D, bias=False, device="meta")
        self.v_proj = nn.Linear(D, D, device="meta")
        self.q_proj = nn.Linear(D, D, device="meta")
        self.out_proj = nn.Linear(D, D, device="meta")


class _Encoder(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.conv1 = nn.Conv1d(d.n_mels, d.d_model, 3, padding=1, device="meta")
        self.conv2 = nn.Conv1d(d.d_model, d.d_model, 3, stride=2, padding=1, device="meta")
        self.embed_positions = nn.Embedding(d.max_src, d.d_model, device="meta")
        self.layers = nn.ModuleList([_EncoderLayer(d.d_model, d.ffn) for _ in range(d.enc_layers)])
        self.layer_norm = nn.LayerNorm(d.d_model, device="meta")
        self.gradient_checkpointing = False     # read by run_distillation.py:1021 / 1037 before anyone enables it


class _Decoder(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.embed_tokens = nn.Embedding(d.vocab, d.d_model, device="meta")
        self.embed_positions = nn.Embedding(d.max_tgt, d.d_model, device="meta")
        self.layers = nn.ModuleList([_DecoderLayer(d.d_model, d.ffn) for _ in range(d.dec_layers)])
        self.layer_norm = nn.LayerNorm(d.d_model, device="meta")
        self.gradient_checkpointing = False


class _Model(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.encoder = _Encoder(d)
        self.decoder = _Decoder(d)


class WhisperConfig:
    """The fields of `transformers.WhisperConfig` this path reads (TF:configuration_whisper.py:127-164), loadable from
    and writable to a checkpoint directory's config.json.  A `transformers.WhisperConfig` works in its place."""

    model_type = "whisper"
    _DEFAULTS = dict(vocab_size=51865, num_mel_bins=80, encoder_layers=4, encoder_attention_heads=6, decoder_layers=4,
                     decoder_attention_heads=6, decoder_ffn_dim=1536, encoder_ffn_dim=1536, d_model=384,
                     max_source_positions=1500, max_target_positions=448, pad_token_id=50256, bos_token_id=50256,
                     eos_token_id=50256, decoder_start_token_id=50257, activation_function="gelu", dropout=0.0,
                     attention_dropout=0.0, activation_dropout=0.0, encoder_layerdrop=0.0, decoder_layerdrop=0.0,
                     scale_embedding=False, use_cache=True, suppress_tokens=None, begin_suppress_tokens=None,
                     max_length=448, forced_decoder_ids=None)

    def __init__(self, **kw):
        for k, v in self._DEFAULTS.items():
            setattr(self, k, v)
        for k, v in kw.items():
            setattr(self, k, v)
        if self.encoder_attention_heads != self.decoder_attention_heads or self.encoder_ffn_dim != self.decoder_ffn_dim:
            raise ValueError("the MI355X engine expects equal encoder / decoder head counts and FFN widths")
        if self.d_model != 64 * self.encoder_attention_heads:
            raise ValueError("the HIP attention kernel implements head_dim 64 (every Whisper checkpoint)")
        if self.activation_function != "gelu":
            raise ValueError("the HIP GEMM epilogue implements Whisper's exact GELU")
        if any(getattr(self, k) for k in ("dropout", "attention_dropout", "activation_dropout", "encoder_layerdrop",
                                          "decoder_layerdrop")) or self.scale_embedding:
            raise ValueError("dropout / layerdrop / scale_embedding are 0 / False in every Whisper config; the MI355X "
                             "path does not implement them")

    def to_dict(self):
        d = {k: v for k, v in vars(self).items() if not k.startswith("_")}
        d["model_type"] = "whisper"
        d["architectures"] = ["WhisperForConditionalGeneration"]
        return d

    @classmethod
    def from_pretrained(cls, path, **_):
        import json
        import os
        with open(os.path.join(path, "config.json")) as f:
            d = json.load(f)
        for k in ("model_type", "architectures", "transformers_version", "torch_dtype", "dtype", "_name_or_path"):
            d.pop(k, None)
        return cls(**d)

    def save_pretrained(self, path):
        import json
        import os
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(self.to_dict(), f, indent=2, sort_keys=True)


class _RowSel:
    """Which decoder positions a forward computed when it was told the label lengths (`valid_len`, see
    distill.trim_dead_positions): the first `Te` positions of every sequence, or -- per-sequence lengths -- the packed live
    rows (engine.LiveRows over the [B, Te] rectangle).  The engine's low-precision logits hold exactly these rows; the
    reference-shaped fp32 `.logits` [B, T, V] keeps its shape with ZERO rows at the dead positions (labels -100 there:
    the reference's CE ignores them and its KL term masks them, run_distillation.py:1453-1462)."""

    def __init__(self, B, T, Te, live):
        self.B, self.T, self.Te, self.live = B, T, Te, live
        self.n = B * Te if live is None else live.n
        self.idx_full = None
        if live is not None:
            i = live.idx.long()
            self.idx_full = (i // Te) * T + i % Te          # row numbers in the full [B * T] layout

    @classmethod
    def make(cls, valid_len, B, T, device, pack_below=0.9):
        from .distill import _as_int
        from .engine import LiveRows
        valid_len = _as_int(valid_len)
        if valid_len is None:
            return None
        lens = None
        if not isinstance(valid_len, int):
            lens = [max(1, min(T, int(x))) for x in valid_len]
            if len(lens) != B:
                raise ValueError("valid_len: one length per sequence of the batch (or one int for the batch)")
            valid_len = max(lens)
        Te = max(1, min(T, int(valid_len)))
        live = None
        if lens is not None and sum(lens) < pack_below * B * Te:
            live = LiveRows.build(lens, Te, device)
        if Te == T and live is None:
            return None
        return cls(B, T, Te, live)

    def same_as(self, other):
        if other is None or (self.B, self.T, self.Te, self.n) != (other.B, other.T, other.Te, other.n):
            return False
        return (self.live is None) == (other.live is None) and (self.live is None or torch.equal(self.live.idx, other.live.idx))

    def select(self, flat):
        """[B * T, C] in (batch, position) order -> the computed rows [n, C]"""
        if self.live is None:
            return flat.reshape(self.B, self.T, -1)[:, :self.Te].reshape(self.B * self.Te, -1)
        return flat.index_select(0, self.idx_full)

    def expand(self, rows, V):
        """computed rows [n, >= V] -> fp32 [B, T, V], zeros at the positions that were not computed"""
        B, T, Te = self.B, self.T, self.Te
        out = torch.zeros(B, T, V, dtype=torch.float32, device=rows.device)
        if self.live is None:
            out[:, :Te] = rows[: B * Te, :V].reshape(B, Te, V)
        else:
            out.view(B * T, V).index_copy_(0, self.idx_full, rows[: self.n, :V].float())
        return out


class _PendingLens:
    """The label lengths of a batch on their way to the host (`WhisperForConditionalGeneration.skip_dead_positions`).

    The reference hands the model `labels` padded to 448 with -100 (run_distillation.py:405-478) and computes all 447
    decoder positions of every row; positions behind a row's last label are dead (distill.trim_dead_positions: same loss,
    same gradients without them -- 71 % of the benchmark's decoder positions).  Leaving them out needs the lengths ON THE
    HOST (they size the launches), and the drop-in collator can put them into the batch (`report_valid_len`) -- but that is
    an edit of the script.  Without it the lengths are read back from the device tensor the model is given: a tiny reduction
    and a non-blocking copy into pinned memory are enqueued FIRST, the whole encoder forward (which does not depend on them)
    is enqueued next, and only then the host waits for the copy's event -- by which time the device has >= 100 ms of encoder
    work queued, so the wait costs the host its run-ahead over the previous step and the device nothing.  The second model
    called with the same `labels` tensor (the teacher: `teacher_model(**batch)` / `(encoder_outputs=..., labels=...)`) finds
    the lengths in a one-entry cache keyed by the tensor's address, shape and version: no second round trip."""

    _cache = {}

    def __init__(self, labels):
        self.key = (labels.data_ptr(), tuple(labels.shape), labels._version, str(labels.device))
        self.lens = _PendingLens._cache.get(self.key)
        self.event = None
        if self.lens is None:
            T = labels.shape[1]
            pos = torch.arange(1, T + 1, device=labels.device)
            lens_dev = ((labels != -100) * pos).amax(dim=1)
            if labels.is_cuda:
                self.host = torch.empty(labels.shape[0], dtype=torch.int64, pin_memory=True)
                self.host.copy_(lens_dev, non_blocking=True)
                self.event = torch.cuda.Event()
                self.event.record()
            else:
                self.host = lens_dev

    def resolve(self):
        if self.lens is None:
            if self.event is not None:
                self.event.synchronize()
            self.lens = [max(1, int(x)) for x in self.host.tolist()]
            _PendingLens._cache.clear()
            _PendingLens._cache[self.key] = self.lens
        return self.lens


class _EngineFn(torch.autograd.Function):
    """forward: engine encode+decode with activations kept; backward: engine backward from d(loss)/d(logits) (and
    optionally d/d(encoder_last_hidden_state)); parameter gradients are returned as views of the flat buffer."""

    @staticmethod
    def forward(ctx, model, input_features, enc_in, decoder_input_ids, sel, *params):
        eng = model.engine
        train = any(ctx.needs_input_grad[5:])  # (grad mode is off inside Function.forward; this is the real signal)
        enc_train = train and model._encoder_requires_grad()
        ectx = None
        if enc_in is None:
            enc, ectx = eng.encode(input_features.to(torch.float32).contiguous(), save=enc_train)
        else:
            rows = enc_in.shape[0] * enc_in.shape[1]
            enc = eng.act(rows, eng.dims.d_model)
            enc[:rows].copy_(enc_in.reshape(rows, -1))
        B, T = decoder_input_ids.shape
        V = eng.dims.vocab
        if isinstance(sel, _PendingLens):       # label lengths read back from the device: the encoder is queued, now wait
            sel = _RowSel.make(sel.resolve(), B, T, decoder_input_ids.device)
        model._last_sel = sel
        if sel is None:
            logits, dctx = eng.decode(decoder_input_ids.contiguous(), enc, save=train)
        else:
            # the dead decoder positions are left out (same loss, same gradients: distill.trim_dead_positions)
            logits, dctx = eng.decode(decoder_input_ids[:, :sel.Te].contiguous(), enc, save=train, live=sel.live)
        # `.logits` keeps the reference's type and shape -- fp32 [B, T, V], accelerate upcasts model outputs to fp32 -- but the
        # allocation is left UNFILLED: the values stay in the engine's bf16 buffer until something other than the reference's
        # own loss expression reads them (lazy_logits.LazyLogits; the wrapper is put on by `forward`, which owns the state)
        out_logits = torch.empty((B, T, V), dtype=torch.float32, device=logits.device)
        state = LazyState(model, logits, sel, (B, T, V), train)
        ctx.set_materialize_grads(False)
        ctx.model, ctx.ectx, ctx.dctx, ctx.shape, ctx.sel = model, ectx, dctx, (B, T, V), sel
        ctx.logits_buf = logits if train else None
        ctx.lazy = state
        model._last_logits_lowp = logits
        model._last_lazy_state = state
        Re = B * eng.dims.max_src
        enc_out = enc[:Re].float().view(B, eng.dims.max_src, -1)
        ctx.mark_non_differentiable(enc_out)
        return out_logits, enc_out

    @staticmethod
    def backward(ctx, g_logits, _g_enc):
        model, eng = ctx.model, ctx.model.engine
        B, T, V = ctx.shape
        st = eng.st
        buf = ctx.logits_buf
        if not lazy_backward(ctx.lazy, buf, g_logits):
            # no loss node left its upstream gradient behind: the caller differentiated through the materialised tensor
            buf.zero_()
            if g_logits is not None:
                if ctx.sel is None:
                    buf[: B * T, :V].copy_(g_logits.reshape(B * T, V))
                else:
                    buf[: ctx.sel.n, :V].copy_(ctx.sel.select(g_logits.reshape(B * T, V)))
        # (the flat buffer is this backward's scratch: cleared here, so the layers' weight gradients are stored, not added)
        # (not when this backward stops at `encoder_outputs` while the encoder is trainable: its weight gradients must read zero)
        every_weight_written = ctx.ectx is not None or not st.is_trainable("model.encoder.layers.0.fc1.weight")
        eng.zero_small_grads(skip_weights=every_weight_written)
        eng.wgrad_overwrite = True
        try:
            denc = eng.backward_decoder(ctx.dctx, buf, want_denc=ctx.ectx is not None)
            if ctx.ectx is not None:
                eng.backward_encoder(ctx.ectx, denc)
        finally:
            eng.wgrad_overwrite = False
        # copies of the flat gradient buffer's ranges: the buffer is zeroed and rewritten by the next backward, so a
        # returned view would change under torch.autograd.grad results, tensor hooks or DDP bucket views that keep it.
        # (AccumulateGrad takes ownership of a fresh non-view gradient without copying it again: still one copy.)
        # Parameters frozen after construction (`requires_grad_(False)`, run_distillation.py:1018-1040) get None.
        # ONE copy of the trainable range (a single launch, not ~600 five-microsecond ones on a host that is the bottleneck of the
        # eager loop); every returned gradient is a contiguous view of that fresh buffer, which nothing else writes.
        lo, hi = st.train_start, st.train_end
        Gc = st.G[lo:hi].clone() if hi > lo else None
        grads = []
        for name, p in zip(model._param_names, model._param_list):
            if p.requires_grad and name in st.g:
                off = st.entries[name][0]
                grads.append(Gc[off - lo: off - lo + p.numel()].view(p.shape))
            else:
                grads.append(None)
        return (None, None, None, None, None, *grads)


def fused_distillation_loss(student_outputs, teacher_outputs, labels, temperature=2.0, kl_weight=1.0, ce_weight=0.8):
    """One-call replacement for lines 1486-1493 of run_distillation.py (`ce_loss`, the two softmaxes, `kl_divergence`,
    `loss = 0.8 * ce_loss + kl_weight * kl_loss`) when both models are `distil_whisper_amd` modules: the fused HIP loss
    kernel reads the two engines' bf16 logits once.  Returns (loss, {"loss", "ce_loss", "kl_loss"}) with the metrics
    detached, like the reference's `metrics` dict (1494-1495).  `loss.backward()` then runs the student's backward."""
    s_buf, t_buf = student_outputs._logits_lowp, teacher_outputs._logits_lowp
    if s_buf is None or t_buf is None:
        raise ValueError("fused_distillation_loss needs outputs of distil_whisper_amd.WhisperForConditionalGeneration")
    model = student_outputs._model
    sel = student_outputs._rows
    t_sel = teacher_outputs._rows
    if sel is not None and t_sel is None:
        # The reference's shared-encoder call `teacher_model(encoder_outputs=..., labels=...)` (run_distillation.py:1478)
        # carries no `valid_len`: the teacher computed all B * T rows.  Take the student's rows out of them (a row gather of
        # the teacher's low-precision logits: B * Te * V * 2 B of traffic, against a step-0 ValueError before).
        BT = sel.B * sel.T
        if t_buf.shape[0] < BT:
            raise ValueError("fused_distillation_loss: the teacher's logits do not cover the student's batch")
        t_buf = sel.select(t_buf[:BT]).contiguous()
    elif (sel is None) != (t_sel is None) or (sel is not None and not sel.same_as(t_sel)):
        raise ValueError("fused_distillation_loss: student and teacher outputs were computed over different decoder "
                         "positions (pass the same valid_len to both forwards, or none to the teacher)")
    state = student_outputs._lazy
    B, T = labels.shape
    R = state.rows
    lab = state.select_rows(labels.reshape(B * T, 1)).reshape(-1).contiguous()
    t_rows = t_buf[:R]
    loss, losses = _FusedFn.apply(student_outputs.logits.as_subclass(torch.Tensor), state, t_rows, lab, float(temperature),
                                  float(ce_weight), float(kl_weight))
    return loss, {"loss": losses[2], "ce_loss": losses[0], "kl_loss": losses[1]}


class WhisperForConditionalGeneration(nn.Module):
    """dtype=torch.float32 (default): fp32 master weights, bf16 GEMM operands, fp32 residual stream -- the reference's
    student under `accelerate` bf16 autocast (SURVEY.md section 8a').  dtype=torch.bfloat16: weights rounded to bf16
    and a bf16 residual stream -- a model loaded with `torch_dtype=torch.bfloat16` (the teacher of
    run_distillation.py:986-1004, every model of run_eval.py / run_pseudo_labelling.py); inference only."""

    # `forward(..., labels=...)` without `valid_len`: read the label lengths back from `labels` and leave the dead decoder
    # positions out (_PendingLens).  Same loss and gradients under the reference's loss lines (CE ignores -100, the KL term is
    # masked by labels >= 0); `.logits` then has ZERO rows behind each row's last label, which a caller that looks at padded
    # positions would see -- so it is OFF unless asked for: `DW_SKIP_DEAD_POSITIONS=1` in the environment of the unedited
    # script, or `model.skip_dead_positions = True`.
    skip_dead_positions = os.environ.get("DW_SKIP_DEAD_POSITIONS", "0") not in ("", "0", "false", "False")

    def __init__(self, config, ops=None, device="cuda:0", state_dict=None, seed=0, frozen_prefixes=(),
                 dtype=torch.float32):
        super().__init__()
        self.config = config
        self.dims = WhisperDims.from_any(config)
        self.ops = ops if ops is not None else _default_ops(device)
        if dtype not in (torch.float32, torch.bfloat16, None):
            raise ValueError(f"dtype {dtype}: the MI355X path computes in bf16 with fp32 or bf16 parameters")
        self.dtype_mode = torch.float32 if dtype is None else dtype
        pure_bf16 = self.dtype_mode == torch.bfloat16
        sd = state_dict if state_dict is not None else random_state_dict(self.dims, seed, device=self.ops.device)
        self.store = ParamStore(self.ops, self.dims, sd, trainable=not pure_bf16,
                                frozen_prefixes=tuple(frozen_prefixes), round_bf16=pure_bf16)
        self.engine = WhisperEngine(self.ops, self.store, self.ops.lowp if pure_bf16 else torch.float32)
        self._decoders = {}
        self.model = _Model(self.dims)
        self.proj_out = nn.Linear(self.dims.d_model, self.dims.vocab, bias=False, device="meta")
        # re-point every parameter of the (meta) module tree at its view in the flat master buffer
        for name in self.store.real_names():
            mod_path, pname = name.rsplit(".", 1)
            mod = self.get_submodule(mod_path)
            p = nn.Parameter(self.store.p[name], requires_grad=self.store.is_trainable(name))
            setattr(mod, pname, p)
        self.proj_out.weight = self.model.decoder.embed_tokens.weight  # tied (TF:modeling_whisper.py:965)
        self._param_names = self.store.real_names()
        self._param_list = [self.get_parameter(n) for n in self._param_names]
        self._enc_params = [p for n, p in zip(self._param_names, self._param_list) if n.startswith("model.encoder.")]
        self._versions = None
        from .generation import GenerationConfig
        self.generation_config = GenerationConfig.from_model_config(config)
        if self.generation_config.decoder_start_token_id is None:
            self.generation_config.decoder_start_token_id = self.dims.decoder_start_token_id
        self.is_gradient_checkpointing = False

    # -- reference surface ------------------------------------------------------------------------------------------
    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        """`nn.Module.state_dict()` with INDEPENDENT tensors.  Every parameter of this module is a view of one flat
        buffer (engine.ParamStore); serializers that look at storages -- safetensors behind `accelerator.save_state`
        (run_distillation.py:1636) -- take tensors that share a storage for aliases of one another, drop all but one and
        then refuse the rest ("None is covering the entire storage").  So the entries are copies, and the tied
        `proj_out.weight` IS `model.decoder.embed_tokens.weight` (one tensor under two names, as `transformers` hands it
        out).  `keep_vars=True` returns the live parameters (views) unchanged."""
        sd = super().state_dict(*args, destination=destination, prefix=prefix, keep_vars=keep_vars)
        if keep_vars:
            return sd
        mine = [k for k in sd if k.startswith(prefix)]
        for k in mine:
            sd[k] = sd[k].clone()
        tied, emb = prefix + "proj_out.weight", prefix + "model.decoder.embed_tokens.weight"
        if tied in sd and emb in sd:
            sd[tied] = sd[emb]
        return sd

    def get_encoder(self):
        return self.model.encoder

    def get_decoder(self):
        return self.model.decoder

    def _encoder_requires_grad(self):
        return any(p.requires_grad for p in self._enc_params)

    def freeze_encoder(self):
        """`student_model.freeze_encoder()` (run_distillation.py:1018-1021; TF:modeling_whisper.py `freeze_encoder` ->
        `encoder._freeze_parameters()`): gradients of the encoder are disabled.  The engine then neither keeps the
        encoder's activations nor runs its backward; the same happens for any parameter the caller switches off with
        `requires_grad_(False)` afterwards (run_distillation.py:1034-1040 `freeze_embed_positions`)."""
        for p in self._enc_params:
            p.requires_grad_(False)
        self.model.encoder._requires_grad = False

    def gradient_checkpointing_enable(self, gradient_checkpointing_kwargs=None):
        """run_distillation.py:1014-1015.  The engine keeps every activation it needs in HBM (288 GB: the full
        distil-large-v3 step at batch 32 peaks at 94 GiB) and never recomputes, so this only records the request."""
        self.is_gradient_checkpointing = True
        self.model.encoder.gradient_checkpointing = True
        self.model.decoder.gradient_checkpointing = True

    def gradient_checkpointing_disable(self):
        self.is_gradient_checkpointing = False
        self.model.encoder.gradient_checkpointing = False
        self.model.decoder.gradient_checkpointing = False

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *model_args, config=None, torch_dtype=None, dtype=None,
                        attn_implementation=None, low_cpu_mem_usage=None, cache_dir=None, revision=None, token=None,
                        subfolder="", variant=None, use_safetensors=None, local_files_only=None, ops=None,
                        device="cuda:0", **kwargs):
        """Load `config.json` + `model.safetensors` (or `pytorch_model.bin`) from a LOCAL checkpoint directory
        (run_distillation.py:986-1004, run_eval.py:547-563 call this with hub names; there is no network here, so a
        name that is not a directory raises).  `attn_implementation` is accepted for the CLI whitelist
        (run_distillation.py:141-148): this class always runs the HIP flash-attention kernel."""
        import json
        import os
        if kwargs:
            raise TypeError(f"from_pretrained() got unexpected keyword arguments {sorted(kwargs)}")
        path = os.path.join(str(pretrained_model_name_or_path), subfolder) if subfolder else \
            str(pretrained_model_name_or_path)
        if not os.path.isdir(path):
            raise OSError(f"{path} is not a local checkpoint directory (no hub access in this environment)")
        if attn_implementation not in (None, "eager", "sdpa", "flash_attention_2", "hip_attention"):
            raise ValueError(f"unknown attn_implementation {attn_implementation!r}")
        if config is None:
            config = WhisperConfig.from_pretrained(path)
        dt = dtype if dtype is not None else torch_dtype
        if isinstance(dt, str):
            dt = {"float32": torch.float32, "bfloat16": torch.bfloat16, "auto": None}.get(dt, dt)
        st_path = os.path.join(path, "model.safetensors" if not variant else f"model.{variant}.safetensors")
        bin_path = os.path.join(path, "pytorch_model.bin")
        if os.path.exists(st_path) and use_safetensors is not False:
            from safetensors.torch import load_file
            sd = load_file(st_path)
        elif os.path.exists(bin_path):
            sd = torch.load(bin_path, map_location="cpu", weights_only=True)
        else:
            raise OSError(f"no model.safetensors / pytorch_model.bin under {path}")
        if "model.decoder.embed_tokens.weight" not in sd and "proj_out.weight" in sd:
            sd["model.decoder.embed_tokens.weight"] = sd["proj_out.weight"]
        model = cls(config, ops=ops, device=device, state_dict=sd, dtype=dt)
        gpath = os.path.join(path, "generation_config.json")
        if os.path.exists(gpath):
            from .generation import GenerationConfig
            with open(gpath) as f:
                model.generation_config = GenerationConfig.from_any(json.load(f))
        return model

    def save_pretrained(self, save_directory, state_dict=None, safe_serialization=True, **kwargs):
        """run_distillation.py:1765, 1791: config.json, generation_config.json and the weights under their HF names
        (the tied `proj_out.weight` is not stored, as `transformers` does for tied weights)."""
        import json
        import os
        os.makedirs(save_directory, exist_ok=True)
        cfg = self.config
        if hasattr(cfg, "save_pretrained"):
            cfg.save_pretrained(save_directory)
        else:
            WhisperConfig(**{k: getattr(cfg, k) for k in WhisperConfig._DEFAULTS if hasattr(cfg, k)}) \
                .save_pretrained(save_directory)
        with open(os.path.join(save_directory, "generation_config.json"), "w") as f:
            json.dump(self.generation_config.to_dict(), f, indent=2, sort_keys=True)
        sd = state_dict if state_dict is not None else self.state_dict()
        sd = {k: v.detach().to("cpu").contiguous() for k, v in sd.items() if k != "proj_out.weight"}
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(sd, os.path.join(save_directory, "model.safetensors"), metadata={"format": "pt"})
        else:
            torch.save(sd, os.path.join(save_directory, "pytorch_model.bin"))

    def _sync_shadow(self):
        v = tuple(p._version for p in self._param_list)
        if v != self._versions:
            self.store.refresh_shadow()  # fp32 master (possibly just updated by an optimizer) -> bf16 GEMM operands
            self._versions = v

    def forward(self, input_features=None, attention_mask=None, decoder_input_ids=None, labels=None,
                encoder_outputs=None, valid_len=None, **kwargs):
        """`valid_len` (optional, HOST integers from the collator -- DataCollatorSpeechSeq2SeqWithPadding(report_valid_len=...)
        puts them into the batch, so the reference's `student_model(**batch)` / `teacher_model(**batch)` pass them on): one
        int = 1 + the last labelled position of the batch, or one per sequence.  The decoder positions behind them are dead
        (distill.trim_dead_positions) and are not computed; `.logits` keeps the reference shape [B, T, V] with zero rows there,
        `.loss` and the gradients are those of the full-length forward.  Absent: every position is computed."""
        d = self.dims
        if labels is not None:
            if labels.shape[1] > d.max_tgt:
                raise ValueError(f"Labels' sequence length {labels.shape[1]} cannot exceed the maximum allowed length "
                                 f"of {d.max_tgt} tokens.")
            if decoder_input_ids is None:
                from .distill import shift_tokens_right
                decoder_input_ids = shift_tokens_right(labels, d.pad_token_id, d.decoder_start_token_id)
        if decoder_input_ids is None:
            raise ValueError("decoder_input_ids or labels are required")
        enc_in = None
        if encoder_outputs is not None:
            enc_in = encoder_outputs[0] if not torch.is_tensor(encoder_outputs) else encoder_outputs
        elif input_features is None:
            raise ValueError("input_features or encoder_outputs are required")
        elif input_features.shape[-1] != 2 * d.max_src:
            raise ValueError(f"Whisper expects the mel input features to be of length {2 * d.max_src}, but found "
                             f"{input_features.shape[-1]}. Make sure to pad the input mel features to {2 * d.max_src}.")
        self._sync_shadow()
        if valid_len is None and labels is not None and self.skip_dead_positions and labels.shape == decoder_input_ids.shape:
            sel = _PendingLens(labels)           # (resolved inside _EngineFn.forward, behind the encoder's launches)
            if sel.lens is not None:             # the other model already fetched them for this very tensor
                sel = _RowSel.make(sel.lens, decoder_input_ids.shape[0], decoder_input_ids.shape[1], decoder_input_ids.device)
        else:
            sel = _RowSel.make(valid_len, decoder_input_ids.shape[0], decoder_input_ids.shape[1], decoder_input_ids.device)
        logits, enc = _EngineFn.apply(self, input_features, enc_in, decoder_input_ids, sel, *self._param_list)
        sel, self._last_sel = self._last_sel, None
        lowp, self._last_logits_lowp = self._last_logits_lowp, None
        state, self._last_lazy_state = self._last_lazy_state, None
        loss = None
        if labels is not None:
            # token-mean CE over labels != -100 (TF:modeling_whisper.py:1083-1087) by a forward-only pass of the fused loss
            # kernel; its gradient is produced -- together with the KD term's, if the caller adds one -- by the engine
            # node's backward (lazy_logits.lazy_backward)
            state.labels = labels
            B, T = labels.shape
            lab = state.select_rows(labels.reshape(B * T, 1)).reshape(-1).contiguous()
            loss = _CEFn.apply(logits, state, lab)
        return Seq2SeqLMOutput(loss=loss, logits=LazyLogits.wrap(logits, state), encoder_last_hidden_state=enc,
                               _logits_lowp=lowp, _model=self, _rows=sel, _lazy=state)

    @torch.no_grad()
    def generate(self, input_features=None, generation_config=None, logits_processor=None, stopping_criteria=None,
                 prefix_allowed_tokens_fn=None, synced_gpus=False, return_timestamps=None, task=None, language=None,
                 is_multilingual=None, prompt_ids=None, prompt_condition_type=None, condition_on_prev_tokens=None,
                 temperature=None, compression_ratio_threshold=None, logprob_threshold=None, no_speech_threshold=None,
                 num_segment_frames=None, attention_mask=None, time_precision=0.02, time_precision_features=0.01,
                 return_token_timestamps=None, return_segments=False, return_dict_in_generate=None,
                 force_unique_generate_call=None, monitor_progress=None, *, use_graphs=None, **kwargs):
        """`WhisperGenerationMixin.generate` (TF:generation_whisper.py:383-968) on the MI355X engine, with the argument
        list of the reference (run_distillation.py:1524-1528, run_eval.py:690-739, 806-844): greedy search over one
        30 s window per row with language / task / timestamp / prompt_ids prefixes, the suppress / begin-suppress /
        min-new-tokens / timestamp logits rules, `assistant_model` (speculative decoding), `encoder_outputs`.
        The token loop is decoding.GreedyDecoder (KV cache; `use_graphs` replays the per-position launch sequence from
        HIP graphs; `use_cache=False` re-decodes the whole prefix every step and exists as a cross-check).
        `num_beams > 1` runs decoding.beam_search_decode (TF `_beam_search`).  `return_timestamps=True` (without
        `force_unique_generate_call`) and inputs longer than 30 s run the reference's timestamp seek loop
        (`seek_decode`, TF:784-903) with `condition_on_prev_tokens`, the fallback thresholds (`temperature` tuple,
        `compression_ratio_threshold`, `logprob_threshold`) and the `no_speech_threshold` skip.
        Single-window greedy decoding also takes GenerationMixin's `repetition_penalty`, `no_repeat_ngram_size` and plain
        sampling (a positive `temperature`, as in the reference's generate_with_fallback, with `top_k` / `top_p`):
        decoding.GreedyDecoder `soft` -- token for token the reference's on the same device and seed.
        Arguments this path does not implement RAISE (nothing is silently ignored): group beam search, those three options
        combined with beams / an assistant / the seek loop, the fallback heuristics outside the seek loop, token-level
        timestamps, custom logits processors.
        Returns what the reference returns: the generated tokens only (decoder prompt and EOS stripped, right-padded
        with pad_token_id), or with `return_dict_in_generate=True` / `force_unique_generate_call=True` the full
        sequences (prompt + generated, as GenerationMixin emits them)."""
        from . import generation as G
        self._sync_shadow()
        eng, d = self.engine, self.dims
        # ---- arguments the engine path does not implement: loud, never ignored
        for name, val in (("logits_processor", logits_processor), ("stopping_criteria", stopping_criteria),
                          ("prefix_allowed_tokens_fn", prefix_allowed_tokens_fn), ("monitor_progress", monitor_progress)):
            if val is not None and (not hasattr(val, "__len__") or len(val) > 0):
                raise NotImplementedError(f"generate({name}=...) is not implemented on the MI355X engine path")
        if return_token_timestamps:
            raise NotImplementedError("return_token_timestamps is not implemented on the MI355X path")
        temps = list(temperature) if isinstance(temperature, (list, tuple)) else [temperature]
        fallback_args = dict(temperatures=temps, compression_ratio_threshold=compression_ratio_threshold,
                             logprob_threshold=logprob_threshold, no_speech_threshold=no_speech_threshold,
                             condition_on_prev_tokens=bool(condition_on_prev_tokens))
        uses_fallback = bool(condition_on_prev_tokens) or compression_ratio_threshold is not None or \
            logprob_threshold is not None or no_speech_threshold is not None or len(temps) > 1
        # plain sampling: as in the reference's generate_with_fallback (TF:generation_whisper.py `do_sample = temperature is not
        # None and temperature > 0.0`), a positive `temperature` IS the switch -- `do_sample=True` alone decodes greedily
        sample_temp = float(temps[0]) if (not uses_fallback and temps[0] is not None and temps[0] > 0.0) else None
        engine_keys = ("encoder_outputs", "assistant_model", "decoder_input_ids", "use_cache")
        unknown = [k for k in kwargs if k not in G._CONFIG_KEYS and k not in engine_keys]
        if unknown:
            raise ValueError(f"The following `model_kwargs` are not used by the model: {unknown} (note: typos in the "
                             "generate arguments will also show up in this list)")
        gc = G.GenerationConfig.from_any(generation_config if generation_config is not None else self.generation_config)
        for k in G._CONFIG_KEYS:
            if k in kwargs and kwargs[k] is not None:
                setattr(gc, k, kwargs[k])
        num_beams = int(getattr(gc, "num_beams", 1) or 1)
        if num_beams > 1 and (getattr(gc, "num_beam_groups", 1) or 1) != 1:
            raise NotImplementedError("group beam search is not implemented on the MI355X path")
        if (getattr(gc, "num_return_sequences", 1) or 1) != 1:
            raise NotImplementedError("num_return_sequences > 1 is not implemented on the MI355X path")
        if num_beams == 1 and getattr(gc, "length_penalty", None) not in (None, 1.0):
            raise NotImplementedError("length_penalty without beam search is not implemented on the MI355X path")
        # history-dependent processors / sampling (GenerationMixin's repetition penalty, no-repeat n-gram, temperature / top-k /
        # top-p sampling): on the single-call KV-cache decoder (decoding.GreedyDecoder `soft`); elsewhere they raise below
        soft = None
        rp, nrn = getattr(gc, "repetition_penalty", None), int(getattr(gc, "no_repeat_ngram_size", 0) or 0)
        if sample_temp is not None or rp not in (None, 1.0) or nrn:
            # (an unset top_k is GenerationMixin's global default of 50 when sampling; run_eval.py:739 passes top_k=0 = off)
            tk = getattr(gc, "top_k", None)
            soft = dict(do_sample=sample_temp is not None, temperature=sample_temp, top_k=50 if tk is None else tk,
                        top_p=getattr(gc, "top_p", None), repetition_penalty=rp, no_repeat_ngram_size=nrn)
            if num_beams > 1 or kwargs.get("assistant_model") is not None or kwargs.get("use_cache", True) is False:
                raise NotImplementedError("sampling / repetition_penalty / no_repeat_ngram_size are implemented for greedy "
                                          "single-window decoding with the KV cache (not with beams, an assistant or use_cache=False)")
        if gc.decoder_start_token_id is None:
            gc.decoder_start_token_id = d.decoder_start_token_id
        # ---- encoder
        encoder_outputs = kwargs.get("encoder_outputs")
        if encoder_outputs is not None:
            if uses_fallback:
                raise NotImplementedError("temperature fallback / condition_on_prev_tokens need input_features (the seek "
                                          "loop encodes every window itself)")
            enc = encoder_outputs
            if not torch.is_tensor(enc):
                enc = enc.last_hidden_state if hasattr(enc, "last_hidden_state") else enc[0]
            B = enc.shape[0] if enc.dim() == 3 else enc.shape[0] // d.max_src
            if enc.numel() != B * d.max_src * d.d_model:
                raise ValueError(f"encoder_outputs must cover {d.max_src} positions of width {d.d_model} per row")
            enc = enc.reshape(-1, d.d_model).to(eng.lowp).contiguous()   # the dtype `encode` returns (GEMM operand)
            dev = enc.device
        else:
            if input_features is None:
                raise ValueError("input_features or encoder_outputs are required")
            frames = input_features.shape[-1]
            rt = return_timestamps if return_timestamps is not None else bool(getattr(gc, "return_timestamps", False))
            if frames > 2 * d.max_src and not rt:
                raise ValueError(
                    "You have passed more than 3000 mel input features (> 30 seconds) which automatically enables "
                    "long-form generation which requires the model to predict timestamp tokens. Please either pass "
                    "`return_timestamps=True` or make sure to pass no more than 3000 mel input features.")
            if prompt_condition_type is not None:
                if prompt_condition_type not in ("first-segment", "all-segments"):
                    raise ValueError("`prompt_condition_type` must be either 'first-segment' or 'all-segments'")
                gc.prompt_condition_type = prompt_condition_type
            if frames > 2 * d.max_src or (rt and not force_unique_generate_call):
                # the reference's seek loop (TF:784-903): with timestamps every window is decoded until its audio is
                # consumed, also when the input is a single 30 s window (run_pseudo_labelling.py:861-996 calls it so)
                if soft is not None:
                    raise NotImplementedError("sampling / repetition_penalty / no_repeat_ngram_size are implemented for "
                                              "single-window decoding, not inside the timestamp seek loop, on the MI355X path")
                return self._generate_seek_loop(input_features, attention_mask, gc, language, task, is_multilingual,
                                                prompt_ids, kwargs, use_graphs, return_dict_in_generate, num_beams,
                                                fallback_args, return_segments)
            if return_segments:
                raise NotImplementedError("return_segments comes with the timestamp seek loop (return_timestamps=True) "
                                          "on the MI355X path")
            if uses_fallback:
                raise NotImplementedError("temperature fallback / condition_on_prev_tokens are implemented for the "
                                          "timestamp seek loop (return_timestamps=True) only on the MI355X path")
            if frames != 2 * d.max_src:
                raise ValueError(f"Whisper expects the mel input features to be of length {2 * d.max_src}, but found "
                                 f"{frames}. Make sure to pad the input mel features to {2 * d.max_src}.")
            B = input_features.shape[0]
            dev = input_features.device
            enc, _ = eng.encode(input_features.to(torch.float32).contiguous(), save=False)
        # ---- decoder prompt (TF:1384-1608, 1853-1918)
        if return_timestamps is None:
            return_timestamps = bool(getattr(gc, "return_timestamps", False))
        if return_timestamps and not hasattr(gc, "no_timestamps_token_id"):
            raise ValueError("You are trying to return timestamps, but the generation config is not properly set. Make "
                             "sure to initialize the generation config with the correct attributes that are needed "
                             "such as `no_timestamps_token_id`.")
        gc.return_timestamps = bool(return_timestamps)
        G.set_language_and_task(gc, language, task, is_multilingual)
        if prompt_condition_type is not None and prompt_condition_type not in ("first-segment", "all-segments"):
            raise ValueError("`prompt_condition_type` must be either 'first-segment' or 'all-segments'")
        if "decoder_input_ids" in kwargs and kwargs["decoder_input_ids"] is not None:
            ids = kwargs["decoder_input_ids"].to(dev).long().clone()
        else:
            def detect():
                return self._detect_language(enc, B, gc)
            rows = G.retrieve_init_tokens(gc, B, detect)
            ids = torch.as_tensor(rows, dtype=torch.long, device=dev)
            if prompt_ids is not None:
                pr = torch.as_tensor(prompt_ids, dtype=torch.long, device=dev).reshape(-1)
                ids = torch.cat([pr[None, :].expand(B, -1), ids], 1)
        P = ids.shape[1]
        max_new, min_new = G.resolve_lengths(gc, P, d.max_tgt, "max_length" in kwargs)
        eos, pad = gc.eos_token_id, gc.pad_token_id
        if isinstance(eos, (list, tuple)):
            if len(eos) != 1:
                raise NotImplementedError("several eos_token_id values are not implemented on the MI355X path")
            eos = eos[0]
        if pad is None:
            pad = eos
        suppress = list(gc.suppress_tokens) if gc.suppress_tokens else None
        begin_suppress = list(gc.begin_suppress_tokens) if gc.begin_suppress_tokens else None
        assistant_model = kwargs.get("assistant_model")
        use_cache = kwargs.get("use_cache", True)
        if use_cache is None:
            use_cache = True
        if use_graphs is None:
            use_graphs = False
        if num_beams > 1:
            # beam search (run_eval.py:143, 693; run_distillation.py:1428-1436): TF `_beam_search` on the KV-cache decoder
            from .decoding import beam_search_decode
            if assistant_model is not None:
                raise ValueError("assistant_model cannot be combined with beam search (TF raises the same)")
            if eos is None:
                raise ValueError("beam search needs eos_token_id in the generation config")
            ts_rules = None
            if gc.return_timestamps:
                ts_rules = dict(begin_index=P, no_timestamps_token_id=int(gc.no_timestamps_token_id),
                                max_initial_timestamp_index=getattr(gc, "max_initial_timestamp_index", None))
            lp = getattr(gc, "length_penalty", None)
            es = getattr(gc, "early_stopping", False)
            seqs = beam_search_decode(eng, enc, ids, max_new, num_beams, eos, pad_token_id=pad, suppress_tokens=suppress,
                                      begin_suppress_tokens=begin_suppress, min_new_tokens=min_new,
                                      length_penalty=1.0 if lp is None else float(lp),
                                      early_stopping=False if es is None else es, timestamp_rules=ts_rules)
        elif assistant_model is not None:
            # speculative decoding (run_eval.py:578-599, 706-707): the assistant drafts, this model verifies.  An
            # assistant with this model's encoder dimensions re-uses the encoder output (the distilled student keeps
            # a frozen copy of the teacher's encoder); otherwise it encodes the features itself.
            from .decoding import assisted_greedy_decode
            a_ts_rules = None
            if gc.return_timestamps:       # (single window, force_unique_generate_call: the seek loop has its own branch)
                if eos is None:
                    raise ValueError("return_timestamps=True needs eos_token_id in the generation config")
                a_ts_rules = dict(begin_index=P, no_timestamps_token_id=int(gc.no_timestamps_token_id),
                                  max_initial_timestamp_index=getattr(gc, "max_initial_timestamp_index", None))
            begin_suppress = None          # TF:719-721: the model must be able to return EOS right away
            assistant_model._sync_shadow()
            ad = assistant_model.dims
            if getattr(assistant_model, "share_encoder_output", None) or \
                    (input_features is None and ad.d_model == d.d_model):
                enc_a = enc.to(assistant_model.engine.lowp)
            else:
                if input_features is None:
                    raise ValueError("the assistant needs input_features (its encoder differs from this model's)")
                enc_a, _ = assistant_model.engine.encode(input_features.to(torch.float32).contiguous(), save=False)
            k = getattr(gc, "num_assistant_tokens", None) or \
                getattr(getattr(assistant_model, "generation_config", None), "num_assistant_tokens", None) or 5
            seqs, self.last_drafted, self.last_accepted = assisted_greedy_decode(
                eng, assistant_model.engine, enc, enc_a, ids, max_new, int(k), eos, suppress_tokens=suppress,
                min_new_tokens=min_new, pad_token_id=pad, timestamp_rules=a_ts_rules)
        elif use_cache:
            from .decoding import GreedyDecoder
            ts_rules = None
            if gc.return_timestamps:
                if eos is None:
                    raise ValueError("return_timestamps=True needs eos_token_id in the generation config")
                ts_rules = dict(begin_index=P, no_timestamps_token_id=int(gc.no_timestamps_token_id),
                                max_initial_timestamp_index=getattr(gc, "max_initial_timestamp_index", None))
            total = P + max_new
            key = (B, total, eos, pad, bool(use_graphs), tuple(suppress or ()), tuple(begin_suppress or ()),
                   None if ts_rules is None else tuple(sorted(ts_rules.items())),
                   None if soft is None else tuple(sorted((k, v) for k, v in soft.items())))
            dec = self._decoders.get(key)
            if dec is None:
                dec = GreedyDecoder(eng, B, total, eos_token_id=eos, suppress_tokens=suppress,
                                    begin_suppress_tokens=begin_suppress, use_graphs=use_graphs,
                                    check_every=16 if use_graphs else 1, timestamp_rules=ts_rules, pad_token_id=pad, soft=soft)
                self._decoders = {key: dec}        # one live decoder (its graphs pin the K/V cache buffers)
            seqs = dec.run(enc, ids, max_new, min_new)
        else:
            if gc.return_timestamps:
                raise ValueError("return_timestamps=True runs on the KV-cache decoder (use_cache=True)")
            seqs = self._greedy_no_cache(enc, ids, max_new, min_new, eos, pad, suppress, begin_suppress)
        seqs = self._trim_finished(seqs, P, eos, pad)
        if return_dict_in_generate or getattr(gc, "return_dict_in_generate", False):
            return G.GenerateOutput(seqs)
        if force_unique_generate_call:
            return seqs
        return G.strip_and_pad(seqs, P, eos, pad)

    def seek_decode(self, input_features, max_frames, init_tokens, lengths, eos, pad, no_timestamps_token_id,
                    max_initial_timestamp_index=None, suppress_tokens=None, begin_suppress_tokens=None,
                    detect_language=None, temperatures=(0.0,), compression_ratio_threshold=None, logprob_threshold=None,
                    no_speech_threshold=None, condition_on_prev_tokens=False, prev_sot_token_id=None, prompt_ids=None,
                    prompt_all_segments=False, num_beams=1, length_penalty=1.0, early_stopping=False, assistant=None,
                    num_assistant_tokens=5):
        """The seek loop itself (TF:generation_whisper.py:784-903): input_features [B, n_mels, frames], max_frames[b] =
        valid mel frames of utterance b.  init_tokens: the decoder prompt rows (list of B lists) or a callable(detect)
        building them (detect() = language ids from the first window); lengths(P) -> (max_new_tokens, min_new_tokens)
        for a decoder prompt of P tokens.  Every pass encodes the next <= 30 s window of each unfinished utterance,
        decodes it with the timestamp rules and advances that utterance by what `retrieve_segment` says it consumed.
          * condition_on_prev_tokens (TF:1853-1918): the tokens of the utterance's earlier segments (last 223, behind
            <|startofprev|>) precede the prompt.  The reference left-pads a batch and masks the pads; rows are
            independent, so here rows are decoded in groups of equal prompt length (no pads, same positions);
          * fallback (TF:970-1116, 1243-1287): a window whose zlib compression ratio of the token bytes exceeds
            `compression_ratio_threshold` or whose average log-probability is below `logprob_threshold` is decoded
            again at the next temperature (sampling; the random stream is this process's, not the reference's);
            `no_speech_threshold`: P(<|nospeech|>) after <|startoftranscript|> above it together with a low average
            log-probability skips the window.  The scores are recomputed by one teacher-forced decoder pass.
          * prompt_ids (`processor.get_prompt_ids`, run_eval.py:709-710; prompt_condition_type "first-segment"): without
            conditioning on previous tokens the prompt precedes the decoder prompt of EVERY window (TF:1909-1911); with
            it the prompt is the utterance's segment zero (TF:1119-1123), so it conditions the following windows like
            any earlier text until the 223-token cut-off pushes it out, and is dropped from the result (TF:906-910);
            prompt_all_segments (prompt_condition_type "all-segments", needs condition_on_prev_tokens): the prompt takes
            the place of <|startofprev|> in front of the previous tokens of every window (TF:1887-1888);
          * num_beams > 1 (run_eval.py:693 / run_pseudo_labelling.py `--generation_num_beams`): the temperature-0 pass of a
            window is a beam search (decoding.beam_search_decode with the timestamp rules); sampled fallback passes use
            one beam like the reference (TF:1010-1012).  The score-based thresholds (logprob / no-speech) are greedy-only;
          * assistant = (engine, encode) of a draft model (run_eval.py:578-599, 706-707 with long-form inputs): the
            temperature-0 pass of a window is decoding.assisted_greedy_decode with the timestamp rules; encode(features)
            gives the assistant's encoder output of a window batch, or None when it shares this model's.
        -> per utterance the list of segments {"start", "end", "tokens"}."""
        import math
        import zlib
        from . import generation as G
        from .decoding import GreedyDecoder, apply_timestamp_rules
        eng, d = self.engine, self.dims
        B = input_features.shape[0]
        dev = input_features.device
        W, V = 2 * d.max_src, d.vocab
        feats = input_features.to(torch.float32)
        seek = [0] * B
        temps = [0.0 if x is None else float(x) for x in (temperatures if isinstance(temperatures, (list, tuple))
                                                         else [temperatures])]

        def window(rows):
            seg = torch.zeros((len(rows), feats.shape[1], W), dtype=torch.float32, device=dev)
            for i, b in enumerate(rows):
                n = min(max_frames[b] - seek[b], W)
                seg[i, :, :n] = feats[b, :, seek[b]:seek[b] + n]
            return seg
        if callable(init_tokens):
            init = init_tokens(lambda: detect_language(eng.encode(window(list(range(B))), save=False)[0]))
        else:
            init = init_tokens
        P0 = len(init[0])
        nts = int(no_timestamps_token_id)
        tb = nts + 1
        cut_off = d.max_tgt // 2 - 1
        prev_sot = prev_sot_token_id
        if prev_sot is None and suppress_tokens is not None and len(suppress_tokens) >= 2:
            prev_sot = suppress_tokens[-2]
        need_scores = logprob_threshold is not None or no_speech_threshold is not None
        if int(num_beams) > 1 and need_scores:
            raise NotImplementedError("logprob_threshold / no_speech_threshold with beam search are not implemented on the "
                                      "MI355X path (the reference scores beams by `sequences_scores`)")
        if no_speech_threshold is not None and logprob_threshold is None:
            raise ValueError("no_speech_threshold needs logprob_threshold (the reference compares both)")

        def vmask(ids):
            mk = torch.zeros(V, dtype=torch.bool, device=dev)
            if ids:
                mk[torch.as_tensor(list(ids), dtype=torch.long, device=dev)] = True
            return mk
        sup, bsup = vmask(suppress_tokens), vmask(begin_suppress_tokens)

        def processed(raw, hist, n, P, min_new):
            """The reference's processed scores of one step: raw f32 [r, V], hist int64 [r, >= n] (n tokens so far)."""
            sc = raw.clone()
            if n - P < min_new:
                sc[:, eos] = float("-inf")
            if n == P:
                sc = sc.masked_fill(bsup[None, :], float("-inf"))
            sc = sc.masked_fill(sup[None, :], float("-inf"))
            return apply_timestamp_rules(sc, hist, n, P, nts, eos, max_initial_timestamp_index)

        def sample(enc, ids, max_new, min_new, temp):
            """Multinomial sampling at temperature `temp` over the processed scores (fallback passes)."""
            r, P = ids.shape
            cache = eng.decode_init(enc, r, P + max_new)
            toks = torch.full((r, P + max_new), pad, dtype=torch.long, device=dev)
            toks[:, :P] = ids
            done = torch.zeros(r, dtype=torch.bool, device=dev)
            logits = eng.decode_multi(ids, cache).view(r, P, -1)[:, -1, :V]
            n = P
            while True:
                pr = torch.softmax(processed(logits.float(), toks, n, P, min_new) / temp, -1)
                nxt = torch.multinomial(pr, 1)[:, 0]
                nxt = torch.where(done, torch.full_like(nxt, pad), nxt)
                toks[:, n] = nxt
                done = done | (nxt == eos)
                n += 1
                if n >= P + max_new or bool(done.all()):
                    break
                logits = eng.decode_step(nxt[:, None].contiguous(), cache)[:, :V]
            return toks[:, :n]

        def scores_of(enc, ids, gens, min_new):
            """(average log-probability of the chosen tokens, P(<|nospeech|>) after <|startoftranscript|>) per row from
            ONE teacher-forced decoder pass over prompt + generated tokens (TF `_retrieve_avg_logprobs`, 1958-1975;
            `WhisperNoSpeechDetection`, TF:generation/logits_process.py)."""
            r, P = ids.shape
            L = max(1, max(len(g) for g in gens))
            full = torch.full((r, P + L), eos, dtype=torch.long, device=dev)
            full[:, :P] = ids
            for i, g in enumerate(gens):
                if g:
                    full[i, P:P + len(g)] = torch.as_tensor(g, dtype=torch.long, device=dev)
            T = full.shape[1]
            logits, _ = eng.decode(full[:, :T - 1].contiguous() if T > 1 else full, enc, save=False)
            logits = logits[:r * (T - 1)].view(r, T - 1, -1)[:, :, :V].float()
            tot = torch.zeros(r, dtype=torch.float64, device=dev)
            first = None
            for j in range(L):
                sc = processed(logits[:, P - 1 + j], full, P + j, P, min_new)
                if j == 0:
                    first = sc
                lp = torch.log_softmax(sc, -1).gather(1, full[:, P + j:P + j + 1])[:, 0]
                live = torch.as_tensor([j < len(g) for g in gens], device=dev)
                tot += torch.where(live, lp.double(), torch.zeros_like(tot))
            avg = [float(tot[i]) / len(g) if g else 0.0 for i, g in enumerate(gens)]
            if P0 > 1:
                nsp = torch.softmax(logits[:, P - P0], -1)[:, nts - 1]
            else:
                nsp = torch.softmax(first, -1)[:, nts - 1]
            return avg, nsp.tolist()

        def ratio(tokens):
            nbytes = int(math.log2(V) / 8) + 1
            raw = b"".join(int(x).to_bytes(nbytes, "little") for x in tokens)
            return len(raw) / len(zlib.compress(raw))

        segments = [[] for _ in range(B)]
        prompt = [int(x) for x in prompt_ids] if prompt_ids is not None else None
        first_segment_prompt = bool(prompt) and not prompt_all_segments
        if first_segment_prompt:
            body = prompt[1:] if (prev_sot is not None and prompt[0] == prev_sot) else prompt
            segments = [[{"tokens": list(body)}] for _ in range(B)]
        do_cond = [bool(condition_on_prev_tokens)] * B
        if not hasattr(self, "_seek_decoders"):
            self._seek_decoders = {}           # reused by later calls (pseudo-labelling decodes batch after batch)
        decoders = self._seek_decoders
        while any(seek[b] < max_frames[b] for b in range(B)):
            rows = [b for b in range(B) if seek[b] < max_frames[b]]
            snf = {b: min(max_frames[b] - seek[b], W) for b in rows}
            enc_all, _ = eng.encode(window(rows), save=False)
            enc_all = enc_all[:len(rows) * d.max_src].view(len(rows), d.max_src, -1)
            # decoder prompts (TF:1853-1918)
            prompts = {}
            cond_now = any(do_cond[b] for b in rows) and len(segments[0]) > 0
            for b in rows:
                pre = []
                if cond_now:
                    if prev_sot is None:
                        raise ValueError("condition_on_prev_tokens needs prev_sot_token_id in the generation config")
                    if do_cond[b] and segments[b]:
                        for sg in segments[b]:
                            tk = sg["tokens"]
                            pre += tk[:-1] if (len(tk) > 2 and tk[-2] >= tb) else tk
                        pre = pre[-cut_off:]
                    pre = (list(prompt) if (prompt and prompt_all_segments) else [prev_sot]) + pre
                elif prompt:
                    pre = list(prompt)
                prompts[b] = pre + list(init[b])
            accepted = {}
            # (the reference left-pads the batch to its longest prompt and derives the lengths from that, TF:835-840)
            max_new, min_new = lengths(max(len(prompts[b]) for b in rows))
            for P in sorted({len(prompts[b]) for b in rows}):
                group = [b for b in rows if len(prompts[b]) == P]
                pending = list(group)
                for ti, temp in enumerate(temps):
                    pos = [rows.index(b) for b in pending]
                    enc = enc_all[pos].reshape(len(pending) * d.max_src, -1).contiguous()
                    ids = torch.as_tensor([prompts[b] for b in pending], dtype=torch.long, device=dev)
                    if temp > 0.0:
                        out = sample(enc, ids, max_new, min_new, temp)[:, P:].tolist()
                    elif assistant is not None:
                        from .decoding import assisted_greedy_decode
                        a_eng, a_encode = assistant
                        enc_a = enc.to(a_eng.lowp) if a_encode is None else a_encode(window(pending))
                        out, nd, na = assisted_greedy_decode(
                            eng, a_eng, enc, enc_a, ids, max_new, int(num_assistant_tokens), eos,
                            suppress_tokens=suppress_tokens, min_new_tokens=min_new, pad_token_id=pad,
                            timestamp_rules=dict(begin_index=P, no_timestamps_token_id=nts,
                                                 max_initial_timestamp_index=max_initial_timestamp_index))
                        self.last_drafted = getattr(self, "last_drafted", 0) + nd
                        self.last_accepted = getattr(self, "last_accepted", 0) + na
                        out = torch.cat([out, torch.full((out.shape[0], P + max_new - out.shape[1]), pad,
                                                         dtype=out.dtype, device=dev)], 1)[:, P:].tolist()
                    elif int(num_beams) > 1:
                        from .decoding import beam_search_decode
                        out = beam_search_decode(
                            eng, enc, ids, max_new, int(num_beams), eos, pad_token_id=pad, suppress_tokens=suppress_tokens,
                            begin_suppress_tokens=begin_suppress_tokens, min_new_tokens=min_new,
                            length_penalty=float(length_penalty), early_stopping=early_stopping,
                            timestamp_rules=dict(begin_index=P, no_timestamps_token_id=nts,
                                                 max_initial_timestamp_index=max_initial_timestamp_index))[:, P:].tolist()
                    else:
                        key = (len(pending), P, max_new, eos, pad, nts, max_initial_timestamp_index,
                               tuple(suppress_tokens or ()), tuple(begin_suppress_tokens or ()))
                        dec = decoders.get(key)
                        if dec is None:
                            if len(decoders) >= 4:     # (a decoder owns its K/V caches: keep a handful alive)
                                decoders.pop(next(iter(decoders)))
                            dec = decoders[key] = GreedyDecoder(
                                eng, len(pending), P + max_new, eos_token_id=eos, suppress_tokens=suppress_tokens,
                                begin_suppress_tokens=begin_suppress_tokens, use_graphs=False, pad_token_id=pad,
                                check_every=4,      # eager passes: stop within 3 steps of the last row's EOS
                                timestamp_rules=dict(begin_index=P, no_timestamps_token_id=nts,
                                                     max_initial_timestamp_index=max_initial_timestamp_index))
                        out = dec.run(enc, ids, max_new, min_new)[:, P:].tolist()
                    gens = []
                    for seq in out:
                        if seq and seq[-1] == pad:     # TF:1064-1071: drop the padding (all but one EOS when pad == EOS)
                            npad = sum(1 for x in seq if x == pad) - (1 if pad == eos else 0)
                            if npad:
                                seq = seq[:-npad]
                        gens.append(seq)
                    avg = nsp = None
                    if need_scores:
                        avg, nsp = scores_of(enc, ids, gens, min_new)
                    again = []
                    for i, b in enumerate(pending):
                        fallback = skip = False
                        if compression_ratio_threshold is not None and gens[i] and \
                                ratio(gens[i]) > compression_ratio_threshold:
                            fallback = True
                        if logprob_threshold is not None and avg[i] < logprob_threshold:
                            fallback = True
                        if no_speech_threshold is not None and avg[i] < logprob_threshold and nsp[i] > no_speech_threshold:
                            fallback, skip = False, True
                        seq = gens[i][:-1] if (gens[i] and gens[i][-1] == eos) else gens[i]
                        accepted[b] = (seq, skip)
                        do_cond[b] = bool(condition_on_prev_tokens) and temp < 0.5
                        if fallback:
                            again.append(b)
                    if not again or ti == len(temps) - 1:
                        break
                    pending = again
            for b in rows:
                seq, skip = accepted[b]
                if skip:
                    seek[b] += snf[b]
                    continue
                segs, offset = G.retrieve_segment(seq, tb, snf[b], time_offset=seek[b] * 0.01)
                seek[b] += offset
                segments[b] += segs
        return [sg[1:] for sg in segments] if first_segment_prompt else segments

    def _generate_seek_loop(self, input_features, attention_mask, gc, language, task, is_multilingual, prompt_ids, kwargs,
                            use_graphs, return_dict_in_generate, num_beams, fallback_args=None, return_segments=False):
        """Timestamp-driven multi-pass transcription: `WhisperGenerationMixin.generate` steps 5-7 (TF:745-968) with
        temperature 0 and no fallback thresholds -- every utterance keeps a `seek` position in mel frames; each pass
        decodes the next <= 30 s window of every unfinished utterance with the timestamp rules, `retrieve_segment`
        splits the tokens at consecutive timestamp pairs and advances `seek` to the last predicted end of segment (or
        past the window).  Inputs of any length ([B, n_mels, frames]; batches of long inputs need `attention_mask`).
        Returns the concatenated segment tokens per utterance, right-padded with pad_token_id (the reference's plain
        return value), a GenerateOutput with `.sequences` and `.segments` (return_dict_in_generate), or -- with
        `return_segments=True`, like the reference (TF:generate, "8. If we return all segments") -- a plain dict
        {"sequences", "segments"}: per utterance the list of {"start", "end", "tokens"} in seconds / token ids (the
        reference's entries also carry the raw GenerationMixin output of their window under "result" and "idxs")."""
        from . import generation as G
        eng, d = self.engine, self.dims
        B, _, frames = input_features.shape
        dev = input_features.device
        W = 2 * d.max_src
        if kwargs.get("decoder_input_ids") is not None:
            raise NotImplementedError("decoder_input_ids with the timestamp seek loop are not implemented on the MI355X "
                                      "path (pass prompt_ids, or force_unique_generate_call=True for a single window)")
        all_segments = getattr(gc, "prompt_condition_type", "first-segment") == "all-segments"
        if all_segments and not (fallback_args or {}).get("condition_on_prev_tokens"):
            raise ValueError("Make sure to set `condition_on_prev_tokens=True` when setting "
                             "`prompt_condition_type='all-segments'`.")
        if kwargs.get("use_cache", True) is False:
            raise NotImplementedError("the timestamp seek loop runs on the KV-cache decoder only")
        assistant, begin_suppress_loop = None, (list(gc.begin_suppress_tokens) if gc.begin_suppress_tokens else None)
        am = kwargs.get("assistant_model")
        if am is not None:
            if num_beams != 1:
                raise ValueError("assistant_model cannot be combined with beam search (TF raises the same)")
            am._sync_shadow()
            shared = bool(getattr(am, "share_encoder_output", None))     # (a distilled student keeps the teacher's encoder)
            a_encode = None if shared else \
                (lambda f: am.engine.encode(f.to(torch.float32).contiguous(), save=False)[0])
            assistant = (am.engine, a_encode)
            begin_suppress_loop = None     # TF:719-721: with an assistant the model must be able to return EOS right away
            self.last_drafted = self.last_accepted = 0
        if not hasattr(gc, "no_timestamps_token_id"):
            raise ValueError("You are trying to return timestamps, but the generation config is not properly set. Make "
                             "sure to initialize the generation config with the correct attributes that are needed "
                             "such as `no_timestamps_token_id`.")
        gc.return_timestamps = True
        G.set_language_and_task(gc, language, task, is_multilingual)
        if B > 1 and frames > W and attention_mask is None:
            raise ValueError("When doing batched long-form audio transcription, make sure to pass an `attention_mask`. "
                             "You can retrieve the `attention_mask` by doing `processor(audio, ..., "
                             "return_attention_mask=True)` ")
        if B > 1 and frames > W:
            max_frames = [int(x) for x in attention_mask.sum(-1).tolist()]
        else:
            max_frames = [frames] * B
        def detect_on(enc0):
            return self._detect_language(enc0, B, gc)
        eos, pad = gc.eos_token_id, gc.pad_token_id
        if isinstance(eos, (list, tuple)):
            eos = eos[0]
        if eos is None:
            raise ValueError("return_timestamps=True needs eos_token_id in the generation config")
        if pad is None:
            pad = eos
        explicit_max_length = "max_length" in kwargs

        def lengths(P):
            max_new, min_new = G.resolve_lengths(gc, P, d.max_tgt, explicit_max_length)
            if gc.max_new_tokens is None:          # TF:1932-1940 mutates max_length on every pass
                gc.max_length = min(gc.max_length + min(d.max_tgt // 2 - 1, P), d.max_tgt)
            return max_new, min_new
        segments = self.seek_decode(input_features, max_frames, lambda det: G.retrieve_init_tokens(gc, B, det), lengths,
                                    eos, pad, int(gc.no_timestamps_token_id),
                                    getattr(gc, "max_initial_timestamp_index", None),
                                    list(gc.suppress_tokens) if gc.suppress_tokens else None,
                                    begin_suppress_loop,
                                    detect_language=detect_on, prev_sot_token_id=getattr(gc, "prev_sot_token_id", None),
                                    prompt_ids=(prompt_ids.tolist() if torch.is_tensor(prompt_ids) else prompt_ids),
                                    prompt_all_segments=all_segments and prompt_ids is not None,
                                    num_beams=num_beams, assistant=assistant,
                                    num_assistant_tokens=(getattr(gc, "num_assistant_tokens", None) or getattr(
                                        getattr(am, "generation_config", None), "num_assistant_tokens", None) or 5),
                                    length_penalty=1.0 if getattr(gc, "length_penalty", None) is None else gc.length_penalty,
                                    early_stopping=getattr(gc, "early_stopping", False) or False,
                                    **(fallback_args or {}))
        rows_out = [[tok for sg in segments[b] for tok in sg["tokens"]] for b in range(B)]
        width = max((len(r) for r in rows_out), default=0)
        seqs = torch.full((B, width), pad, dtype=torch.long, device=dev)
        for b, r in enumerate(rows_out):
            if r:
                seqs[b, :len(r)] = torch.as_tensor(r, dtype=torch.long, device=dev)
        if return_segments:
            return {"sequences": seqs, "segments": segments}
        if return_dict_in_generate or getattr(gc, "return_dict_in_generate", False):
            out = G.GenerateOutput(seqs)
            out.segments = segments
            return out
        return seqs

    # -- helpers of generate ----------------------------------------------------------------------------------------
    @staticmethod
    def _trim_finished(seqs, P, eos, pad):
        """GenerationMixin stops as soon as every row has emitted EOS: drop the all-padding columns a decoder that
        checks the stop condition every few steps (HIP-graph replay) may have appended."""
        if eos is None or seqs.shape[1] <= P:
            return seqs
        gen = seqs[:, P:]
        is_eos = gen == eos
        n = gen.shape[1]
        first = torch.where(is_eos.any(1), is_eos.float().argmax(1) + 1, torch.full((gen.shape[0],), n, device=gen.device))
        keep = int(first.max().item())
        return seqs[:, : P + keep]

    def _detect_language(self, enc, B, gc):
        """TF:1610-1674 `detect_language`: one decoder step on <|startoftranscript|>, argmax over the language ids."""
        d = self.dims
        ids = torch.full((B, 1), gc.decoder_start_token_id, dtype=torch.long, device=enc.device)
        logits, _ = self.engine.decode(ids, enc, save=False)
        sc = logits[:B, : d.vocab].float()
        lang_ids = torch.as_tensor(sorted(gc.lang_to_id.values()), dtype=torch.long, device=enc.device)
        mask = torch.full((d.vocab,), float("-inf"), device=enc.device)
        mask[lang_ids] = 0.0
        return (sc + mask).argmax(-1).tolist()

    def _greedy_no_cache(self, enc, ids, max_new, min_new, eos, pad, suppress, begin_suppress):
        """Greedy search that re-decodes the whole prefix at every step (no KV cache): the cross-check of the cached
        decoder."""
        eng, d = self.engine, self.dims
        B, dev = ids.shape[0], ids.device
        done = torch.zeros(B, dtype=torch.bool, device=dev)

        def mask(tokens):
            if not tokens:
                return None
            m = torch.zeros(d.vocab, device=dev)
            m[torch.as_tensor(list(tokens), device=dev)] = float("-inf")
            return m
        sup, bsup = mask(suppress), mask(begin_suppress)
        for step in range(max_new):
            T = ids.shape[1]
            logits, _ = eng.decode(ids.contiguous(), enc, save=False)
            sc = logits[: B * T, : d.vocab].view(B, T, -1)[:, -1].float()
            if eos is not None and step < min_new:
                sc[:, eos] = float("-inf")
            if step == 0 and bsup is not None:
                sc = sc + bsup
            if sup is not None:
                sc = sc + sup
            nxt = sc.argmax(-1)
            if eos is not None:
                nxt = torch.where(done, torch.full_like(nxt, pad), nxt)
                done |= nxt == eos
            ids = torch.cat([ids, nxt[:, None]], 1)
            if eos is not None and bool(done.all()):
                break
        return ids
