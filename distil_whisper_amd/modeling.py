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
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .engine import ParamStore, WhisperDims, WhisperEngine
from .student_init import mel_filter_bank, random_state_dict


def _default_ops(device):
    from .ops_hip import HipOps  # raises when the HIP library or the GPU is missing: there is no CPU fallback
    return HipOps(device)


@dataclass
class Seq2SeqLMOutput:
    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    encoder_last_hidden_state: Optional[torch.Tensor] = None


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


class _Decoder(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.embed_tokens = nn.Embedding(d.vocab, d.d_model, device="meta")
        self.embed_positions = nn.Embedding(d.max_tgt, d.d_model, device="meta")
        self.layers = nn.ModuleList([_DecoderLayer(d.d_model, d.ffn) for _ in range(d.dec_layers)])
        self.layer_norm = nn.LayerNorm(d.d_model, device="meta")


class _Model(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.encoder = _Encoder(d)
        self.decoder = _Decoder(d)


class _EngineFn(torch.autograd.Function):
    """forward: engine encode+decode with activations kept; backward: engine backward from d(loss)/d(logits) (and
    optionally d/d(encoder_last_hidden_state)); parameter gradients are returned as views of the flat buffer."""

    @staticmethod
    def forward(ctx, model, input_features, enc_in, decoder_input_ids, *params):
        eng = model.engine
        train = any(ctx.needs_input_grad[4:])  # (grad mode is off inside Function.forward; this is the real signal)
        ectx = None
        if enc_in is None:
            enc, ectx = eng.encode(input_features.to(torch.float32).contiguous(), save=train and model._enc_trainable)
        else:
            rows = enc_in.shape[0] * enc_in.shape[1]
            enc = eng.act(rows, eng.dims.d_model)
            enc[:rows].copy_(enc_in.reshape(rows, -1))
        logits, dctx = eng.decode(decoder_input_ids.contiguous(), enc, save=train)
        B, T = decoder_input_ids.shape
        V = eng.dims.vocab
        ctx.model, ctx.ectx, ctx.dctx, ctx.shape = model, ectx, dctx, (B, T, V)
        ctx.logits_buf = logits if train else None
        Re = B * eng.dims.max_src
        out_logits = logits[: B * T, :V].float().view(B, T, V)       # accelerate upcasts model outputs to fp32
        enc_out = enc[:Re].float().view(B, eng.dims.max_src, -1)
        ctx.mark_non_differentiable(enc_out)
        return out_logits, enc_out

    @staticmethod
    def backward(ctx, g_logits, _g_enc):
        model, eng = ctx.model, ctx.model.engine
        B, T, V = ctx.shape
        st = eng.st
        buf = ctx.logits_buf
        buf.zero_()
        buf[: B * T, :V].copy_(g_logits.reshape(B * T, V))
        eng.zero_small_grads()
        denc = eng.backward_decoder(ctx.dctx, buf, want_denc=ctx.ectx is not None)
        if ctx.ectx is not None:
            eng.backward_encoder(ctx.ectx, denc)
        grads = []
        for name, p in zip(model._param_names, model._param_list):
            grads.append(st.g[name].clone() if (p.requires_grad and name in st.g) else None)
        return (None, None, None, None, *grads)


class WhisperForConditionalGeneration(nn.Module):
    def __init__(self, config, ops=None, device="cuda:0", state_dict=None, seed=0, frozen_prefixes=()):
        super().__init__()
        self.config = config
        self.dims = WhisperDims.from_any(config)
        self.ops = ops if ops is not None else _default_ops(device)
        sd = state_dict if state_dict is not None else random_state_dict(self.dims, seed, device=self.ops.device)
        self.store = ParamStore(self.ops, self.dims, sd, trainable=True, frozen_prefixes=tuple(frozen_prefixes))
        self.engine = WhisperEngine(self.ops, self.store, torch.float32)
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
        self._enc_trainable = self.store.is_trainable("model.encoder.conv1.weight")
        self._versions = None

    # -- reference surface ------------------------------------------------------------------------------------------
    def get_encoder(self):
        return self.model.encoder

    def get_decoder(self):
        return self.model.decoder

    def freeze_encoder(self):
        """Mirror of `student_model.freeze_encoder()` (run_distillation.py:1018-1021): the flat layout has to be
        rebuilt with the encoder in the frozen range."""
        raise RuntimeError("construct the model with frozen_prefixes=('model.encoder.',) instead: the parameter layout "
                           "of the MI355X engine is fixed at construction")

    def _sync_shadow(self):
        v = tuple(p._version for p in self._param_list)
        if v != self._versions:
            self.store.refresh_shadow()  # fp32 master (possibly just updated by an optimizer) -> bf16 GEMM operands
            self._versions = v

    def forward(self, input_features=None, attention_mask=None, decoder_input_ids=None, labels=None,
                encoder_outputs=None, **kwargs):
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
        logits, enc = _EngineFn.apply(self, input_features, enc_in, decoder_input_ids, *self._param_list)
        loss = None
        if labels is not None:
            loss = F.cross_entropy(logits.view(-1, d.vocab), labels.reshape(-1))
        return Seq2SeqLMOutput(loss=loss, logits=logits, encoder_last_hidden_state=enc)

    @torch.no_grad()
    def generate(self, input_features=None, max_new_tokens=32, decoder_start_ids=None, eos_token_id=None,
                 use_cache=True, use_graphs=False, suppress_tokens=None, begin_suppress_tokens=None,
                 encoder_outputs=None, assistant_model=None, num_assistant_tokens=5, return_timestamps=False,
                 no_timestamps_token_id=None, max_initial_timestamp_index=None, **kwargs):
        """Greedy decoding (run_distillation.py:1524-1528 `generate_step`, run_eval.py:739) on the engine.  With
        use_cache the decoder runs one token per step against a KV cache (static cross-attention K/V computed once,
        self-attention K/V appended in place; decoding.GreedyDecoder), optionally with the per-position launch
        sequence replayed from HIP graphs (use_graphs); use_cache=False re-decodes the whole prefix every step and
        exists as the cross-check.  `encoder_outputs` (engine layout or [B, 1500, D]) skips the encoder, as
        run_eval.py's benchmark_gen does (806-844).  Beam search, timestamp rules and the temperature-fallback logic
        of TF:generation_whisper.py are outside this round's scope."""
        self._sync_shadow()
        eng, d = self.engine, self.dims
        if encoder_outputs is not None:
            enc = encoder_outputs.last_hidden_state if hasattr(encoder_outputs, "last_hidden_state") else encoder_outputs
            B = enc.shape[0] if enc.dim() == 3 else enc.shape[0] // d.max_src
            enc = enc.reshape(-1, d.d_model).to(eng.stream).contiguous()
            dev = enc.device
        else:
            B = input_features.shape[0]
            dev = input_features.device
            enc, _ = eng.encode(input_features.to(torch.float32).contiguous(), save=False)
        ids = torch.full((B, 1), d.decoder_start_token_id, dtype=torch.long, device=dev) \
            if decoder_start_ids is None else decoder_start_ids.clone()
        total = ids.shape[1] + max_new_tokens
        if total > d.max_tgt:
            raise ValueError(f"prompt + max_new_tokens = {total} exceeds max_target_positions = {d.max_tgt}")
        if assistant_model is not None:
            # speculative decoding (run_eval.py:578-599, 706-707): the assistant drafts, this model verifies; an
            # assistant with this model's encoder dimensions re-uses the encoder output (the distilled student
            # keeps a frozen copy of the teacher's encoder), otherwise it encodes the features itself
            from .decoding import assisted_greedy_decode
            if suppress_tokens or begin_suppress_tokens:
                raise ValueError("assisted decoding does not take suppress_tokens / begin_suppress_tokens here")
            assistant_model._sync_shadow()
            ad = assistant_model.dims
            if getattr(assistant_model, "share_encoder_output", None) or \
                    (input_features is None and ad.d_model == d.d_model):
                enc_a = enc.to(assistant_model.engine.stream)
            else:
                if input_features is None:
                    raise ValueError("the assistant needs input_features (its encoder differs from this model's)")
                enc_a, _ = assistant_model.engine.encode(input_features.to(torch.float32).contiguous(), save=False)
            out, self.last_drafted, self.last_accepted = assisted_greedy_decode(
                eng, assistant_model.engine, enc, enc_a, ids, max_new_tokens, num_assistant_tokens, eos_token_id)
            return out
        ts_rules = None
        if return_timestamps:
            # WhisperTimeStampLogitsProcessor (TF:generation_whisper.py:1774-1812); needs the vocabulary landmarks
            if no_timestamps_token_id is None or eos_token_id is None:
                raise ValueError("return_timestamps=True needs no_timestamps_token_id and eos_token_id")
            if not use_cache:
                raise ValueError("return_timestamps=True runs on the KV-cache decoder (use_cache=True)")
            ts_rules = dict(begin_index=ids.shape[1], no_timestamps_token_id=int(no_timestamps_token_id),
                            max_initial_timestamp_index=max_initial_timestamp_index)
            use_graphs = False            # the rule kernels have not been exercised under stream capture yet
        if use_cache:
            from .decoding import GreedyDecoder
            key = (B, total, eos_token_id, bool(use_graphs), tuple(suppress_tokens or ()),
                   tuple(begin_suppress_tokens or ()), None if ts_rules is None else tuple(sorted(ts_rules.items())))
            dec = self._decoders.get(key)
            if dec is None:
                dec = GreedyDecoder(eng, B, total, eos_token_id=eos_token_id, suppress_tokens=suppress_tokens,
                                    begin_suppress_tokens=begin_suppress_tokens, use_graphs=use_graphs,
                                    check_every=16 if use_graphs else 1, timestamp_rules=ts_rules)
                self._decoders = {key: dec}        # one live decoder (its graphs pin the K/V cache buffers)
            return dec.run(enc, ids, max_new_tokens)
        done = torch.zeros(B, dtype=torch.bool, device=ids.device)
        sup = None
        if suppress_tokens:
            sup = torch.zeros(d.vocab, device=dev)
            sup[torch.as_tensor(list(suppress_tokens), device=dev)] = float("-inf")
        bsup = None
        if begin_suppress_tokens:
            bsup = torch.zeros(d.vocab, device=dev)
            bsup[torch.as_tensor(list(begin_suppress_tokens), device=dev)] = float("-inf")
        for step in range(max_new_tokens):
            T = ids.shape[1]
            logits, _ = eng.decode(ids.contiguous(), enc, save=False)
            sc = logits[: B * T, : d.vocab].view(B, T, -1)[:, -1].float()
            if sup is not None:
                sc = sc + sup
            if step == 0 and bsup is not None:
                sc = sc + bsup
            nxt = sc.argmax(-1)
            if eos_token_id is not None:
                nxt = torch.where(done, torch.full_like(nxt, eos_token_id), nxt)
                done |= nxt == eos_token_id
            ids = torch.cat([ids, nxt[:, None]], 1)
            if eos_token_id is not None and bool(done.all()):
                break
        return ids
