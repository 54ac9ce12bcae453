"""The HIP flash-attention kernel as a `transformers` attention implementation (plug point 3).

The reference reaches attention through `ALL_ATTENTION_FUNCTIONS.get_interface(config._attn_implementation, ...)`
(TF:models/whisper/modeling_whisper.py:337-351; registry TF:modeling_utils.py:5093-5131; the CLI whitelists the
implementation names at run_distillation.py:141-148).  `register()` adds the MI355X kernel under the name "hip_attention":

    import distil_whisper_amd.attention_interface as hip_attn
    hip_attn.register()
    model = WhisperForConditionalGeneration.from_pretrained(..., attn_implementation="hip_attention")

after which an UNMODIFIED `transformers` Whisper model runs every attention core (encoder self 1500x1500, decoder self
causal, cross 447x1500, and the 1-query steps of cached decoding) in csrc/attention.hip, forward and backward
(`torch.autograd.Function` over dw_attn_fwd / dw_attn_bwd).  Contract of the callable (TF:modeling_whisper.py:215-238):
    fn(module, query [B,H,Lq,64], key [B,H,Lk,64], value, attention_mask or None, dropout=, scaling=, **kw)
        -> (attn_output [B,Lq,H,64] contiguous, None)
For a custom implementation name `transformers` builds no mask (masking_utils.py:813-820), so causality comes from
`module.is_causal` exactly as in `sdpa_attention_forward`; an explicit mask, dropout > 0 or a head size other than 64
raise -- nothing falls back to another implementation.
"""
import torch

NAME = "hip_attention"
_OPS = {}


def _ops(device):
    key = str(device)
    if key not in _OPS:
        from .ops_hip import HipOps          # raises without the HIP library / GPU: no fallback
        _OPS[key] = HipOps(device)
    return _OPS[key]


class _HipAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, causal, scale, ops):
        B, H, Lq, hd = q.shape
        Lk = k.shape[2]
        # [B,H,L,64] -> the kernel's token-major layout [B*L, H*64] (heads are column slices)
        qr = q.transpose(1, 2).reshape(B * Lq, H * hd).to(ops.lowp).contiguous()
        kr = k.transpose(1, 2).reshape(B * Lk, H * hd).to(ops.lowp).contiguous()
        vr = v.transpose(1, 2).reshape(B * Lk, H * hd).to(ops.lowp).contiguous()
        o, lse = ops.attn_fwd(qr, kr, vr, B, H, Lq, Lk, causal, scale)
        ctx.save_for_backward(qr, kr, vr, o, lse)
        ctx.meta = (B, H, Lq, Lk, causal, scale, ops, q.dtype, k.dtype, v.dtype)
        return o.view(B, Lq, H, hd).to(q.dtype)

    @staticmethod
    def backward(ctx, g):
        qr, kr, vr, o, lse = ctx.saved_tensors
        B, H, Lq, Lk, causal, scale, ops, dq_t, dk_t, dv_t = ctx.meta
        do = g.reshape(B * Lq, H * 64).to(ops.lowp).contiguous()
        dq, dk, dv = ops.attn_bwd(qr, kr, vr, o, do, lse, B, H, Lq, Lk, causal, scale)
        dq = dq.view(B, Lq, H, 64).transpose(1, 2).to(dq_t)
        dk = dk.view(B, Lk, H, 64).transpose(1, 2).to(dk_t)
        dv = dv.view(B, Lk, H, 64).transpose(1, 2).to(dv_t)
        return dq, dk, dv, None, None, None


def hip_attention_forward(module, query, key, value, attention_mask=None, dropout=0.0, scaling=None,
                                is_causal=None, **kwargs):
    if attention_mask is not None:
        raise NotImplementedError("hip_attention: explicit attention masks are not implemented (the Whisper training and "
                                  "greedy-decoding paths never pass one, SURVEY.md section 8a')")
    if dropout:
        raise NotImplementedError("hip_attention: attention dropout is not implemented (0.0 in every Whisper config)")
    if query.shape[-1] != 64 or key.shape[-1] != 64 or value.shape[-1] != 64:
        raise NotImplementedError(f"hip_attention: head_dim {query.shape[-1]} (the kernel implements Whisper's 64)")
    if key.shape[1] != query.shape[1]:
        raise NotImplementedError("hip_attention: grouped-query attention is not implemented (Whisper has none)")
    Lq, Lk = query.shape[2], key.shape[2]
    causal = is_causal if is_causal is not None else getattr(module, "is_causal", True)
    causal = bool(Lq > 1 and causal)          # sdpa_attention_forward's rule (TF:integrations/sdpa_attention.py)
    if causal and Lk != Lq:
        raise NotImplementedError("hip_attention: causal attention of several new queries against a longer KV cache")
    scale = float(scaling) if scaling is not None else 64 ** -0.5
    out = _HipAttention.apply(query, key, value, causal, scale, _ops(query.device))
    return out, None


def register(name: str = NAME):
    """Add the kernel to `transformers`' attention registry (idempotent).  Returns the registered name."""
    from transformers import AttentionInterface
    AttentionInterface.register(name, hip_attention_forward)
    return name
