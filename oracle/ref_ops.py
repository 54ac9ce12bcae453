"""TEST INFRASTRUCTURE -- not part of the product path.

Plain-torch restatement of every entry point of libdwamd.so (include/dwamd.h), with the same Python interface as
distil_whisper_amd.ops_hip.HipOps.  Two uses, both in tests only:
  * `-m gpu` tests run each HIP kernel and this restatement on the same device tensors and compare;
  * `-m "not gpu"` tests inject it into the host engine to check the hand-written forward/backward orchestration on
    CPU against the `transformers` reference (tests/test_engine_cpu.py).
The product (engine/trainer/bench) never imports this module; HipOps raises if the HIP library is missing.

Each function cites the reference arithmetic it restates (TF: = transformers 5.15 models/whisper).
"""
import math

import torch
import torch.nn.functional as F


class RefOps:
    name = "ref"

    def __init__(self, device="cpu", lowp=torch.bfloat16):
        """lowp = torch.float32 turns every bf16 rounding into the identity: used to check the engine's hand-written
        backward exactly against autograd."""
        self.device = torch.device(device)
        self.lowp = lowp

    def _bf(self, x):
        return x.to(self.lowp)

    def empty(self, shape, dtype):
        # Uninitialised memory is NaN here (floating types), so that the host-logic tests catch any path on which stale
        # rows of a buffer can reach a result (the HIP allocator hands back whatever the block held before)
        t = torch.empty(shape, dtype=dtype, device=self.device)
        if t.is_floating_point():
            t.fill_(float("nan"))
        return t

    def zeros(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype, device=self.device)

    # TF:feature_extraction_whisper.py:135-168 (_torch_extract_fbank_features), restated without torch.stft:
    # explicit reflect padding, framing, periodic Hann, 400-point real DFT by matmul in float64 then cast.
    def logmel(self, audio, filters):
        B, N = audio.shape
        x = audio.to(torch.float64)
        xp = F.pad(x.unsqueeze(1), (200, 200), mode="reflect").squeeze(1)
        frames = xp.unfold(1, 400, 160)[:, : N // 160]  # drop the last frame, as log_spec[..., :-1]
        win = torch.hann_window(400, periodic=True, dtype=torch.float64, device=audio.device)
        n = torch.arange(400, dtype=torch.float64, device=audio.device)
        k = torch.arange(201, dtype=torch.float64, device=audio.device)
        ang = 2.0 * math.pi * torch.outer(n, k) / 400.0
        fw = frames * win
        re = fw @ torch.cos(ang)
        im = fw @ torch.sin(ang)
        power = re * re + im * im                       # [B, T, 201]
        mel = power @ filters.to(torch.float64)         # [B, T, M]
        log_spec = torch.clamp(mel, min=1e-10).log10().transpose(1, 2)
        mx = log_spec.amax(dim=(1, 2), keepdim=True)
        log_spec = torch.maximum(log_spec, mx - 8.0)
        return ((log_spec + 4.0) / 4.0).to(torch.float32).contiguous()

    # nn.Linear under bf16 autocast: bf16 operands, fp32 accumulation (TF:modeling_whisper.py:279-282 etc.)
    def gemm(self, a, b, *, trans_a=False, trans_b=False, bias=None, act=0, want_z=False, zgrad=None, residual=None,
             r_row_mod=0, round_res=True, out_dtype=None, out=None, tile=0, atomic_acc=False, split_k=0, ln=None,
             kv_append=None, colsum=None, z_row_pad=0, out_row_pad=0, overwrite=False):
        # (z_row_pad / out_row_pad: memory layout of the HIP path's buffers, nothing to restate here; overwrite: with atomic_acc,
        # `out` is STORED instead of added to -- the engine only clears the accumulated ranges of the gradient buffer before such
        # a backward, so stale values of the previous step sit in `out`)
        out_dtype = self.lowp if out_dtype is None else out_dtype
        if ln is not None:                     # operand = bf16(LayerNorm(a)) (decode-step fusion of the HIP kernel)
            a = self.layernorm_fwd(a, ln[0], ln[1], ln[2] if len(ln) > 2 else 1e-5, save_stats=False)[0]
        A = a.float().t() if trans_a else a.float()
        Bm = b.float() if trans_b else b.float().t()
        v = A @ Bm
        if bias is not None:
            v = v + bias
        def dgelu(zz):
            cdf = 0.5 * (1.0 + torch.erf(zz * 0.7071067811865476))
            pdf = 0.3989422804014327 * torch.exp(-0.5 * zz * zz)
            return cdf + zz * pdf
        z = self._bf(v) if want_z else None
        if want_z == "grad":                   # the kernel's alternative by-product: gelu'(z) rounded to fp16
            z = dgelu(z.float())
            z = z.to(torch.float16) if self.lowp != torch.float32 else z      # exact-arithmetic mode keeps it exact
            z._is_gelu_grad = True
        if act == 1:
            v = F.gelu(self._bf(v).float())
        if zgrad is not None:
            stored = zgrad.dtype == torch.float16 or getattr(zgrad, "_is_gelu_grad", False)
            v = v * (zgrad.float() if stored else dgelu(zgrad.float()))
        if residual is not None:
            r = residual.float()
            if r_row_mod > 0:
                idx = torch.arange(v.shape[0], device=v.device) % r_row_mod
                r = r[idx]
            v = (self._bf(v).float() if round_res else v) + r
        v = v.to(out_dtype)
        if colsum is not None:                 # bias gradient of the producing Linear: column sums of the stored values
            colsum += v.float().sum(0)
        if kv_append is not None:              # columns >= split go to the K/V cache rows of their positions
            cache, split, rpb, pitch, row0 = kv_append
            m = torch.arange(v.shape[0], device=v.device)
            rows = (m // rpb) * pitch + row0 + (m % rpb)
            cache.view(-1, v.shape[1] - split)[rows] = v[:, split:].to(cache.dtype)
        if out is not None:
            if atomic_acc and not overwrite:
                out += v
            else:
                out.copy_(v)
            v = out
        return (v, z) if want_z else v

    # nn.LayerNorm in fp32 (autocast keeps layer_norm in fp32), output cast to bf16 by the consumer
    def layernorm_fwd(self, x, gamma, beta, eps=1e-5, save_stats=True, out=None):
        xf = x.float()
        mu = xf.mean(-1)
        var = ((xf - mu[:, None]) ** 2).mean(-1)
        rstd = torch.rsqrt(var + eps)
        y = (xf - mu[:, None]) * rstd[:, None] * gamma + beta
        y = self._bf(y)
        if out is not None:
            out.copy_(y)
            y = out
        return y, (mu if save_stats else None), (rstd if save_stats else None)

    def layernorm_bwd(self, dy, x, mean, rstd, gamma, dres, dgamma, dbeta, out_lowp=None, colsum=None):
        xf, d = x.float(), dy.float()
        xh = (xf - mean[:, None]) * rstd[:, None]
        g = d * gamma
        c1 = g.mean(-1, keepdim=True)
        c2 = (g * xh).mean(-1, keepdim=True)
        dx = rstd[:, None] * (g - c1 - xh * c2)
        dgamma += (d * xh).sum(0)
        dbeta += d.sum(0)
        if dres is None:
            dres = dx
        else:
            dres += dx
        if out_lowp is not None:
            out_lowp.copy_(self._bf(dres))
            if colsum is not None:
                colsum += out_lowp.float().sum(0)
        return dres

    # softmax(scale q k^T + mask) v  (TF:modeling_whisper.py:215-238), bf16 in/out, fp32 softmax
    @staticmethod
    def _heads(t, B, L, H):
        return t.reshape(B, L, H, 64).permute(0, 2, 1, 3).float()

    def attn_fwd(self, q, k, v, B, H, Lq, Lk, causal, scale, out=None, kv_batch_rows=None):
        if kv_batch_rows is not None and kv_batch_rows != Lk:  # padded KV cache: batch b starts at row b*kv_batch_rows
            k = k.reshape(B, kv_batch_rows, -1)[:, :Lk].reshape(B * Lk, -1)
            v = v.reshape(B, kv_batch_rows, -1)[:, :Lk].reshape(B * Lk, -1)
        qh, kh, vh = self._heads(q, B, Lq, H), self._heads(k, B, Lk, H), self._heads(v, B, Lk, H)
        s = (qh @ kh.transpose(-1, -2)) * scale
        if causal:    # 1 / True: key <= query; 2: bottom-right aligned (new queries against a longer KV cache)
            mask = torch.ones(Lq, Lk, dtype=torch.bool, device=s.device).tril(Lk - Lq if int(causal) == 2 else 0)
            s = s.masked_fill(~mask, float("-inf"))
        lse = torch.logsumexp(s, -1)
        p = torch.exp(s - lse[..., None])
        o = self._bf(p).float() @ vh
        o = self._bf(o.permute(0, 2, 1, 3).reshape(B * Lq, H * 64))
        if out is not None:
            out.copy_(o)
            o = out
        return o, lse

    def attn_fwd_varlen(self, q, k, v, H, max_q, q_start, q_len, causal, scale, out, Lk=0, kv_batches=0, self_attention=True,
                        flops=0.0):
        # ragged batches over packed rows (include/dwamd.h dw_attn_fwd_varlen): sequence i = rows q_start[i] ... + q_len[i]
        for i, (s0, n) in enumerate(zip(q_start.tolist(), q_len.tolist())):
            if n == 0:
                continue
            qi = q[s0:s0 + n]
            if self_attention:
                ki, vi, lk = k[s0:s0 + n], v[s0:s0 + n], n
            else:
                b = min(i, kv_batches - 1)
                ki, vi, lk = k[b * Lk:(b + 1) * Lk], v[b * Lk:(b + 1) * Lk], Lk
            o, _ = self.attn_fwd(qi, ki, vi, 1, H, n, lk, causal, scale)
            out[s0:s0 + n].copy_(o)
        return out

    def attn_bwd(self, q, k, v, o, do, lse, B, H, Lq, Lk, causal, scale, dq=None, dk=None, dv=None, dq_colsum=None,
                 dv_colsum=None):
        qh, kh, vh = self._heads(q, B, Lq, H), self._heads(k, B, Lk, H), self._heads(v, B, Lk, H)
        oh, doh = self._heads(o, B, Lq, H), self._heads(do, B, Lq, H)
        s = (qh @ kh.transpose(-1, -2)) * scale
        if causal:
            mask = torch.ones(Lq, Lk, dtype=torch.bool, device=s.device).tril()
            s = s.masked_fill(~mask, float("-inf"))
        p = torch.exp(s - lse[..., None])
        delta = (oh * doh).sum(-1, keepdim=True)
        dvh = self._bf(p).float().transpose(-1, -2) @ doh
        dp = doh @ vh.transpose(-1, -2)
        ds = self._bf(p * (dp - delta)).float()
        dqh = (ds @ kh) * scale
        dkh = (ds.transpose(-1, -2) @ qh) * scale

        def back(t, L):
            return self._bf(t.permute(0, 2, 1, 3).reshape(B * L, H * 64))

        rq, rk, rv = back(dqh, Lq), back(dkh, Lk), back(dvh, Lk)
        if dq is not None:
            dq.copy_(rq); rq = dq
        if dk is not None:
            dk.copy_(rk); rk = dk
        if dv is not None:
            dv.copy_(rv); rv = dv
        if dq_colsum is not None:
            dq_colsum += rq.float().sum(0)
        if dv_colsum is not None:
            dv_colsum += rv.float().sum(0)
        return rq, rk, rv

    # run_distillation.py:1453-1462, 1486-1493 + CrossEntropyLoss (TF:modeling_whisper.py:1083-1087), verbatim math
    def distill_loss(self, s_logits, t_logits, labels, V, temperature, ce_weight, kl_weight, grad_scale, want_grad,
                     grad_out=None, weights_dev=None):
        if weights_dev is not None:        # dw_distill_loss_w: the mix comes from device memory
            ce_weight, kl_weight = weights_dev[0].detach(), weights_dev[1].detach()
        with torch.enable_grad():      # (callers may sit inside an autograd.Function.forward, where grad mode is off)
            zs = s_logits[:, :V].float().detach().requires_grad_(want_grad)
            zt = t_logits[:, :V].float().detach()
            ce = F.cross_entropy(zs, labels, ignore_index=-100)
            teacher = F.softmax(zt / temperature, dim=-1)
            student = F.log_softmax(zs / temperature, dim=-1)
            div = F.kl_div(student, teacher, reduction="none")
            mask = (labels >= 0).unsqueeze(-1)
            kl = (div * mask).sum() / mask.sum() * temperature ** 2
            total = ce_weight * ce + kl_weight * kl
            if want_grad:
                (g,) = torch.autograd.grad(total * grad_scale, zs)
                dst = s_logits if grad_out is None else grad_out
                dst.zero_()
                dst[:, :V] = torch.nan_to_num(g).to(dst.dtype) if float((labels != -100).sum()) == 0 else g.to(dst.dtype)
        n = (labels != -100).sum().float()
        return torch.stack([ce.detach(), kl.detach(), total.detach(), n]).float()

    # GenerationMixin._sample with Whisper's logits processors (TF:generation/logits_process.py, installed by
    # TF:models/whisper/generation_whisper.py:1774-1812): min-new-tokens, begin-suppress, suppress, timestamp rules
    # (the restatement in distil_whisper_amd.decoding.apply_timestamp_rules is pinned against the transformers class in
    # tests/test_longform.py), argmax, finished rows filled with the pad token.
    def greedy_select(self, logits, V, tokens, n, cur, *, suppress=None, begin_suppress=None, first=False, no_eos=False,
                      forced=False, ts_begin=-1, max_initial=-1, begin_index=1, eos=-1, fill=-1, done=None):
        from distil_whisper_amd.decoding import apply_timestamp_rules
        B = tokens.shape[0]
        if forced:
            cur.copy_(tokens[:, n].view(B, 1))
            return
        neg = float("-inf")
        sc = logits[:B, :V].float()
        if no_eos and eos >= 0:
            sc[:, eos] = neg
        if first and begin_suppress is not None:
            sc = sc.masked_fill(begin_suppress[:V].bool()[None, :], neg)
        if suppress is not None:
            sc = sc.masked_fill(suppress[:V].bool()[None, :], neg)
        if ts_begin >= 0:
            sc = apply_timestamp_rules(sc, tokens, n, begin_index, ts_begin - 1, eos, None if max_initial < 0 else max_initial)
        nxt = sc.argmax(-1)
        if eos >= 0:
            nxt = torch.where(done, torch.full_like(nxt, fill), nxt)
            done.logical_or_(nxt == eos)
        tokens[:, n].copy_(nxt)
        cur.copy_(nxt.view(B, 1))

    def embed_fwd(self, ids, tok, pos, out_dtype, rows_alloc=0):
        B, T = ids.shape
        x = (tok.float()[ids.reshape(-1)] + pos.float()[:T].repeat(B, 1)).to(out_dtype)
        if rows_alloc > B * T:
            x = torch.cat([x, x.new_zeros(rows_alloc - B * T, x.shape[1])], 0)
        return x

    def embed_bwd(self, dx, ids, dtok, dpos):
        B, T = ids.shape
        dtok.index_add_(0, ids.reshape(-1), dx)
        if dpos is not None:
            dpos[:T] += dx.reshape(B, T, -1).sum(0)

    # conv1d(k=3, pad=1) as im2col (TF:modeling_whisper.py:566-567)
    def im2col_mel(self, mel, kpad, out=None):
        B, Cc, T = mel.shape
        xp = F.pad(mel, (1, 1))                                # [B, C, T+2]
        cols = torch.stack([xp[:, :, k:k + T] for k in range(3)], 1)  # [B, 3, C, T]
        cols = cols.permute(0, 3, 1, 2).reshape(B * T, 3 * Cc)
        res = torch.zeros(B * T, kpad, dtype=self.lowp, device=mel.device)
        res[:, : 3 * Cc] = self._bf(cols)
        if out is not None:
            out.copy_(res)
            return out
        return res

    def im2col_s2(self, a, B, T, out=None):
        Cc = a.shape[1]
        x = a.reshape(B, T, Cc)
        xp = F.pad(x, (0, 0, 1, 1))                             # [B, T+2, C]
        cols = torch.stack([xp[:, k:k + T:2] for k in range(3)], 2)   # [B, T/2, 3, C]
        res = cols.reshape(B * T // 2, 3 * Cc).contiguous()
        if out is not None:
            out.copy_(res)
            return out
        return res

    @staticmethod
    def _gelu_grad(zz):
        cdf = 0.5 * (1.0 + torch.erf(zz * 0.7071067811865476))
        pdf = 0.3989422804014327 * torch.exp(-0.5 * zz * zz)
        return cdf + zz * pdf

    def gelu_bwd(self, dy, z, out=None):
        res = self._bf(self._bf(dy).float() * self._gelu_grad(z.float()))
        if out is not None:
            out.copy_(res)
            return out
        return res

    def col2im_s2_gelu_bwd(self, dxcol, z, B, T, out=None):
        Cc = z.shape[1]
        d = dxcol.float().reshape(B, T // 2, 3, Cc)
        acc = torch.zeros(B, T + 2, Cc, dtype=torch.float32, device=z.device)
        for k in range(3):
            acc[:, k:k + T:2] += d[:, :, k]
        da = self._bf(acc[:, 1:T + 1].reshape(B * T, Cc)).float()
        zz = z.float()
        cdf = 0.5 * (1.0 + torch.erf(zz * 0.7071067811865476))
        pdf = 0.3989422804014327 * torch.exp(-0.5 * zz * zz)
        res = self._bf(da * (cdf + zz * pdf))
        if out is not None:
            out.copy_(res)
            return out
        return res

    def pack_conv_weight(self, w, kpad, out=None):
        D, Cc, _ = w.shape
        wp = torch.zeros(D, kpad, dtype=self.lowp, device=w.device)
        wp[:, : 3 * Cc] = self._bf(w.permute(0, 2, 1).reshape(D, 3 * Cc))
        if out is not None:
            out.copy_(wp)
            return out
        return wp

    def unpack_conv_grad(self, gwp, gw, accumulate):
        D, Cc, _ = gw.shape
        g = gwp[:, : 3 * Cc].reshape(D, 3, Cc).permute(0, 2, 1)
        if accumulate:
            gw += g
        else:
            gw.copy_(g)

    def cast_bf16(self, x, out=None):
        if out is not None:
            out.copy_(x)
            return out
        return self._bf(x)

    def cast_f32(self, x, out=None):
        if out is not None:
            out.copy_(x)
            return out
        return x.float()

    def colsum(self, x, out, accumulate):
        s = x.float().sum(0)
        if accumulate:
            out += s
        else:
            out.copy_(s)
        return out

    def gather_rows(self, src, idx, out):
        out[:idx.numel()] = src[idx.long()]
        return out

    def scatter_rows(self, src, idx, out):
        out[idx.long()] = src[:idx.numel()]
        return out

    def add(self, a, b, out_dtype):
        return (a.float() + b.float()).to(out_dtype)

    def sumsq(self, g, out):
        out += (g.double() ** 2).sum().float()
        return out

    # torch.optim.AdamW (single-tensor path) + clip_grad_norm_ coefficient (run_distillation.py:1377-1407, 1611)
    def adamw(self, p, g, m, v, shadow, sumsq, max_norm, grad_mul, lr, beta1, beta2, eps, weight_decay, step):
        clip = grad_mul
        if max_norm > 0 and sumsq is not None:
            norm = torch.sqrt(sumsq[0]) * abs(grad_mul)
            clip = grad_mul * torch.clamp(max_norm / (norm + 1e-6), max=1.0)
        gg = g * clip
        p.mul_(1.0 - lr * weight_decay)
        m.mul_(beta1).add_(gg, alpha=1.0 - beta1)
        v.mul_(beta2).addcmul_(gg, gg, value=1.0 - beta2)
        bc1 = 1.0 - beta1 ** step
        bc2 = 1.0 - beta2 ** step
        denom = v.sqrt() / math.sqrt(bc2) + eps
        p.addcdiv_(m, denom, value=-lr / bc1)
        if shadow is not None:
            shadow.copy_(p)

    # device-resident optimizer scalars (include/dwamd.h dw_adam_tick / dw_adamw_dev): same arithmetic as `adamw`
    def adam_state(self, lr, beta1, beta2, step=0):
        return torch.tensor([lr, step, beta1, beta2, 0, 0, 0, 0], dtype=torch.float64, device=self.device)

    def adam_tick(self, state, gate=None):
        apply = gate is None or float(gate.reshape(-1)[0]) > 0.0
        if apply:
            state[1] += 1.0
        step = max(float(state[1]), 1.0)
        state[4] = state[0] / (1.0 - float(state[2]) ** step)
        state[5] = math.sqrt(1.0 - float(state[3]) ** step)
        state[6] = 1.0 if apply else 0.0

    def adamw_dev(self, p, g, m, v, shadow, sumsq, max_norm, grad_mul, state, eps, weight_decay):
        if float(state[6]) == 0.0:
            return
        self.adamw(p, g, m, v, shadow, sumsq, max_norm, grad_mul, float(state[0]), float(state[2]), float(state[3]),
                   eps, weight_decay, int(state[1]))
