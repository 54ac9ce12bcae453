"""Lazy `.logits` of the drop-in `WhisperForConditionalGeneration`: the reference's OWN loss lines, unedited, on the fused kernel.

The reference computes the distillation loss with five tensor expressions over the two models' fp32 `[B, T, V]` logits
(run_distillation.py:1453-1462, 1486-1493):

    teacher_distribution = nn.functional.softmax(teacher_outputs.logits / temperature, dim=-1)
    student_distribution = nn.functional.log_softmax(student_outputs.logits / temperature, dim=-1)
    divergence = nn.KLDivLoss(reduction="none")(student_distribution, teacher_distribution) * (labels >= 0).unsqueeze(-1)
    kl_loss = divergence.sum() / padding_mask.sum() * temperature**2
    loss = 0.8 * student_outputs.loss + kl_weight * kl_loss

At large-v3 that is ~20 passes over 2.97 GB temporaries per step (forward and autograd's backward through them), although the
values live in the engines' bf16 logit buffers and ONE pass of `dw_distill_loss` computes CE, KL and d(loss)/d(logits)
(round 5: 412 ms per step through the verbatim loop against 368 with the one-call `fused_distillation_loss`, which needs a
script edit).  This module removes the edit: `.logits` of the drop-in model is a `torch.Tensor` subclass whose storage is an
UNFILLED fp32 `[B, T, V]` allocation.  `__torch_function__` recognises exactly the expression above --

    LazyLogits  --/ T-->  scaled  --softmax / log_softmax(dim=-1)-->  distribution
    F.kl_div(log_softmax(student), softmax(teacher), reduction="none")  -->  divergence  --* mask[B, T, 1]-->  masked  --.sum()

-- and answers the final `.sum()` with the fused kernel's value (as a device scalar tied into autograd).  ANY other use of
the tensor (indexing, `argmax`, `float()`, arithmetic, `torch.cat`, printing, a temperature that is a tensor, another `dim`,
another reduction ...) first fills the allocation from the bf16 buffer and then runs the requested function on the real
tensor: exactly what the eager model returned before, so nothing can observe a difference except the time.

Backward: the CE node (`.loss`) and the KL-sum node do not return `[B, T, V]` gradients.  Each records the upstream gradient it
received (a device scalar: autograd's d(loss)/d(ce), d(loss)/d(sum)) in the forward's shared state and returns None; the
engine node (`modeling._EngineFn.backward`) then finds `g_logits is None`, builds the device-resident mix
`[g_ce, g_sum * n_valid / T^2]` and runs ONE pass of `dw_distill_loss_w`, which writes d(loss)/d(logits) in bf16 over the
logits buffer -- what the native trainer does.  A real gradient arriving as well (the caller also used the materialised
tensor) is added on top.  Masks that are not `labels >= 0` make the returned sum NaN instead of silently wrong (the fused
pass keys both terms on the labels; the check is a device-side comparison, no host sync).
"""
import numbers

import torch
import torch.nn.functional as F

__all__ = ["LazyLogits", "LazyState", "STATS"]

# process-wide counters (tests and bench.py read them): `.sum()` calls answered by the fused kernel, fp32 [B, T, V] fills,
# backward passes whose d(loss)/d(logits) came from the fused kernel
STATS = {"lazy_sums": 0, "fills": 0, "lazy_backwards": 0}


class LazyState:
    """One forward's shared record: where the values live, and what the loss nodes hand to the engine's backward."""

    def __init__(self, model, buf, sel, shape, requires_grad):
        self.model, self.buf, self.sel, self.shape = model, buf, sel, shape      # buf: bf16 [rows >= R, ldv]; sel: _RowSel or None
        self.requires_grad = requires_grad
        self.labels = None          # labels the forward was called with (CE)
        self.filled = False
        self.ce_g = None            # upstream gradient of the CE node (0-dim device tensor)
        self.kl = []                # [(g_sum, teacher rows, temperature, labels_kl rows, n_kl)] of the KL-sum nodes
        self.fused = None           # (g_total, ce_w, kl_w, teacher rows, temperature) of fused_distillation_loss
        self.lazy_sums = 0          # how many `.sum()` calls were answered by the fused kernel (tests / bench read it)

    @property
    def rows(self):
        B, T, _ = self.shape
        return B * T if self.sel is None else self.sel.n

    def select_rows(self, x2d):
        """[B * T, C] in (batch, position) order -> the rows the engine computed"""
        return x2d if self.sel is None else self.sel.select(x2d)

    def fill(self, real):
        """fp32 [B, T, V] <- the bf16 buffer (zero rows at positions that were not computed)"""
        if self.filled:
            return
        B, T, V = self.shape
        with torch.no_grad():
            if self.sel is None:
                real.copy_(self.buf[: B * T, :V].view(B, T, V))
            else:
                real.copy_(self.sel.expand(self.buf, V))
        self.filled = True
        STATS["fills"] += 1

    def pending(self):
        return self.ce_g is not None or bool(self.kl) or self.fused is not None


def _plain(x):
    return x.as_subclass(torch.Tensor) if isinstance(x, LazyLogits) else x


def _materialise(x):
    """LazyLogits -> the filled real tensor; expression node -> its value by the reference's own torch ops; else x."""
    if isinstance(x, LazyLogits):
        st = getattr(x, "_dw", None)
        real = x.as_subclass(torch.Tensor)
        if st is not None:
            st.fill(real)
        return real
    if isinstance(x, _Expr):
        return x.materialise()
    return x


def _tree(fn, obj):
    if isinstance(obj, (list, tuple)):
        return type(obj)(_tree(fn, o) for o in obj)
    if isinstance(obj, dict):
        return {k: _tree(fn, v) for k, v in obj.items()}
    return fn(obj)


_META = {"shape", "dtype", "device", "requires_grad", "grad_fn", "ndim", "is_cuda", "layout", "is_leaf", "names", "is_meta",
         "is_sparse", "is_quantized", "grad", "_version", "output_nr", "is_cpu", "_dw", "retains_grad", "itemsize", "nbytes"}
_META_FN = {"size", "dim", "stride", "numel", "nelement", "is_floating_point", "is_complex", "element_size", "is_contiguous",
            "storage_offset", "get_device", "ndimension", "type", "__len__", "__hash__", "__reduce_ex__", "_is_view", "is_shared",
            "requires_grad_", "register_hook", "retain_grad", "is_same_size", "as_subclass"}
_DIV = {"div", "true_divide", "__truediv__", "divide"}
_MUL = {"mul", "multiply", "__mul__", "__rmul__"}


def _fname(func):
    n = getattr(func, "__name__", None)
    if n == "__get__":                      # property of torch.Tensor: the descriptor's own name
        return getattr(getattr(func, "__self__", None), "__name__", "__get__"), True
    return n, False


def _dispatch(func, args, kwargs):
    """The recognised steps of the reference's KD expression.  Returns NotImplemented for everything else."""
    name, is_prop = _fname(func)
    a0 = args[0] if args else None
    if is_prop or name in _META_FN:
        if isinstance(a0, LazyLogits) and (name in _META or name in _META_FN):
            with torch._C.DisableTorchFunctionSubclass():       # metadata only: never fills
                return func(*args, **kwargs)
        return NotImplemented
    if name == "float" and len(args) == 1 and not kwargs and isinstance(a0, LazyLogits) and _st(a0) is not None:
        return a0                               # already fp32 (`logits.float()` of code written for autocast outputs): still lazy
    if name == "to" and isinstance(a0, LazyLogits) and _st(a0) is not None and len(args) == 2 and not kwargs and args[1] is torch.float32:
        return a0
    if name in _DIV and len(args) == 2 and not kwargs:
        x, t = args
        if isinstance(x, LazyLogits) and _st(x) is not None and isinstance(t, numbers.Real) and not isinstance(t, bool) and t > 0:
            return _Scaled(x, float(t))
        return NotImplemented
    if name in ("softmax", "log_softmax"):
        x = a0
        dim = kwargs.get("dim", args[1] if len(args) > 1 else None)
        dtype = kwargs.get("dtype", args[3] if len(args) > 3 else None)
        if isinstance(x, LazyLogits) and _st(x) is not None:
            x = _Scaled(x, 1.0)
        if isinstance(x, _Scaled) and dim in (-1, 2) and dtype in (None, torch.float32):
            return _Dist(name, x)
        return NotImplemented
    if name == "kl_div":
        inp = a0
        tgt = args[1] if len(args) > 1 else kwargs.get("target")
        red = kwargs.get("reduction", args[4] if len(args) > 4 else "mean")
        log_target = kwargs.get("log_target", args[5] if len(args) > 5 else False)
        if (isinstance(inp, _Dist) and isinstance(tgt, _Dist) and inp.kind == "log_softmax" and tgt.kind == "softmax" and
                red == "none" and not log_target and kwargs.get("size_average") is None and kwargs.get("reduce") is None and
                inp.x.temperature == tgt.x.temperature and inp.x.state.shape == tgt.x.state.shape):
            return _Div(inp.x, tgt.x)
        return NotImplemented
    if name in _MUL and len(args) == 2 and not kwargs:
        d, m = (args[0], args[1]) if isinstance(args[0], _Div) else (args[1], args[0])
        if isinstance(d, _Div) and torch.is_tensor(m) and not isinstance(m, LazyLogits):
            B, T, _ = d.student.state.shape
            if tuple(m.shape) == (B, T, 1) and m.dtype in (torch.bool, torch.int64, torch.int32, torch.uint8):
                return _Masked(d, m)
        return NotImplemented
    if name == "sum" and len(args) == 1 and not any(v is not None for v in kwargs.values()) and isinstance(a0, _Masked):
        return _kl_sum(a0)
    return NotImplemented


def _st(x):
    return getattr(x, "_dw", None)


def _fallback(func, args, kwargs):
    with torch._C.DisableTorchFunctionSubclass():
        return func(*_tree(_materialise, args), **_tree(_materialise, kwargs))


class LazyLogits(torch.Tensor):
    """fp32 [B, T, V] logits whose values live in the engine's bf16 buffer until something other than the reference's loss
    expression reads them (module docstring).  Created by `LazyLogits.wrap(real, state)`."""

    @staticmethod
    def wrap(real, state):
        lz = real.as_subclass(LazyLogits)        # same storage, stays attached to the autograd graph
        lz._dw = state
        return lz

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        r = _dispatch(func, args, kwargs)
        if r is not NotImplemented:
            return r
        return _fallback(func, args, kwargs)

    def materialize(self):
        """The filled plain tensor (same storage, same autograd node)."""
        return _materialise(self)


class _Expr:
    """Tensor-like node of the recognised expression (any object with __torch_function__ takes part in torch.* dispatch).
    Python operators and the few methods the reference's lines use are spelled out; everything else materialises."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        r = _dispatch(func, args, kwargs)
        if r is not NotImplemented:
            return r
        return _fallback(func, args, kwargs)

    def _bin(self, fn, other, swap=False):
        a, b = (other, self) if swap else (self, other)
        r = _dispatch(fn, (a, b), {})
        return r if r is not NotImplemented else _fallback(fn, (a, b), {})

    def __mul__(self, o): return self._bin(torch.mul, o)
    def __rmul__(self, o): return self._bin(torch.mul, o, True)
    def __truediv__(self, o): return self._bin(torch.div, o)
    def __rtruediv__(self, o): return self._bin(torch.div, o, True)
    def __add__(self, o): return self._bin(torch.add, o)
    def __radd__(self, o): return self._bin(torch.add, o, True)
    def __sub__(self, o): return self._bin(torch.sub, o)
    def __rsub__(self, o): return self._bin(torch.sub, o, True)
    def __neg__(self): return -self.materialise()
    def __getitem__(self, i): return self.materialise()[i]
    def __len__(self): return len(self.materialise())

    def sum(self, *a, **k):
        r = _dispatch(torch.sum, (self, *a), k)
        return r if r is not NotImplemented else self.materialise().sum(*a, **k)

    def __getattr__(self, name):                 # any other tensor attribute / method: on the real value
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return getattr(self.materialise(), name)


class _Scaled(_Expr):
    def __init__(self, lz, temperature):
        self.lz, self.temperature, self.state = lz, temperature, lz._dw

    def materialise(self):
        real = _materialise(self.lz)
        return real if self.temperature == 1.0 else real / self.temperature


class _Dist(_Expr):
    def __init__(self, kind, x):
        self.kind, self.x = kind, x

    def materialise(self):
        return (F.softmax if self.kind == "softmax" else F.log_softmax)(self.x.materialise(), dim=-1)


class _Div(_Expr):
    def __init__(self, student, teacher):
        self.student, self.teacher = student, teacher

    def materialise(self):
        return F.kl_div(F.log_softmax(self.student.materialise(), dim=-1), F.softmax(self.teacher.materialise(), dim=-1),
                        reduction="none")


class _Masked(_Expr):
    def __init__(self, div, mask):
        self.div, self.mask = div, mask

    def materialise(self):
        return self.div.materialise() * self.mask


class _CEFn(torch.autograd.Function):
    """`.loss`: token-mean CE over labels != -100 (TF:modeling_whisper.py:1083-1087) from one forward-only pass of the fused
    kernel.  backward: the upstream gradient goes into the shared state; no [B, T, V] gradient is produced here."""

    @staticmethod
    def forward(ctx, logits, state, lab_rows):
        ctx.set_materialize_grads(False)
        ctx.state = state
        R = state.rows
        losses = state.model.ops.distill_loss(state.buf[:R], state.buf[:R], lab_rows, state.shape[2], 1.0, 1.0, 0.0, 1.0, False)
        return losses[0].clone()

    @staticmethod
    def backward(ctx, g):
        if g is not None:
            st = ctx.state
            st.ce_g = g if st.ce_g is None else st.ce_g + g
        return None, None, None


class _KLSumFn(torch.autograd.Function):
    """sum over (b, t, v) of mask * softmax(z_t / T) * (log softmax(z_t / T) - log softmax(z_s / T)): the value the
    reference's `divergence.sum()` has, from one forward-only pass of the fused kernel over the two bf16 buffers."""

    @staticmethod
    def forward(ctx, logits, state, t_rows, temperature, lab_kl, n_kl, bad):
        ctx.set_materialize_grads(False)
        R = state.rows
        losses = state.model.ops.distill_loss(state.buf[:R], t_rows, lab_kl, state.shape[2], temperature, 0.0, 1.0, 1.0, False)
        s = losses[1] * (n_kl / (temperature * temperature))          # kernel: kl = sum / n_kl * T^2
        s = torch.where(n_kl > 0, s, torch.zeros_like(s))             # (an all-masked batch sums to 0, as the reference's does)
        s = torch.where(bad, torch.full_like(s, float("nan")), s)
        ctx.state, ctx.payload = state, (t_rows, temperature, lab_kl, n_kl)
        state.lazy_sums += 1
        STATS["lazy_sums"] += 1
        return s

    @staticmethod
    def backward(ctx, g):
        if g is not None:
            ctx.state.kl.append((g,) + ctx.payload)
        return (None,) * 7


class _FusedFn(torch.autograd.Function):
    """`fused_distillation_loss`: CE + KL + their mix from one forward-only kernel pass; the gradient pass runs in the
    engine node's backward with the mix scaled by the upstream gradient (device weights)."""

    @staticmethod
    def forward(ctx, logits, state, t_rows, lab, temperature, ce_w, kl_w):
        ctx.set_materialize_grads(False)
        R = state.rows
        have_t = t_rows is not None
        losses = state.model.ops.distill_loss(state.buf[:R], t_rows if have_t else state.buf[:R], lab, state.shape[2], temperature,
                                              ce_w, kl_w if have_t else 0.0, 1.0, False)
        ctx.mark_non_differentiable(losses)
        ctx.state, ctx.payload = state, (ce_w, kl_w if have_t else 0.0, t_rows if have_t else state.buf[:R], temperature, lab)
        return (losses[2] if have_t else losses[0]).clone(), losses

    @staticmethod
    def backward(ctx, g, _g_losses):
        if g is not None:
            ctx.state.fused = (g,) + ctx.payload
        return (None,) * 7


def _kl_sum(masked):
    s_x, t_x = masked.div.student, masked.div.teacher
    st, tt = s_x.state, t_x.state
    B, T, V = st.shape
    # the teacher's rows in the student's row selection (fused_distillation_loss does the same)
    if st.sel is None and tt.sel is None:
        t_rows = tt.buf[: B * T]
    elif st.sel is not None and tt.sel is None:
        t_rows = st.sel.select(tt.buf[: B * T]).contiguous()
    elif st.sel is not None and tt.sel is not None and st.sel.same_as(tt.sel):
        t_rows = tt.buf[: st.rows]
    else:
        return masked.materialise().sum()        # different decoder positions computed: the exact slow way
    if t_rows.shape[1] != st.buf.shape[1]:
        return masked.materialise().sum()
    t_rows = t_rows[: st.rows]
    if not t_rows.is_contiguous():
        t_rows = t_rows.contiguous()
    mask = masked.mask.reshape(B * T).to(torch.bool)
    lab_kl = st.select_rows(torch.where(mask, 0, -100).to(torch.int64).view(B * T, 1)).reshape(-1).contiguous()
    n_kl = (lab_kl >= 0).sum().to(torch.float32)
    if st.labels is not None:
        # both terms of the one backward pass are keyed on the forward's labels: a mask other than `labels >= 0` must not
        # pass silently (device-side comparison; NaN is loud)
        bad = (mask != (st.labels.reshape(B * T) >= 0)).any()
    else:
        bad = torch.zeros((), dtype=torch.bool, device=mask.device)
    return _KLSumFn.apply(_plain(s_x.lz), st, t_rows, float(s_x.temperature), lab_kl, n_kl, bad)


def lazy_backward(state, buf, g_logits):
    """Called by the engine node's backward when loss nodes left their upstream gradients in `state`: writes
    d(loss)/d(logits) (bf16) over `buf` = the student's logits buffer with ONE pass of the fused kernel.  Returns False when
    nothing is pending (the caller then uses `g_logits` the ordinary way)."""
    if not state.pending():
        return False
    ops = state.model.ops
    B, T, V = state.shape
    R = state.rows
    dev = buf.device
    zero = torch.zeros((), dtype=torch.float32, device=dev)
    extra = None
    if g_logits is not None:                      # the caller ALSO differentiated through the materialised tensor
        extra = state.select_rows(g_logits.reshape(B * T, V))
    lab_ce = None
    if state.labels is not None:
        lab_ce = state.select_rows(state.labels.reshape(B * T, 1)).reshape(-1).contiguous()
    passes = []
    if state.fused is not None:
        g, ce_w, kl_w, t_rows, temp, lab = state.fused
        passes.append((torch.stack([g * ce_w, g * kl_w]).float(), t_rows, temp, lab))
    kls = list(state.kl)
    if state.ce_g is not None or kls:
        ce = state.ce_g.float() if state.ce_g is not None else zero
        if kls:
            g, t_rows, temp, lab_kl, n_kl = kls.pop(0)
            w_kl = g.float() * n_kl / (temp * temp)
            # one pass for CE + the (first) KL term: keyed on the forward's labels (mask == labels >= 0 was checked forward)
            lab = lab_ce if lab_ce is not None else lab_kl
            passes.append((torch.stack([ce, w_kl]).float(), t_rows, temp, lab))
        else:
            passes.append((torch.stack([ce, zero]).float(), buf[:R], 1.0, lab_ce))
    for g, t_rows, temp, lab_kl, n_kl in kls:      # further KL terms (not in the reference): one pass each
        passes.append((torch.stack([zero, g.float() * n_kl / (temp * temp)]).float(), t_rows, temp, lab_kl))
    first = True
    for w, t_rows, temp, lab in passes:
        if first and len(passes) == 1:
            ops.distill_loss(buf[:R], t_rows, lab, V, temp, 0.0, 0.0, 1.0, True, weights_dev=w.contiguous())   # in place
        else:
            tmp = ops.empty(tuple(buf[:R].shape), buf.dtype)
            ops.distill_loss(buf[:R], t_rows, lab, V, temp, 0.0, 0.0, 1.0, True, grad_out=tmp, weights_dev=w.contiguous())
            acc = tmp.float() if first else acc + tmp.float()
        first = False
    if len(passes) > 1:
        buf[:R].copy_(acc)
    if buf.shape[0] > R:
        buf[R:].zero_()
    if extra is not None:
        buf[:R, :V] += extra.to(buf.dtype)
    state.ce_g, state.kl, state.fused = None, [], None
    STATS["lazy_backwards"] += 1
    return True
