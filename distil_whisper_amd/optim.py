"""`torch.optim.Optimizer` over the fused clip + AdamW kernels, for users of the REFERENCE's training loop.

The reference builds `torch.optim.AdamW` over two parameter groups (decay / no decay for LayerNorm parameters and biases,
run_distillation.py:1377-1407), clips with `accelerator.clip_grad_norm_` (1611) and steps (1612-1614).  Over the drop-in
modules that costs ~44 ms of a 430 ms step at large-v3 (round 4: `via_reference_loop`): `clip_grad_norm_` and the
multi-tensor AdamW each make several passes over 3 GB of parameters, gradients and moments, and the bf16 GEMM operands are
re-cast afterwards.  `FusedAdamW` keeps the loop's shape --

    optimizer = FusedAdamW(optimizer_grouped_parameters, lr=..., betas=..., eps=...)          # instead of torch.optim.AdamW
    ...
    loss.backward()
    grad_norm = optimizer.clip_grad_norm_(max_grad_norm)      # instead of accelerator.clip_grad_norm_(params, max_grad_norm)
    optimizer.step(); lr_scheduler.step(); optimizer.zero_grad()

(Under `accelerate`, `accelerator.prepare(optimizer)` returns an `AcceleratedOptimizer` wrapper that forwards only the
`torch.optim.Optimizer` methods -- it has no `__getattr__`, so `optimizer.clip_grad_norm_` does not exist on it
(run_distillation.py:1449).  Use the module-level `clip_grad_norm_(optimizer, max_grad_norm)` below, which unwraps it, or
construct `FusedAdamW(..., max_grad_norm=...)` and delete the script's `accelerator.clip_grad_norm_` line: `step()` clips
by itself then.  Leaving `accelerator.clip_grad_norm_(student_model.parameters(), ...)` in place also works -- it scales
`p.grad` in place with torch's multi-tensor passes first -- it is merely the slow form.)

-- and runs ONE read of the gradients for the norm (`dw_sumsq_f32`) and ONE pass per parameter group segment
(`dw_adamw_dev`: clip coefficient from the device-resident norm, AdamW, bf16 shadow refresh) over the model's flat
parameter / moment buffers.  Same arithmetic as `DistillationTrainer.optimizer_step` (same kernels); `torch.optim.AdamW`
semantics: decoupled weight decay, bias corrections in double, a parameter without a gradient is not touched, `lr` is read
from the group every step (LR schedulers work unchanged), `state_dict()` / `load_state_dict()` round-trip the moments.
"""
import torch

__all__ = ["FusedAdamW", "clip_grad_norm_", "unwrap_optimizer"]


def unwrap_optimizer(optimizer):
    """The FusedAdamW behind `accelerate.optimizer.AcceleratedOptimizer` (its `.optimizer`) or a LR-scheduler-style wrapper;
    the optimizer itself when it is not wrapped."""
    seen = 0
    while not isinstance(optimizer, FusedAdamW) and hasattr(optimizer, "optimizer") and seen < 4:
        optimizer, seen = optimizer.optimizer, seen + 1
    return optimizer


def clip_grad_norm_(optimizer, max_norm, norm_type=2.0):
    """Line 1611 of run_distillation.py for a FusedAdamW that went through `accelerator.prepare`:

        grad_norm = clip_grad_norm_(optimizer, training_args.max_grad_norm)

    Returns the total gradient norm as a device scalar; the scaling is fused into the next `optimizer.step()`."""
    opt = unwrap_optimizer(optimizer)
    if not isinstance(opt, FusedAdamW):
        raise TypeError("clip_grad_norm_: expected a FusedAdamW (possibly wrapped by accelerate), got " + type(opt).__name__)
    return opt.clip_grad_norm_(max_norm, norm_type)


def _unwrap(model):
    while hasattr(model, "module") and not hasattr(model, "store"):
        model = model.module
    return model


class FusedAdamW(torch.optim.Optimizer):
    merge_segments = True      # False: one norm / update launch per (group, contiguous run of its tensors) -- the A/B of tools/ref_loop_profile.py

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, *, model, max_grad_norm=None):
        """`params`: parameters or parameter-group dicts of ONE `distil_whisper_amd.WhisperForConditionalGeneration`
        (`model`, possibly DDP-wrapped), as `torch.optim.AdamW` takes them.  `max_grad_norm`: clip inside `step()` without a
        separate `clip_grad_norm_` call (the norm is still available as `last_grad_norm`)."""
        if lr < 0.0 or eps < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0) or weight_decay < 0.0:
            raise ValueError("FusedAdamW: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.model = _unwrap(model)
        st = self.model.store
        if st.G is None:
            raise ValueError("FusedAdamW: the model has no trainable parameters (dtype=bfloat16 models are inference-only)")
        self.ops, self.st = self.model.ops, st
        by_id = {id(p): n for n, p in zip(self.model._param_names, self.model._param_list)}
        self._ranges = {}                       # id(param) -> (offset, padded end, name)
        for group in self.param_groups:
            for p in group["params"]:
                name = by_id.get(id(p))
                if name is None:
                    raise ValueError("FusedAdamW: every parameter must belong to `model` (a distil_whisper_amd module)")
                off, shape, _ = st.entries[name]
                n = 1
                for d in shape:
                    n *= d
                self._ranges[id(p)] = (off, off + ((n + 63) // 64) * 64, name)
        for group in self.param_groups:
            b1, b2 = group["betas"]
            group["_adam"] = self.ops.adam_state(group["lr"], b1, b2, 0)      # device-resident [lr, step, beta1, beta2, ...]
            group["_lr_dev"] = group["lr"]
        self._sumsq = self.ops.zeros((1,), torch.float32)
        self._segments = None
        self._seg_key = None
        self._norm_segments, self._step_plan, self._step_key = None, None, None
        self._gathered = False
        self._max_norm = float(max_grad_norm) if max_grad_norm else 0.0
        self._clip_once = None
        self.last_grad_norm = None

    # -- flat gradient -----------------------------------------------------------------------------------------------
    def _plan(self):
        """Contiguous [start, end) segments per group over the parameters that HAVE a gradient (torch.optim.AdamW skips the
        others entirely: no decay, no moment update)."""
        key = tuple(id(p) for g in self.param_groups for p in g["params"] if p.grad is not None)
        if key == self._seg_key:
            return
        # The reference's groups hold EVERY named parameter, also the ones it froze (run_distillation.py:1386-1401 does not
        # filter on requires_grad; torch.optim.AdamW skips a parameter without a gradient).  Parameters of the store's
        # frozen region (no gradient / moment storage: the sinusoidal encoder positions, a model built with
        # frozen_prefixes) are accepted in the groups and must simply never carry a gradient.
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is not None and self._ranges[id(p)][0] < self.st.train_start:
                    raise ValueError(f"FusedAdamW: {self._ranges[id(p)][2]} lies in the model's frozen region but has a gradient")
        plan = []
        for gi, group in enumerate(self.param_groups):
            rs = sorted(self._ranges[id(p)][:2] for p in group["params"] if p.grad is not None)
            segs = []
            for a, b in rs:
                if segs and segs[-1][1] == a:
                    segs[-1][1] = b
                else:
                    segs.append([a, b])
            plan.append(segs)
        self._segments, self._seg_key = plan, key
        # The gradient norm is ONE quantity over every parameter that has a gradient: the union of the groups' segments, merged
        # (the reference's two groups -- decay / no decay -- interleave tensor by tensor: 321 segments at distil-large-v3, whose
        # union is the single trainable range; summed per segment that was 321 x two launches per step, 3 ms of launch floors).
        self._norm_segments = self._merged([s for segs in plan for s in segs]) if self.merge_segments else [s for segs in plan for s in segs]
        self._step_plan, self._step_key = None, None

    @staticmethod
    def _merged(segs):
        out = []
        for a, b in sorted((a, b) for a, b in segs):
            if out and out[-1][1] >= a:
                out[-1][1] = max(out[-1][1], b)
            else:
                out.append([a, b])
        return out

    def _classes(self):
        """Groups whose update is the same function of (p, g, m, v) -- equal lr, betas, eps, weight decay and step count -- are
        stepped together over their MERGED segments (the reference's default weight_decay = 0 makes its two groups one class:
        one launch over the trainable range instead of one per tensor).  Each group keeps its own device-resident state."""
        key = tuple((g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"], g.get("_steps", 0)) for g in self.param_groups)
        if self._step_plan is not None and key == self._step_key:
            return self._step_plan
        by = {}
        for gi, k in enumerate(key):
            if self._segments[gi]:
                by.setdefault(k if self.merge_segments else (gi,), []).append(gi)
        plan = [(gis, self._merged([s for gi in gis for s in self._segments[gi]])) for gis in by.values()]
        self._step_plan, self._step_key = plan, key
        return plan

    def _gather(self):
        """p.grad (separate tensors: autograd clones, DDP bucket views, accumulated micro-batches) -> the flat buffer the
        kernels read; one multi-tensor copy."""
        if self._gathered:
            return
        self._plan()
        dst, src = [], []
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is not None:
                    a, _, name = self._ranges[id(p)]
                    d = self.st.G[a:a + p.numel()].view(p.shape)
                    # (the drop-in's backward hands out views of the flat buffer as gradients: already in place -- a self-copy of
                    # 3 GB would cost a read and a write of every gradient, 1.3 ms per step at distil-large-v3)
                    if p.grad.data_ptr() == d.data_ptr() and p.grad.dtype == d.dtype and p.grad.is_contiguous():
                        continue
                    dst.append(d)
                    src.append(p.grad)
        if dst:
            torch._foreach_copy_(dst, src)
        self._sumsq.zero_()
        for a, b in self._norm_segments:
            self.ops.sumsq(self.st.G[a:b], self._sumsq)
        self.last_grad_norm = torch.sqrt(self._sumsq[0])
        self._gathered = True

    @torch.no_grad()
    def clip_grad_norm_(self, max_norm, norm_type=2.0):
        """`torch.nn.utils.clip_grad_norm_` / `accelerator.clip_grad_norm_` for this optimizer's parameters: returns the total
        gradient norm (device scalar, no host sync); the scaling by min(1, max_norm / (norm + 1e-6)) happens inside the next
        `step()` (fused into the update: `p.grad` itself is left as it is)."""
        if float(norm_type) != 2.0:
            raise NotImplementedError("FusedAdamW.clip_grad_norm_: the fused kernels compute the 2-norm")
        self._gather()
        self._clip_once = float(max_norm)
        return self.last_grad_norm

    # -- step --------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._gather()
        max_norm = self._clip_once if self._clip_once is not None else self._max_norm
        st, ops = self.st, self.ops
        plan = self._classes()                               # (before the step counts move: they are part of the class key)
        for group, segs in zip(self.param_groups, self._segments):
            if not segs:
                continue
            if group["lr"] != group["_lr_dev"]:              # an LR scheduler moved it
                group["_adam"][0:1].fill_(group["lr"])
                group["_lr_dev"] = group["lr"]
            ops.adam_tick(group["_adam"], None)
            group["_steps"] = group.get("_steps", 0) + 1
        for gis, segs in plan:
            group = self.param_groups[gis[0]]                # (every group of the class holds the same scalars)
            for a, b in segs:
                ops.adamw_dev(st.P[a:b], st.G[a:b], st.M[a:b], st.V[a:b], st.S[a:b], self._sumsq, max_norm, 1.0, group["_adam"],
                              group["eps"], group["weight_decay"])
        if any(self._ranges[id(p)][2].startswith("model.encoder.conv") for g in self.param_groups for p in g["params"]
               if p.grad is not None):
            st.repack_conv()                                  # conv weights are consumed in a packed GEMM layout
        self._clip_once = None
        self._gathered = False
        return loss

    def zero_grad(self, set_to_none=True):
        self._gathered = False
        super().zero_grad(set_to_none=set_to_none)

    # -- checkpointing (accelerator.save_state / load_state) -------------------------------------------------------------
    _PRIVATE_KEYS = ("params", "_adam", "_lr_dev", "_steps")

    def state_dict(self):
        """Flat form: the moments of the whole trainable range in two tensors + the parameter groups (every public key,
        e.g. LambdaLR's `initial_lr`; `params` holds parameter NAMES).  `torch_state_dict()` is the per-parameter
        torch.optim.AdamW layout."""
        st = self.st
        lo, hi = st.train_start, st.train_end
        groups = [{k: v for k, v in g.items() if k not in self._PRIVATE_KEYS} for g in self.param_groups]
        for g, src in zip(groups, self.param_groups):
            g["params"] = [self._ranges[id(p)][2] for p in src["params"]]
            g["step"] = float(src["_adam"][1].item())
        return {"state": {"exp_avg": st.M[lo:hi].clone(), "exp_avg_sq": st.V[lo:hi].clone(), "range": (lo, hi)},
                "param_groups": groups}

    def _restore_groups(self, saved_groups, steps):
        if len(saved_groups) != len(self.param_groups):
            raise ValueError(f"FusedAdamW.load_state_dict: the checkpoint has {len(saved_groups)} parameter groups, this "
                             f"optimizer {len(self.param_groups)}")
        for g, saved, step in zip(self.param_groups, saved_groups, steps):
            for k, v in saved.items():                  # every saved key (lr, betas, eps, weight_decay, initial_lr, ...)
                if k not in self._PRIVATE_KEYS and k != "step":
                    g[k] = v
            b1, b2 = g["betas"]
            g["_adam"] = self.ops.adam_state(g["lr"], b1, b2, step)
            g["_lr_dev"] = g["lr"]
            g["_steps"] = int(step)
        self._step_plan, self._step_key = None, None

    def load_state_dict(self, state_dict):
        """Accepts this class's flat form and the torch.optim.AdamW layout (`state` keyed by parameter index with
        `step` / `exp_avg` / `exp_avg_sq`): a checkpoint written by the reference's run with torch.optim.AdamW over the same
        parameter groups resumes here, and `torch_state_dict()` goes the other way."""
        state = state_dict["state"]
        if "range" not in state:
            return self.load_torch_state_dict(state_dict)
        st = self.st
        lo, hi = state["range"]
        if (lo, hi) != (st.train_start, st.train_end):
            raise ValueError(f"FusedAdamW.load_state_dict: the checkpoint covers the flat range [{lo}, {hi}), this model's "
                             f"trainable range is [{st.train_start}, {st.train_end}) -- another frozen layout (freeze_encoder / "
                             f"requires_grad_ before constructing the model's store); use torch_state_dict() / "
                             f"load_torch_state_dict() to move moments between layouts by parameter")
        st.M[lo:hi].copy_(state["exp_avg"])
        st.V[lo:hi].copy_(state["exp_avg_sq"])
        self._restore_groups(state_dict["param_groups"], [g.get("step", 0.0) for g in state_dict["param_groups"]])

    def torch_state_dict(self):
        """The torch.optim.AdamW layout of the same state: {"state": {index: {"step", "exp_avg", "exp_avg_sq"}},
        "param_groups": [... "params": [indices]]}; indices count the parameters group by group as torch does.  (`step` of a
        parameter that never had a gradient is still the group's count: this optimizer keeps one count per group.)"""
        st, state, groups, idx = self.st, {}, [], 0
        for g in self.param_groups:
            step = float(g["_adam"][1].item())
            out = {k: v for k, v in g.items() if k not in self._PRIVATE_KEYS}
            out["params"] = []
            for p in g["params"]:
                a = self._ranges[id(p)][0]
                if step > 0 and p.requires_grad and a >= st.train_start:     # (torch holds no state for a parameter it never stepped)
                    state[idx] = {"step": torch.tensor(step), "exp_avg": st.M[a:a + p.numel()].view(p.shape).clone(),
                                  "exp_avg_sq": st.V[a:a + p.numel()].view(p.shape).clone()}
                out["params"].append(idx)
                idx += 1
            groups.append(out)
        return {"state": state, "param_groups": groups}

    def load_torch_state_dict(self, state_dict):
        st = self.st
        saved = state_dict["param_groups"]
        if len(saved) != len(self.param_groups) or any(len(a["params"]) != len(b["params"]) for a, b in zip(saved, self.param_groups)):
            raise ValueError("FusedAdamW.load_torch_state_dict: the checkpoint's parameter groups do not match this optimizer's "
                             "(same grouping and order as the optimizer that wrote it are required, as for torch.optim)")
        steps = []
        for g, sg in zip(self.param_groups, saved):
            step = 0.0
            for p, i in zip(g["params"], sg["params"]):
                ps = state_dict["state"].get(i)
                a = self._ranges[id(p)][0]
                if a < st.train_start:
                    continue
                if ps is None:
                    st.M[a:a + p.numel()].zero_()
                    st.V[a:a + p.numel()].zero_()
                    continue
                if tuple(ps["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"FusedAdamW.load_torch_state_dict: moment shape {tuple(ps['exp_avg'].shape)} for "
                                     f"{self._ranges[id(p)][2]} {tuple(p.shape)}")
                st.M[a:a + p.numel()].view(p.shape).copy_(ps["exp_avg"])
                st.V[a:a + p.numel()].view(p.shape).copy_(ps["exp_avg_sq"])
                step = max(step, float(ps["step"]))
            steps.append(step)
        self._restore_groups(saved, steps)
