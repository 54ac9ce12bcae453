"""TEST INFRASTRUCTURE -- the reference's training / eval loop body, restated VERBATIM as the checker for the drop-in
boundary: run_distillation.py:760-778 (`get_parameter_names`), 1377-1407 (AdamW parameter groups), 1453-1462
(`kl_divergence`), 1465-1495 (`train_step`), 1498-1522 (`eval_step`), 1606-1614 (backward, clip, optimizer step,
zero_grad).  `accelerator.backward` / `accelerator.clip_grad_norm_` are what accelerate resolves them to without mixed
precision scaling: `loss.backward()` and `torch.nn.utils.clip_grad_norm_`; `accelerator.prepare(model)` is the optional
`wrap` (DistributedDataParallel in the GPU test).

The loop only touches the two model objects through the surface SURVEY.md 8(b) lists, so the same function drives the
`transformers` classes (fixtures, live CPU comparison) and `distil_whisper_amd.modeling` (the product): the tests in
tests/test_reference_loop.py compare the two.  Nothing on the product path imports this module.
"""
import torch
import torch.nn as nn


def get_parameter_names(model, forbidden_layer_types, forbidden_module=None):
    """run_distillation.py:760-778."""
    result = []
    for name, child in model.named_children():
        if forbidden_module is not None and isinstance(child, tuple(forbidden_module)):
            continue
        result += [f"{name}.{n}" for n in get_parameter_names(child, forbidden_layer_types, forbidden_module)
                   if not isinstance(child, tuple(forbidden_layer_types))]
    result += list(model._parameters.keys())
    return result


def make_optimizer(student_model, learning_rate=1e-4, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8, optimizer_cls=None):
    """run_distillation.py:1377-1407.  `optimizer_cls(groups, lr=, betas=, eps=)` replaces torch.optim.AdamW (the drop-in
    distil_whisper_amd.optim.FusedAdamW bound to its model); the parameter groups are the reference's either way."""
    decay_parameters = get_parameter_names(student_model, [nn.LayerNorm])
    decay_parameters = [name for name in decay_parameters if "bias" not in name]
    optimizer_grouped_parameters = [
        {"params": [p for n, p in student_model.named_parameters() if n in decay_parameters and p.requires_grad],
         "weight_decay": weight_decay},
        {"params": [p for n, p in student_model.named_parameters() if n not in decay_parameters and p.requires_grad],
         "weight_decay": 0.0},
    ]
    if optimizer_cls is not None:
        return optimizer_cls(optimizer_grouped_parameters, lr=learning_rate, betas=betas, eps=eps)
    return torch.optim.AdamW(params=optimizer_grouped_parameters, lr=learning_rate, betas=betas, eps=eps)


def kl_divergence(target_distribution, log_predicted_distribution, labels):
    """run_distillation.py:1453-1462."""
    kl_loss = nn.KLDivLoss(reduction="none")
    divergence = kl_loss(log_predicted_distribution, target_distribution)
    padding_mask = labels >= 0
    padding_mask = padding_mask.unsqueeze(-1)
    divergence = divergence * padding_mask
    divergence = divergence.sum() / padding_mask.sum()
    return divergence


class ReferenceLoop:
    def __init__(self, student_model, teacher_model, BaseModelOutput, *, share_hidden_states=False,
                 teacher_dtype=torch.float32, kl_weight=1.0, max_grad_norm=1.0, learning_rate=1e-4, weight_decay=0.0,
                 lr_lambda=None, wrap=None, fused_loss=None, autocast=None, optimizer_cls=None):
        self.optimizer = make_optimizer(student_model, learning_rate, weight_decay, optimizer_cls=optimizer_cls)   # (built before `prepare`, as in the script)
        self.lr_scheduler = torch.optim.lr_scheduler.LambdaLR(self.optimizer, lr_lambda or (lambda step: 1.0))
        self.student_model = wrap(student_model) if wrap is not None else student_model
        self.teacher_model = teacher_model
        self.BaseModelOutput = BaseModelOutput
        self.share_hidden_states, self.teacher_dtype = share_hidden_states, teacher_dtype
        self.kl_weight, self.max_grad_norm = kl_weight, max_grad_norm
        self.fused_loss = fused_loss      # distil_whisper_amd.modeling.fused_distillation_loss (the optional one-call form)
        self.autocast = autocast          # a context-manager factory (HF classes under CPU bf16 autocast in the fixtures)

    def _ctx(self):
        return self.autocast() if self.autocast is not None else torch.autocast("cpu", enabled=False)

    def train_step(self, batch, temperature=2.0):
        """run_distillation.py:1465-1495."""
        student_model, teacher_model = self.student_model, self.teacher_model
        student_model.train()
        teacher_model.eval()

        with self._ctx():
            student_outputs = student_model(**batch)
            with torch.no_grad():
                if self.share_hidden_states:
                    encoder_outputs = self.BaseModelOutput(
                        student_outputs.encoder_last_hidden_state.to(dtype=self.teacher_dtype))
                    teacher_outputs = teacher_model(encoder_outputs=encoder_outputs, labels=batch["labels"])
                else:
                    teacher_outputs = teacher_model(**batch)

        if self.fused_loss is not None:
            return self.fused_loss(student_outputs, teacher_outputs, batch["labels"], temperature, self.kl_weight)
        ce_loss = student_outputs.loss
        teacher_distribution = nn.functional.softmax(teacher_outputs.logits.float() / temperature, dim=-1)
        student_distribution = nn.functional.log_softmax(student_outputs.logits.float() / temperature, dim=-1)
        kl_loss = kl_divergence(teacher_distribution, student_distribution, batch["labels"]) * temperature ** 2

        loss = 0.8 * ce_loss + self.kl_weight * kl_loss
        metrics = {"loss": loss, "ce_loss": ce_loss, "kl_loss": kl_loss}
        return loss, metrics

    def eval_step(self, batch):
        """run_distillation.py:1498-1522 (temperature is always 1 for eval)."""
        student_model, teacher_model = self.student_model, self.teacher_model
        student_model.eval()
        teacher_model.eval()

        with torch.no_grad(), self._ctx():
            student_outputs = student_model(**batch)
            if self.share_hidden_states:
                encoder_outputs = self.BaseModelOutput(
                    student_outputs.encoder_last_hidden_state.to(dtype=self.teacher_dtype))
                teacher_outputs = teacher_model(encoder_outputs=encoder_outputs, labels=batch["labels"])
            else:
                teacher_outputs = teacher_model(**batch)

        ce_loss = student_outputs.loss
        student_distribution = nn.functional.log_softmax(student_outputs.logits.float(), dim=-1)
        teacher_distribution = nn.functional.softmax(teacher_outputs.logits.float(), dim=-1)
        kl_loss = kl_divergence(teacher_distribution, student_distribution, batch["labels"])

        loss = 0.8 * ce_loss + self.kl_weight * kl_loss
        metrics = {"loss": loss, "ce_loss": ce_loss, "kl_loss": kl_loss}
        return metrics

    def training_iteration(self, batch, temperature=2.0):
        """run_distillation.py:1606-1614.  Returns (metrics, gradient norm before clipping)."""
        loss, train_metric = self.train_step(batch, temperature=temperature)
        loss.backward()                                                             # accelerator.backward(loss)
        if hasattr(self.optimizer, "clip_grad_norm_"):          # the drop-in optimizer's own (fused) form of line 1611
            grad_norm = self.optimizer.clip_grad_norm_(self.max_grad_norm)
        else:
            grad_norm = torch.nn.utils.clip_grad_norm_(self.student_model.parameters(), self.max_grad_norm)
        self.optimizer.step()
        self.lr_scheduler.step()
        self.optimizer.zero_grad()
        return {k: v.detach().float() for k, v in train_metric.items()}, grad_norm.detach().float()
