"""Eval-side exchange of generated token ids across ranks (collective C4).

Generation shards by utterance / pack with no data-path collective; the only exchange is the one the reference does
before metrics and before writing pseudo-labels: pad the per-rank id matrices to a common width and concatenate them
in rank order (run_distillation.py:1527 and 1695-1697, run_pseudo_labelling.py:893-895:
`accelerator.pad_across_processes(ids, dim=1, pad_index=pad)` then `accelerator.gather_for_metrics`).  Same semantics
here over `torch.distributed` (backend nccl = RCCL on the GPUs, gloo in the CPU tests): ranks may hold different row
counts (the reference's loader duplicates samples to even the last batch out and `gather_for_metrics` drops the
duplicates again; a contiguous shard of unequal length needs neither).
"""
import torch
import torch.distributed as dist


def _world(group):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def pad_across_processes(tensor, dim=1, pad_index=0, group=None):
    """Pad `tensor` along `dim` with `pad_index` to the largest size any rank holds (accelerate's function of the same
    name with `pad_first=False`).  Returns `tensor` itself when nothing has to change."""
    if _world(group) == 1:
        return tensor
    size = torch.tensor([tensor.shape[dim]], dtype=torch.long, device=tensor.device)
    dist.all_reduce(size, op=dist.ReduceOp.MAX, group=group)
    width = int(size.item())
    if width == tensor.shape[dim]:
        return tensor
    shape = list(tensor.shape)
    shape[dim] = width
    out = tensor.new_full(shape, pad_index)
    out.narrow(dim, 0, tensor.shape[dim]).copy_(tensor)
    return out


def gather_rows(tensor, pad_index=0, group=None):
    """Concatenate `[n_r, L_r]` id matrices of all ranks in rank order into `[sum n_r, max L_r]` (every rank gets the
    whole result).  Rows are padded with `pad_index`; row counts may differ between ranks."""
    world = _world(group)
    if world == 1:
        return tensor
    tensor = pad_across_processes(tensor.contiguous(), dim=1, pad_index=pad_index, group=group)
    n = torch.tensor([tensor.shape[0]], dtype=torch.long, device=tensor.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    rows = max(counts)
    if tensor.shape[0] < rows:
        pad = tensor.new_full((rows - tensor.shape[0],) + tuple(tensor.shape[1:]), pad_index)
        tensor = torch.cat([tensor, pad], dim=0)
    parts = [torch.empty_like(tensor) for _ in range(world)]
    dist.all_gather(parts, tensor, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)


def gather_token_lists(rows, pad_index, device, group=None):
    """Lists of token ids (one per utterance / pack of this rank's shard, ragged) -> the lists of ALL ranks in rank
    order.  `pad_index` must not occur inside a row (it is the tokenizer's pad id in the reference); each row's length
    travels with it, so trailing tokens equal to `pad_index` survive too."""
    width = max((len(r) for r in rows), default=0)
    mat = torch.full((len(rows), width + 1), pad_index, dtype=torch.long)
    for i, r in enumerate(rows):
        mat[i, 0] = len(r)
        if len(r):
            mat[i, 1:1 + len(r)] = torch.as_tensor(r, dtype=torch.long)
    full = gather_rows(mat.to(device), pad_index=pad_index, group=group).cpu()
    return [full[i, 1:1 + int(full[i, 0])].tolist() for i in range(full.shape[0])]
