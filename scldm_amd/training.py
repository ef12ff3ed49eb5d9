"""Data-parallel flow-matching training step for the scldm_amd DiT (SURVEY.md section 8a row T1, 8e training row).

The reference trains through Lightning + DDP (experiments/scripts/train_ldm.py; src/scldm/models.py:443-470): per-rank
mini-batches, `Transport.training_losses`, `loss.mean().backward()`, gradient all-reduce, AdamW.  Here the backward is one
C call (scldm_dit_train_backward), so every gradient exists at the same instant and there is nothing to overlap with: the
exchange is ONE all-reduce of a flat fp32 buffer per bucket (the base DiT's 9.8 M parameters = 39 MB fit a single bucket),
which on xGMI's point-to-point links is the efficient shape - few, large collectives.  `torch.nn.parallel.DistributedDataParallel`
also works on the module unchanged (its hooks see ordinary .grad tensors); this helper avoids its per-bucket bookkeeping.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def grad_buckets(params, bucket_bytes: int = 256 << 20):
    """Deterministic partition of the trainable parameters into buckets of at most bucket_bytes (at least one each)."""
    buckets, cur, size = [], [], 0
    for p in params:
        if not p.requires_grad:
            continue
        nbytes = p.numel() * p.element_size()
        if cur and size + nbytes > bucket_bytes:
            buckets.append(cur)
            cur, size = [], 0
        cur.append(p)
        size += nbytes
    if cur:
        buckets.append(cur)
    return buckets


def _shared_flat_grad(params):
    """One 1-D tensor spanning every .grad when all of them are contiguous views of the same storage (what
    scldm_amd.nnets._DiTTrainFn.backward produces), else None."""
    grads = [p.grad for p in params if p.requires_grad]
    if not grads or any(g is None or not g.is_contiguous() for g in grads):
        return None
    g0 = grads[0]
    st = g0.untyped_storage()
    if any(g.untyped_storage().data_ptr() != st.data_ptr() or g.dtype != g0.dtype or g.device != g0.device for g in grads):
        return None
    lo = min(g.storage_offset() for g in grads)
    hi = max(g.storage_offset() + g.numel() for g in grads)
    if sum(g.numel() for g in grads) < 0.9 * (hi - lo):     # mostly gaps: not the flat layout this shortcut is for
        return None
    return torch.empty(0, dtype=g0.dtype, device=g0.device).set_(st, lo, (hi - lo,))


@torch.no_grad()
def allreduce_gradients(params, group=None, bucket_bytes: int = 256 << 20, average: bool = True) -> int:
    """Sum (or average) .grad over the ranks of `group` with one all_reduce per bucket.  A parameter whose .grad is None on
    this rank contributes zeros (and receives the reduced value).  Returns the number of collectives issued."""
    if not dist.is_available() or not dist.is_initialized():
        return 0
    world = dist.get_world_size(group)
    if world == 1:
        return 0
    calls = 0
    params = list(params)
    flat = _shared_flat_grad(params)
    if flat is not None and flat.numel() * flat.element_size() <= bucket_bytes:
        # the HIP backward returns every gradient as a view of ONE buffer: reduce that buffer in place (no gather / scatter
        # copies; the alignment gaps between the views are reduced along with them and never read)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat.div_(world)
        return 1
    for bucket in grad_buckets(params, bucket_bytes):
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat.div_(world)
        off = 0
        for p in bucket:
            n = p.numel()
            g = flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n
        calls += 1
    return calls


def train_step(dit, transport, optimizer, x1: torch.Tensor, condition: dict[str, torch.Tensor], group=None) -> torch.Tensor:
    """One optimisation step on this rank's mini-batch: loss = mean_b training_losses(...)["loss"] (models.py:443-470),
    backward through the HIP kernels, gradient all-reduce (mean over ranks), optimizer.step().  Returns the local loss."""
    optimizer.zero_grad(set_to_none=True)
    loss = transport.training_losses(dit, x1, {"condition": condition})["loss"].mean()
    loss.backward()
    allreduce_gradients(dit.parameters(), group)
    optimizer.step()
    return loss.detach()
