"""Generation harness: the counterpart of the reference's `LatentDiffusion.sample` (src/scldm/models.py:766-819)
for callers that hold a scldm_amd DiT + TransformerVAE, plus the batch-sharded multi-GPU variant
(one process per GPU, cells split across ranks, ONE all-gather of the generated latents over RCCL/xGMI).

The reference harness itself keeps working unchanged with scldm_amd modules (it only calls
`forward_with_cfg`, `decode` and `Sampler.sample_ode`); this module is the fused fast path.
"""
from __future__ import annotations

import torch


def shard_bounds(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced split of n cells: the first n % world ranks get one extra cell."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


@torch.no_grad()
def sample_latents(dit, z0: torch.Tensor, condition: dict[str, torch.Tensor] | None, guidance_weight: dict[str, float] | None,
                   num_steps: int = 101, sampling_method: str = "euler") -> torch.Tensor:
    """z0 (B,S,C) noise -> final latents (2B,S,C): rows [0,B) unconditional, rows [B,2B) guided (models.py:801-812)."""
    if guidance_weight is not None and condition is not None:
        assert set(guidance_weight.keys()) == set(condition.keys()), (
            f"Guidance weight keys {set(guidance_weight.keys())} must match condition keys {set(condition.keys())}")
    B = z0.shape[0]
    if condition is not None:
        for k, v in condition.items():
            if len(v) != B:
                raise ValueError(f"Condition '{k}' length ({len(v)}) must match batch size ({B})")
    z2 = torch.cat([z0, z0], dim=0)
    cond2 = None if condition is None else {k: torch.cat([v, v], dim=0) for k, v in condition.items()}
    return dit.sample_ode_cfg(z2, cond2, guidance_weight, num_steps, sampling_method)


@torch.no_grad()
def sample_cells(dit, vae, condition, guidance_weight, batch_size: int, genes: torch.Tensor, size_factors: torch.Tensor,
                 num_steps: int = 101, sampling_method: str = "euler", z0: torch.Tensor | None = None, draw_counts: bool = True):
    """Reference `LatentDiffusion.sample` minus the size-factor draw (models.py:785 is caller-side):
    returns (counts or NB distribution, latents) with 2*batch_size rows, unconditional first."""
    if len(genes) != batch_size:
        raise ValueError(f"genes batch dimension ({genes.shape[0]}) must match batch_size ({batch_size})")
    dev = dit.pos_embed.device
    if z0 is None:
        z0 = torch.randn((batch_size, dit.seq_len, vae.encoder.latent_embedding), device=dev)
    z = sample_latents(dit, z0, condition, guidance_weight, num_steps, sampling_method)
    genes2 = torch.cat([genes, genes], dim=0)
    lib = torch.exp(size_factors).view(-1, 1)
    nb = vae.decode(z, genes2, torch.cat([lib, lib], dim=0))
    return (nb.sample() if draw_counts else nb), z


@torch.no_grad()
def sample_latents_sharded(sample_fn, z0: torch.Tensor, condition: dict[str, torch.Tensor] | None, group=None) -> torch.Tensor:
    """Batch-sharded sampling over the ranks of `group` (default: WORLD).

    Every rank passes the SAME global (z0, condition); rank r integrates only its contiguous shard of cells with
    `sample_fn(z0_shard, condition_shard) -> (2*b_r, S, C)` and the per-rank results are exchanged with one
    all_gather (padded to the largest shard).  Returns the global (2B, S, C) tensor, unconditional rows first,
    identical on every rank and identical to the single-process result (cells are independent).
    """
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    B = z0.shape[0]
    lo, hi = shard_bounds(B, rank, world)
    cond = None if condition is None else {k: v[lo:hi] for k, v in condition.items()}
    local = sample_fn(z0[lo:hi], cond)                      # (2*b, S, C)
    b = hi - lo
    bmax = (B + world - 1) // world
    pad = torch.zeros((2, bmax) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[0, :b] = local[:b]
    pad[1, :b] = local[b:]
    flat = torch.empty((world * 2,) + tuple(pad.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(flat, pad, group=group)    # the only collective on the path (concatenated along dim 0)
    out = flat.view((world, 2) + tuple(pad.shape[1:]))
    unc, gui = [], []
    for r in range(world):
        rlo, rhi = shard_bounds(B, r, world)
        unc.append(out[r, 0, : rhi - rlo])
        gui.append(out[r, 1, : rhi - rlo])
    return torch.cat(unc + gui, dim=0)
