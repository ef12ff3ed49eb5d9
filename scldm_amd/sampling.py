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
def sample_cells(dit, vae, condition, guidance_weight, batch_size: int, genes: torch.Tensor, size_factors: torch.Tensor | None,
                 num_steps: int = 101, sampling_method: str = "euler", z0: torch.Tensor | None = None, draw_counts: bool = True,
                 size_factor_sampler: "SizeFactorSampler | None" = None, seed: int | None = None):
    """Reference `LatentDiffusion.sample` (models.py:766-819): returns (counts or NB distribution, latents) with
    2*batch_size rows, unconditional first.  Log size factors are either given or drawn on device by a
    `SizeFactorSampler` (models.py:785 -> _sample_log_size_factors).  `seed`: the negative-binomial draw's (default: from torch's
    host generator)."""
    if size_factors is None:
        if size_factor_sampler is None:
            raise ValueError("pass size_factors or a size_factor_sampler")
        size_factors = size_factor_sampler.sample(condition, batch_size)
    if len(genes) != batch_size:
        raise ValueError(f"genes batch dimension ({genes.shape[0]}) must match batch_size ({batch_size})")
    dev = dit.pos_embed.device
    if z0 is None:
        z0 = torch.randn((batch_size, dit.seq_len, vae.encoder.latent_embedding), device=dev)
    z = sample_latents(dit, z0, condition, guidance_weight, num_steps, sampling_method)
    genes2 = torch.cat([genes, genes], dim=0)
    lib = torch.exp(size_factors).view(-1, 1)
    lib2 = torch.cat([lib, lib], dim=0)
    if draw_counts:   # models.py:819 `nb.sample()`: fused into the decoder's normalisation pass (mu / theta stay on chip)
        return vae.decode_sample(z, genes2, lib2, seed=seed), z
    return vae.decode(z, genes2, lib2), z


def generate_cells_stream(dit, vae, batches, guidance_weight, genes: torch.Tensor, num_steps: int = 101, sampling_method: str = "euler",
                          size_factor_sampler: "SizeFactorSampler | None" = None, seeds=None, merge_batches: int = 1):
    """The reference's prediction loop (`trainer.predict` -> `predict_step` per batch -> `.cpu()`, src/scldm/models.py:707-764,742)
    as a two-stage pipeline over an iterable of batches: while batch i + 1's CFG ODE runs on the caller's stream, batch i's MCAB
    decode + negative-binomial draw, CSR assembly and pinned device-to-host copies run on a second HIP stream - the ODE kernel is
    matrix-pipe-bound and (below ~700 cells) leaves CU slots empty, the decode is VALU-bound: they share the chip.

    `batches` yields `condition` dicts, or `(condition, log_size_factors)` / `(condition, log_size_factors, z0)` tuples (size
    factors None -> drawn by `size_factor_sampler`, models.py:785); `genes` is the (B, G) gene-index matrix of a batch (every batch
    has B cells).  Yields per batch, in order, `(indptr, indices, data, latents)` as HOST tensors for the 2B generated rows
    (unconditional first): exactly what `sample_cells(..., draw_counts=True)` -> `dense_to_csr` -> `to_host` returns for that batch -
    same kernels, same seeds (`seeds`: optional iterable of draw seeds, one per batch), only the streams differ.
    `merge_batches=k`: k consecutive batches share ONE solve (their noise and size factors are still drawn batch by batch, their rows
    decoded and yielded batch by batch: the same arrays, since a cell's trajectory does not depend on its batch) - throughput for small
    batches (the reference generates with 128 cells per batch, generation.yaml:16: 4 batches per solve are 1.8 x the cells per second) at the
    price of the first result waiting for k batches of input."""
    from .datamodule import dense_to_csr, to_host
    dev = dit.pos_embed.device
    if dev.type != "cuda":
        raise RuntimeError("generate_cells_stream needs the model on a CUDA (ROCm) device; there is no CPU path")
    # (normal priority: a lowest-priority HIP stream starves batch i's decode until batch i + 1's ODE has drained, and the host - blocked in
    # finish(i) - then queues batch i + 2 too late: 20.45 against 19.55 ms per batch, profiles/r6_gen_stream_ab.txt)
    side = torch.cuda.Stream(device=dev)
    genes2 = torch.cat([genes, genes], dim=0)
    seeds = iter(seeds) if seeds is not None else None

    first = [True]

    def finish(item):
        z, sf, ready, seed = item
        with torch.cuda.stream(side):
            side.wait_event(ready)
            lib = torch.exp(sf).view(-1, 1)
            # (after the first batch the VAE's packed weight copies are checked through the version counters only: the device-side
            # fingerprint pass - five small launches per decode - is for `.data` updates, which do not happen inside this loop)
            keep_fp = getattr(vae, "check_weight_fingerprint", None)
            if keep_fp is not None and not first[0]:
                vae.check_weight_fingerprint = False
            try:
                counts = vae.decode_sample(z, genes2, torch.cat([lib, lib], dim=0), seed=seed)
            finally:
                if keep_fp is not None:
                    vae.check_weight_fingerprint = keep_fp
                first[0] = False
            indptr, indices, data = dense_to_csr(counts)          # (its one host wait - nnz sizes the arrays - waits for `side` only)
            return to_host(indptr, indices, data, z)              # z, sf stay referenced by `item` until `side` has drained

    # The CFG plan of a solve must not wait for the host (torch.unique does: behind the PREVIOUS batch's solve, which is what the pipeline wants
    # to run ahead of): dense label-tuple rows, out-of-range labels clamped on the device and raised by check_labels() after the loop.
    # (the flag is set around each solve only - not across `yield`, where it would change the consumer's own calls)
    yield from _stream_loop(dit, vae, batches, guidance_weight, genes, num_steps, sampling_method, size_factor_sampler, seeds, dev, finish,
                            max(1, int(merge_batches)))
    if hasattr(dit, "check_labels"):
        dit.check_labels()


def _stream_loop(dit, vae, batches, guidance_weight, genes, num_steps, sampling_method, size_factor_sampler, seeds, dev, finish, merge=1):
    import contextlib
    from .nnets import weights_unchanged
    B = genes.shape[0]
    pending = []           # items of the previous solve, decoded (and yielded) once the next solve is queued
    group = []             # batches waiting to share one solve
    solved_any = False

    def solve(group):
        # ONE solve over the members' cells: a cell's trajectory does not depend on the batch it is sampled in (tested), so each
        # member's rows are exactly those of its own solve - at 128 cells a launch leaves a quarter of the CUs idle and walks one tile
        # per CU; four members per solve are 1.8 x the cells per second
        conds, sfs, z0s, sds = zip(*group)
        nb = len(group)
        cond = None if conds[0] is None else {k: torch.cat([c[k] for c in conds], dim=0) for k in conds[0]}
        z0 = z0s[0] if nb == 1 else torch.cat(z0s, dim=0)
        had_flag = getattr(dit, "deferred_label_check", None)
        if had_flag is not None:
            dit.deferred_label_check = True
        try:
            with (weights_unchanged() if solved_any else contextlib.nullcontext()):   # (the first solve checked the DiT's packed weights)
                z = sample_latents(dit, z0, cond, guidance_weight, num_steps, sampling_method)      # queued; the host does not wait
        finally:
            if had_flag is not None:
                dit.deferred_label_check = had_flag
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(dev))
        Bt = nb * B
        items = []
        for j in range(nb):
            zj = z if nb == 1 else torch.cat([z[j * B:(j + 1) * B], z[Bt + j * B:Bt + (j + 1) * B]], dim=0)   # unconditional rows first
            items.append((zj, sfs[j], ready, sds[j]))
        if nb > 1:          # (the per-member copies were queued behind the solve on the same stream)
            ready2 = torch.cuda.Event()
            ready2.record(torch.cuda.current_stream(dev))
            items = [(zj, sf, ready2, sd) for zj, sf, _, sd in items]
        return items

    # (no torch.no_grad() around the loop: a context held across `yield` would leak into the consumer's code; nothing here records a graph)
    for batch in batches:
        cond, sf, z0 = (tuple(batch) + (None, None))[:3] if isinstance(batch, (tuple, list)) else (batch, None, None)
        if sf is None:
            if size_factor_sampler is None:
                raise ValueError("pass log size factors with each batch or a size_factor_sampler")
            sf = size_factor_sampler.sample(cond, B)
        if z0 is None:
            z0 = torch.randn((B, dit.seq_len, vae.encoder.latent_embedding), device=dev)
        seed = int(next(seeds)) if seeds is not None else int(torch.randint(0, 2 ** 62, (), dtype=torch.int64).item())
        group.append((cond, sf, z0, seed))
        if len(group) < merge:
            continue
        items = solve(group)
        group, solved_any = [], True
        for it in pending:
            yield finish(it)
        pending = items
    if group:
        items = solve(group)
        for it in pending:
            yield finish(it)
        pending = items
    for it in pending:
        yield finish(it)


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


class SizeFactorSampler:
    """Device-side `LatentDiffusion._sample_log_size_factors` (src/scldm/models.py:473-597).

    The reference walks the batch in Python with one `.item()` host synchronisation and one `Normal(...).sample()` per cell.
    Here the vocabulary encoder's statistics are flattened ONCE into dense device tables (mean, std, valid) - indexed by the
    label for the independent path, by the mixed-radix code of the component labels for the joint path - and a batch is one
    gather plus one `randn`: log_sf = mean[idx] + std[idx] * eps, zero where statistics are missing (the reference's
    fall-back).  Key selection follows the reference: joint statistics when the strategy is "joint" and the joint key is in
    both maps, else `size_factor_condition_key`, else the alphabetically first key common to the condition and both maps."""

    def __init__(self, vocabulary_encoder, condition_strategy: str, device):
        self.device = torch.device(device)
        self.strategy = condition_strategy
        enc = vocabulary_encoder
        self.mu = getattr(enc, "mu_size_factor", None)
        self.sd = getattr(enc, "sd_size_factor", None)
        self.size_factor_condition_key = getattr(enc, "size_factor_condition_key", None)
        self.joint_key = getattr(enc, "joint_key", None)
        self.joint_components = getattr(enc, "joint_components", None)
        self.joint_idx_2_classes = getattr(enc, "joint_idx_2_classes", None)
        self._tables: dict = {}

    # -- table builders (host, once per key) ------------------------------------------------------------------
    def _dense(self, pairs, size):
        tab = torch.zeros((3, size), dtype=torch.float32)
        for idx, (mean, std) in pairs.items():
            if mean is not None and std is not None and 0 <= idx < size:
                tab[0, idx], tab[1, idx], tab[2, idx] = float(mean), float(std), 1.0
        return tab.to(self.device)

    def _independent_table(self, key):
        if ("ind", key) not in self._tables:
            mu, sd = self.mu[key], self.sd[key]
            ids = [int(k) for k in set(mu) | set(sd)]
            size = (max(ids) + 1) if ids else 1
            self._tables[("ind", key)] = self._dense({int(k): (mu.get(k), sd.get(k)) for k in set(mu) | set(sd)}, size)
        return self._tables[("ind", key)]

    def _joint_table(self, keys):
        tkey = ("joint", tuple(keys))
        if tkey not in self._tables:
            parts = [tuple(int(p) for p in k.split("_")) for k in self.joint_idx_2_classes]
            parts = [p for p in parts if len(p) == len(keys)]
            radix = [max((p[d] for p in parts), default=0) + 1 for d in range(len(keys))]
            pairs = {}
            mu, sd = self.mu[self.joint_key], self.sd[self.joint_key]
            for k, cls in self.joint_idx_2_classes.items():
                p = tuple(int(x) for x in k.split("_"))
                if len(p) != len(keys):
                    continue
                code = 0
                for d, v in enumerate(p):
                    code = code * radix[d] + v
                pairs[code] = (mu.get(cls), sd.get(cls))
            size = 1
            for r in radix:
                size *= r
            self._tables[tkey] = (self._dense(pairs, size), radix)
        return self._tables[tkey]

    # -- the call (device, no host synchronisation) -----------------------------------------------------------
    @torch.no_grad()
    def sample(self, condition: dict[str, torch.Tensor] | None, batch_size: int, generator: torch.Generator | None = None,
               eps: torch.Tensor | None = None) -> torch.Tensor:
        zeros = torch.zeros(batch_size, device=self.device)
        if condition is None or self.mu is None or self.sd is None:
            return zeros
        use_joint = (self.strategy == "joint" and self.joint_idx_2_classes is not None and self.joint_key is not None
                     and self.joint_key in self.mu and self.joint_key in self.sd)
        if use_joint:
            keys = [k for k in self.joint_components if k in condition] if self.joint_components is not None else list(condition.keys())
            if any(len(condition[k]) != batch_size for k in keys):
                return zeros
            tab, radix = self._joint_table(keys)
            code = torch.zeros(batch_size, dtype=torch.long, device=self.device)
            inside = torch.ones(batch_size, dtype=torch.bool, device=self.device)
            for d, k in enumerate(keys):
                lab = condition[k].to(device=self.device, dtype=torch.long)
                inside &= (lab >= 0) & (lab < radix[d])
                code = code * radix[d] + lab.clamp(0, radix[d] - 1)
        else:
            sel = self.size_factor_condition_key
            if not (sel and sel in condition and sel in self.mu and sel in self.sd):
                inter = sorted(set(condition.keys()) & set(self.mu.keys()) & set(self.sd.keys()))
                if not inter:
                    return zeros
                sel = inter[0]
            if len(condition[sel]) != batch_size:
                raise ValueError(f"Condition '{sel}' length ({len(condition[sel])}) must match batch size ({batch_size})")
            tab = self._independent_table(sel)
            lab = condition[sel].to(device=self.device, dtype=torch.long)
            inside = (lab >= 0) & (lab < tab.shape[1])
            code = lab.clamp(0, tab.shape[1] - 1)
        if eps is None:
            eps = torch.randn(batch_size, device=self.device, generator=generator)
        valid = inside & (tab[2, code] > 0)
        return torch.where(valid, tab[0, code] + tab[1, code] * eps, zeros)
