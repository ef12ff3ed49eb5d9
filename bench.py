#!/usr/bin/env python3
"""bench.py - cells/sec of CFG flow-matching sampling with the MI355X DiT path (one JSON line on rank 0).

step     = one full sampling pass of the per-GPU batch: noise (already resident in HBM) -> final guided latents,
           n_evals DiT-with-CFG evaluations (3 sample-forwards per requested cell per evaluation), fused in one
           C call (scldm_sample_ode); for N > 1 followed by the single RCCL all-gather of the generated latents.
value    = whole-job requested cells per second = N * B_per_gpu / (time per step), max over ranks.
roofline = the fused DiT block kernel (dominant): algorithmic FLOPs per launch / mean launch duration measured
           with HIP events on the launch stream inside the timed region, against the dense MFMA peak.
parity_path = the same workload at precision "bf16x3" (split-bf16, <= 1e-4 vs the reference): cells/s, roofline against
           its own peak (three bf16 MFMAs per product sum: 2.5 PF / 3), error measured against the exact-fp32 path.
reference_class_path = the same at precision "fp16" (10 mantissa bits = the TF32 arithmetic the reference itself computes in).
cpu_baseline = the CPU oracle ("port" of the reference algorithm) on this box's host cores, bounded sample.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches the N ranks itself (one process per
GPU through torch.distributed.run, before this process touches a GPU); under an external launcher (the driver's
`python -m torch.distributed.run ... bench.py --gpus N`) it runs as the rank it was started as.
"""
from __future__ import annotations

import argparse
import json
import os
import platform
import socket
import statistics
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # north_star target row: dentate_gyrus shape, batch 4096 x 100 Euler evaluations on one MI355X, bf16
    "dentate_b4096_euler100": dict(vocab={"clusters": 14}, strategy="mutually_exclusive", B=4096, evals=100, method="euler", scale=1.0),
    # BASELINE.json configs[1]
    "dentate_b512_euler50": dict(vocab={"clusters": 14}, strategy="mutually_exclusive", B=512, evals=50, method="euler", scale=1.0),
    # a small interactive batch (384 sample-forwards per evaluation = 192 32-token tiles: the launch runs on the 32-token-tile kernel)
    "dentate_b128_euler50": dict(vocab={"clusters": 14}, strategy="mutually_exclusive", B=128, evals=50, method="euler", scale=1.0),
    # configs[2]: 100 Heun steps = 200 evaluations, guidance 2.0
    "hlca_b2048_heun100": dict(vocab={"cell_type": 50}, strategy="mutually_exclusive", B=2048, evals=200, method="heun", scale=2.0),
    # configs[3] per-GPU shard: 8192 cells over 8 GPUs = 1024 per GPU, joint conditioning
    "parse1m_b1024_euler100": dict(vocab={"cell_type": 18, "cytokine": 91}, strategy="joint", B=1024, evals=100, method="euler", scale=1.0),
    # configs[3] as north_star states it: 8192 cells GLOBAL, sharded over the ranks of the job (strong scaling: B = 8192 / N per GPU)
    "parse1m_b8192_euler100_strong": dict(vocab={"cell_type": 18, "cytokine": 91}, strategy="joint", B=8192, evals=100, method="euler",
                                          scale=1.0, strong=True),
}
TRAIN_WORKLOADS = {
    # BASELINE.json configs[4] restated on the reference's own DiT shape (ldm_base.yaml; "DiT-L" is not a reference config):
    # replogle labels (cell_line 4, gene 2024, joint), cfg_dropout_prob 0.8, per-GPU batch 1024, AdamW, data-parallel
    "replogle_train_b1024": dict(vocab={"cell_line": 4, "gene": 2024}, strategy="joint", B=1024),
    # the same step on the DiT-L shape configs[4] names (1024 wide, 24 layers, 16 heads; not a reference config):
    # outside the fused family, so it runs on the generic GEMM-based HIP path
    "replogle_train_ditl_b256": dict(vocab={"cell_line": 4, "gene": 2024}, strategy="joint", B=256,
                                     shape=dict(n_embed=1024, n_layer=24, n_head=16)),
    # same, 1 024 cells per GPU (29 GB of saved activations): the batch at which the 256 x 256-tile GEMM fills the chip
    "replogle_train_ditl_b1024": dict(vocab={"cell_line": 4, "gene": 2024}, strategy="joint", B=1024,
                                      shape=dict(n_embed=1024, n_layer=24, n_head=16)),
}


def dit_flops(n_embed=256, n_layer=8, n_embed_input=16, seq_len=16, multiple_of=4):
    """Algorithmic FLOPs (2 x MAC, GEMMs + attention contractions) of one DiT sample-forward (BASELINE.md section 3 formula)."""
    D, S = n_embed, seq_len
    H = multiple_of * ((int(2 * (4 * D) / 3) + multiple_of - 1) // multiple_of)
    block = 2 * D * 6 * D + S * (2 * D * 3 * D + 2 * D * D + 6 * D * H) + 2 * (2 * S * S * D)
    return n_layer * block + 2 * 256 * D + 2 * D * D + S * 2 * n_embed_input * D + 2 * D * 2 * D + S * 2 * D * n_embed_input
FLOPS_PER_SAMPLE_FWD = 210_763_776            # BASELINE.md section 3 (adaLN + timestep MLP counted once per sample-forward)
FLOPS_PER_SAMPLE_BLOCK = 26_247_168 - 2 * 256 * 1536  # fused block kernel: everything of a block except the adaLN projection
# What the path EXECUTES: the trunk per sample-forward, the adaLN projections once per conditioning ROW (1 unconditional row +
# one per unique label tuple per conditional pass), the timestep MLP once per evaluation.
FLOPS_TRUNK_PER_SAMPLE_FWD = 8 * FLOPS_PER_SAMPLE_BLOCK + 16 * 2 * 16 * 256 + 16 * 2 * 256 * 16
FLOPS_ADALN_PER_ROW = 2 * 256 * (8 * 1536 + 512)
FLOPS_TMLP = 2 * 256 * 256 * 2
# dense MFMA peaks, MI355X_MICROARCH.md:41-42; bf16x3 issues three bf16 MFMAs per algorithmic product sum
PEAK = {"bf16": 2.5e15, "fp16": 2.5e15, "fp32": 157.3e12, "bf16x3": 2.5e15 / 3}


# --------------------------------------------------------------------------------------------------------------------
# launcher
# --------------------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int, argv: list[str], script: str = os.path.abspath(__file__)) -> int:
    """Start `n` ranks of this script on one node (one process per GPU) and return the job's exit code.  Called before the
    parent has made any GPU call: the ranks are CHILD processes (never an exec of a process that initialised the GPU)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this platform
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), script, *argv]
    return subprocess.run(cmd, env=env).returncode


# --------------------------------------------------------------------------------------------------------------------
# models and inputs
# --------------------------------------------------------------------------------------------------------------------
class FakeSampler(torch.nn.Module):
    """Per-cell, order-preserving stand-in for the HIP sampler (BENCH_FAKE=1: the CPU / gloo test of the launcher, timing
    harness and all-gather; tests/test_bench_launcher.py).  Never used for a reported number."""
    precision = "fake"

    def sample_ode_cfg(self, z2, cond2, scales, num_steps, method):
        lab = sum(v.float() for v in cond2.values()).view(-1, 1, 1)
        bad = os.environ.get("BENCH_FAKE_CORRUPT_RANK")     # test hook: this rank computes something else (the self-check must catch it)
        return z2 * 2.0 + lab + (1e-3 if bad is not None and bad == os.environ.get("RANK") else 0.0)

    def block_timing(self, enable=None):
        return None if enable is not None else (0, 0.0)

    def layers_per_launch(self):
        return 0


def make_model(wl, precision, device, seed=0):
    if os.environ.get("BENCH_FAKE") == "1":
        return FakeSampler()
    from scldm_amd.nnets import DiT
    shape = dict(n_embed=256, n_layer=8, n_head=8)
    shape.update(wl.get("shape", {}))
    m = DiT(n_embed_input=16, seq_len=16, dropout=0.0, bias=True, norm_layer="layernorm",
            multiple_of=4, layernorm_eps=1e-8, class_vocab_sizes=wl["vocab"], cfg_dropout_prob=0.8,
            condition_strategy=wl["strategy"], **shape)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in m.parameters():  # random-init weights of the reference architecture (no checkpoints offline);
            p.copy_(torch.randn(p.shape, generator=g) * 0.05)  # adaLN-Zero init would make the network output zeros
    m = m.to(device).eval()
    m.precision = precision
    return m


def make_inputs(wl, B, device, seed):
    g = torch.Generator().manual_seed(seed)
    z0 = torch.randn(B, 16, 16, generator=g)
    labels = {k: torch.randint(0, v, (B,), generator=g) for k, v in wl["vocab"].items()}
    z2 = torch.cat([z0, z0]).to(device)
    cond2 = {k: torch.cat([v, v]).to(device) for k, v in labels.items()}
    scales = {k: wl["scale"] for k in wl["vocab"]}
    return z2, cond2, scales


def n_passes(wl):
    return 1 if wl["strategy"] == "joint" else len(wl["vocab"])


def unique_label_rows(cond2, B):
    cols = torch.stack([v[B:] for _, v in sorted(cond2.items())], dim=1)
    return int(torch.unique(cols, dim=0).shape[0])


def executed_flops_per_eval(wl, B, cond2):
    """FLOPs one CFG evaluation executes on this path (see FLOPS_TRUNK_PER_SAMPLE_FWD)."""
    P = n_passes(wl)
    rows = 1 + P * unique_label_rows(cond2, B)
    return (2 + P) * B * FLOPS_TRUNK_PER_SAMPLE_FWD + rows * FLOPS_ADALN_PER_ROW + FLOPS_TMLP


def sync(device):
    if device.type == "cuda":
        torch.cuda.synchronize()


def run_steps(m, wl, z2, cond2, scales, steps, dist_on, world, bmax=None):
    """`steps` sampling passes; for N > 1 each followed by the path's single collective: an all-gather of the generated latents
    (1 KB per cell).  Ranks may hold different numbers of cells (a strong-scaling split of a global batch that N does not divide):
    every rank then contributes [unconditional | guided] blocks padded to `bmax` cells, as scldm_amd.sampling.sample_latents_sharded does."""
    out = None
    for _ in range(steps):
        out = m.sample_ode_cfg(z2, cond2, scales, wl["evals"] + 1 if wl["method"] == "euler" else wl["evals"] // 2 + 1, wl["method"])
        if dist_on:
            import torch.distributed as dist
            b = out.shape[0] // 2
            if bmax is not None and bmax != b:
                pad = torch.zeros((2, bmax) + tuple(out.shape[1:]), device=out.device, dtype=out.dtype)
                pad[0, :b] = out[:b]
                pad[1, :b] = out[b:]
                out = pad.view((2 * bmax,) + tuple(out.shape[1:]))
            gathered = torch.empty((world * out.shape[0],) + tuple(out.shape[1:]), device=out.device, dtype=out.dtype)
            dist.all_gather_into_tensor(gathered, out)  # the single collective of the path: generated latents, 1 KB per cell
            out = gathered
    return out


def shard_cells(wl, world, rank):
    """(cells of this rank, cells of the largest rank): a weak-scaling workload gives every rank wl['B'] cells; a strong-scaling one
    splits wl['B'] GLOBAL cells like scldm_amd.sampling.shard_bounds (the first B % world ranks get one extra cell)."""
    if not wl.get("strong"):
        return wl["B"], wl["B"]
    base, rem = divmod(wl["B"], world)
    return base + (1 if rank < rem else 0), base + (1 if rem else 0)


def time_workload(m, wl, device, steps, warmup, dist_on, world, rank, time_blocks):
    """`steps` timed steps bracketed by barrier + synchronize on both sides; returns (seconds: max over ranks, block timing,
    last output, inputs).  A strong-scaling workload gives every rank B / world of its cells."""
    B, bmax = shard_cells(wl, world, rank)
    z2, cond2, scales = make_inputs(wl, B, device, seed=1234 + rank)
    run_steps(m, wl, z2, cond2, scales, warmup, dist_on, world, bmax)
    if time_blocks:
        m.block_timing(True)
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
    sync(device)
    t0 = time.perf_counter()
    out = run_steps(m, wl, z2, cond2, scales, steps, dist_on, world, bmax)
    sync(device)
    if dist_on:
        dist.barrier()
    dt = time.perf_counter() - t0
    blocks = m.block_timing() if time_blocks else None
    if time_blocks:
        m.block_timing(False)
    if dist_on:
        mine = torch.tensor([dt], device=device, dtype=torch.float64)
        every = torch.empty(world, device=device, dtype=torch.float64)
        dist.all_gather_into_tensor(every, mine)
        time_workload.per_rank_s = [float(v) for v in every.tolist()]
        dt = max(time_workload.per_rank_s)
    return dt, blocks, out, (z2, cond2, scales, B)


def time_steps_each(m, wl, device, steps, warmup, seed=1234):
    """Seconds of each of `steps` single-GPU steps (synchronised individually) after `warmup` untimed ones; returns (times, inputs)."""
    B = wl["B"]
    z2, cond2, scales = make_inputs(wl, B, device, seed=seed)
    run_steps(m, wl, z2, cond2, scales, warmup, False, 1)
    sync(device)
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        run_steps(m, wl, z2, cond2, scales, 1, False, 1)
        sync(device)
        ts.append(time.perf_counter() - t0)
    return ts, (z2, cond2, scales, B)


def cross_rank_check(m, wl, out, B, device, world, rank, n_check=8):
    """Self-validation of the sharded run (no curve can be measured on a 1-GPU lease: this makes the first N-GPU run its own test):
    rank 0 regenerates `n_check` cells of ANOTHER rank's shard from that rank's seed, integrates them locally with its own copy of
    the weights and requires BIT equality with the rows that came back through the all-gather - both CFG halves.  Proves the
    gathered tensor is ordered [rank][2B rows], that every rank ran the same model on its own inputs, and that a cell's result
    does not depend on its batch (the property the sharding relies on).  Returns a record for the JSON line."""
    if world < 2 or out is None:
        return None
    rec = {"checked_rank": None, "cells": 0, "bit_equal": None}
    if rank == 0:
        r = world - 1 if world > 1 else 0
        B, bmax = shard_cells(wl, world, r)          # the checked rank's shard (it may be one cell shorter than rank 0's)
        n = min(n_check, B)
        z2, cond2, scales = make_inputs(wl, B, device, seed=1234 + r)
        idx = torch.cat([torch.arange(n), torch.arange(B, B + n)]).to(device)
        zs, cs = z2[idx], {k: v[idx] for k, v in cond2.items()}
        steps = wl["evals"] + 1 if wl["method"] == "euler" else wl["evals"] // 2 + 1
        local = m.sample_ode_cfg(zs, cs, scales, steps, wl["method"])
        block = out[r * 2 * bmax:(r + 1) * 2 * bmax]
        got = torch.cat([block[:n], block[bmax:bmax + n]])
        rec = {"checked_rank": r, "cells": n, "bit_equal": bool(torch.equal(local, got)),
               "max_abs_diff": float((local - got).abs().max())}
        if not rec["bit_equal"]:
            raise SystemExit(f"bench.py: cross-rank self-check FAILED: rank {r}'s first {n} cells, recomputed on rank 0, differ from the "
                             f"gathered rows (max |diff| {rec['max_abs_diff']:.3e})")
    return rec


def time_allgather(out, B, device, world, reps=5):
    """The path's single collective on its own: mean seconds of an all-gather of one rank's (2B, 16, 16) latents."""
    import torch.distributed as dist
    mine = out[:2 * B].contiguous()
    gathered = torch.empty((world * mine.shape[0],) + tuple(mine.shape[1:]), device=device, dtype=mine.dtype)
    dist.all_gather_into_tensor(gathered, mine)
    sync(device)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        dist.all_gather_into_tensor(gathered, mine)
    sync(device)
    return (time.perf_counter() - t0) / reps


def make_optimizer(params, lr, kind="native", max_grad_norm=None):
    """AdamW as the reference's trainer configures it (src/scldm/models.py configure_optimizers): `native` = scldm_amd.optim.AdamW (the
    same arithmetic in one HIP launch per step), `torch` = torch.optim.AdamW(fused=True)."""
    if kind == "torch":
        return torch.optim.AdamW(params, lr=lr, fused=True)
    from scldm_amd.optim import AdamW
    return AdamW(params, lr=lr, max_grad_norm=max_grad_norm)


def time_training(wl, precision, device, steps, warmup, dist_on, world, optimizer="native", graphed=False):
    """One step = Transport.training_losses forward + HIP backward + gradient all-reduce (N > 1) + fused AdamW on the per-GPU
    batch of synthetic latents (standing in for frozen-VAE output).  Returns seconds for `steps` steps (max over ranks)."""
    from scldm_amd.training import train_step
    from scldm_amd.transport import create_transport
    import gc
    gc.collect()              # a previous record's model must not be finalised (hipFree = device-wide synchronisations) inside this timed loop
    torch.cuda.synchronize()
    m = make_model(wl, precision, device).train()
    opt = make_optimizer(m.parameters(), 1e-4, optimizer)
    tr = create_transport("Linear", "velocity", "velocity", 1e-5, 1e-5)
    g = torch.Generator().manual_seed(3)
    x1 = torch.randn(wl["B"], 16, 16, generator=g).to(device)
    cond = {k: torch.randint(0, v, (wl["B"],), generator=g).to(device) for k, v in wl["vocab"].items()}
    step = lambda: train_step(m, tr, opt, x1, cond)
    time_training.graphed = False
    if graphed in ("fused", "fused_eager") and not dist_on:
        # the product's one-call step (scldm_amd.training.FusedTrainStep -> scldm_dit_train_step, replayed as a HIP graph): batch
        # preparation, forward, loss, backward, AdamW + EMA; the latents are copied into the graph's static input every step
        from scldm_amd.ema import EMA
        from scldm_amd.training import FusedTrainStep
        ema = EMA(model=m, beta=0.9999, update_every=10, update_after_step=10_000)      # ldm_base.yaml:51-55
        fstep = FusedTrainStep(m, tr, opt, wl["B"], list(wl["vocab"]), ema=ema, seed=7, graph=graphed == "fused")
        def step():
            loss = fstep(x1, cond)
            ema.update()
            return loss
        time_training.graphed = True
        graphed = False
    if graphed and not dist_on:
        # the product's whole-step HIP graph (scldm_amd.training.GraphedTrainStep): forward, backward and optimizer captured once,
        # one graph launch per mini-batch (the batch is copied into the graph's static inputs every step, as a data loader's would be)
        try:
            from scldm_amd.training import GraphedTrainStep
            gstep = GraphedTrainStep(m, tr, opt, x1, cond)
            step = lambda: gstep(x1, cond)
            time_training.graphed = True
        except Exception as e:   # a capture failure must not take the bench line down: the eager step is timed instead, and said so
            note(f"GraphedTrainStep failed ({e!r}): timing the eager train_step")
            torch.cuda.synchronize()
    for _ in range(warmup):
        step()
    # everything built so far (modules, ctypes tables, torch's own objects) moves to the permanent generation: a full collection of
    # this heap takes ~70 ms (measured: one lands on the 40th eager step and reads as +3.5 ms per step in a 20-step window) - the usual
    # practice of a training loop (gc.freeze after set-up); the per-step garbage itself is still collected
    gc.collect()
    gc.freeze()
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sync_obj = m.__dict__.get("_train_step_sync")
    time_training.info = {"gradient_collectives_per_step": sync_obj.collectives if sync_obj is not None else 0,
                          "gradient_buckets": len(m.grad_bucket_plan()) if hasattr(m, "grad_bucket_plan") else None,
                          "overlapped_with_backward": bool(sync_obj is not None and sync_obj.handled)}
    if dist_on:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # every rank must hold the same parameters after the same number of averaged steps
        chk = torch.stack([p.detach().double().sum() for p in m.parameters()]).sum().reshape(1)
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        time_training.info["replicas_in_sync"] = bool(float(hi - lo) <= 1e-6 * max(1.0, abs(float(hi))))
        if not time_training.info["replicas_in_sync"]:
            raise SystemExit("bench.py: data-parallel replicas diverged (parameter checksums differ across ranks)")
    return dt, float(loss)


def train_end_to_end(wl, precision, device, n_genes=17002, S=6147, steps=20):
    """The LDM training step as the reference runs it (src/scldm/models.py:628-663 + hooks): tokenised counts of the batch -> frozen
    VAE encode (models.py:641; fp16 operands = the reference's TF32 class) -> flow-matching step -> AdamW -> EMA hook (models.py:83-87,
    ldm_base.yaml:51-55), everything inside ONE HIP graph replay per step.  ms per step (wall, 20 replays between synchronisations) and the
    device time of every stage of one eager step (HIP events between the sub-calls)."""
    from scldm_amd.ema import EMA
    from scldm_amd.training import FusedTrainStep
    from scldm_amd.transport import create_transport
    import gc
    gc.collect()
    torch.cuda.synchronize()
    B = wl["B"]
    m = make_model(wl, precision, device).train()
    vae = make_vae(n_genes, device)
    for p in vae.parameters():
        p.requires_grad_(False)
    vae.precision = "fp16"
    opt = make_optimizer([p for p in m.parameters() if p.requires_grad], 1e-4, "native")
    tr = create_transport("Linear", "velocity", "velocity", 1e-5, 1e-5)
    ema = EMA(model=m, beta=0.9999, update_every=10, update_after_step=10_000)
    g = torch.Generator().manual_seed(5)
    genes = torch.stack([torch.randperm(n_genes, generator=g)[:S] for _ in range(8)]).repeat(B // 8, 1).to(device)
    counts = torch.poisson(torch.full((B, S), 1.5), generator=g).to(device)
    cond = {k: torch.randint(0, v, (B,), generator=g).to(device) for k, v in wl["vocab"].items()}
    rec = {"workload": f"replogle_train_b{B} from tokenised counts", "cells": B, "tokens_per_cell": S, "n_genes": n_genes, "dtype": precision,
           "vae_encode_precision": "fp16", "ema": "beta 0.9999, every 10 steps, after 10 000 (ldm_base.yaml:51-55)",
           "gradient_clip": "global norm 10.0 inside the optimizer launch (training/default.yaml:15-16: gradient_clip_val 10, algorithm norm)"}
    for graph in (True, False):
        fs = FusedTrainStep(m, tr, opt, B, list(wl["vocab"]), ema=ema, vae=vae, seed=11, graph=graph, encode_shape=(B, S), grad_clip_norm=10.0)
        def step():
            loss = fs(condition=cond, counts_subset=counts, genes_subset=genes)
            ema.update()
            return loss
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        rec["ms_per_step_graph" if graph else "ms_per_step_eager_one_c_call"] = 1e3 * (time.perf_counter() - t0) / steps
        if not graph:
            st = [fs.profile_stages() for _ in range(5)]
            for _ in range(5):
                ema.update()
            rec["stage_ms"] = {k: statistics.median(s_[k] for s_ in st) for k in st[0]}
        del fs
    rec["ms_per_step"] = rec["ms_per_step_graph"]
    rec["cells_per_s"] = B / (rec["ms_per_step"] / 1e3)
    rec["last_grad_norm"] = float(opt.last_grad_norm)
    return rec


def fused_kernel_roofline(m, blocks, n_fwd, precision, wl, evals_per_step):
    """FLOPs per launch / mean launch duration (HIP events on the launch stream) for the fused DiT layer kernel."""
    n_launch, tot_ms = blocks
    avg_s = tot_ms / n_launch / 1e3
    lpl = m.layers_per_launch()
    flops_launch = n_fwd * FLOPS_PER_SAMPLE_BLOCK * lpl
    ach = flops_launch / avg_s / 1e12
    # algorithmic HBM bytes per launch: the latents in / velocities out once per evaluation, the fp32 residual handed from launch
    # to launch (written by every launch but the last, read by every launch but the first), every layer's packed weights once
    n_l = 8
    launches_per_eval = (n_l + lpl - 1) // lpl
    wbytes = {"bf16": 2, "fp16": 2, "bf16x3": 4, "fp32": 4}[precision] * (768 * 256 + 256 * 256 + 3 * 256 * 704)
    per_eval = 2 * n_fwd * 16 * 16 * 4 + (launches_per_eval - 1) * 2 * n_fwd * 16 * 256 * 4 + n_l * wbytes
    return {"bound": "mfma", "kernel": "dit_forward_kernel", "achieved": ach, "peak": PEAK[precision] / 1e12, "unit": "TFLOP/s",
            "frac": ach / (PEAK[precision] / 1e12), "launches": n_launch, "avg_launch_us": avg_s * 1e6, "layers_per_launch": lpl,
            "algorithmic_flops_per_launch": flops_launch, "algorithmic_hbm_bytes_per_launch": per_eval / launches_per_eval}


def make_vae(n_genes, device):
    from scldm_amd.layers import InputTransformerVAE
    from scldm_amd.nnets import Decoder, Encoder
    from scldm_amd.stochastic_layers import NegativeBinomialTransformerLayer
    from scldm_amd.vae import TransformerVAE
    kw = dict(n_embed=32, n_embed_latent=16, n_head=8, n_head_cross=4, n_layer=8, dropout=0.0, bias=False, multiple_of=4,
              layernorm_eps=1e-8, norm_layer="layernorm")
    vae = TransformerVAE(Encoder(n_inducing_points=16, positional_encoding=True, **kw),
                         Decoder(n_genes=n_genes, n_inducing_points=16, shared_embedding=True, use_adaln=False, **kw),
                         NegativeBinomialTransformerLayer(n_genes=n_genes, shared_theta=True, n_embed=32),
                         InputTransformerVAE(n_genes=n_genes, n_embed=32, agg_func="log1p"))
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for n, p in vae.named_parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (1.0 if "embedding" in n or "inducing" in n else 0.05) + (1.0 if ".ln_" in n and n.endswith("weight") else 0.0))
    return vae.to(device).eval()


MCAB_DECODE_FLOPS_PER_GENE = 23_104   # BASELINE.md section 3 (per decoded gene; the 16-token trunk adds 3.3 MFLOP per cell)


def throughput_and_latency(fn, calls=20, repeats=3):
    """(seconds per call with `calls` calls enqueued back to back between two synchronisations - the headline's protocol: a
    throughput; median of `repeats`), (seconds of ONE synchronised call - a latency: launch ramp, host enqueue and drain included)"""
    fn(); torch.cuda.synchronize()
    thr = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        for _ in range(calls):
            fn()
        torch.cuda.synchronize()
        thr.append((time.perf_counter() - t0) / calls)
    lat = []
    for _ in range(5):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); lat.append(time.perf_counter() - t0)
    return statistics.median(thr), statistics.median(lat)


def decode_inclusive(m, wl, device, n_genes=17002):
    """Same sampling pass followed by the MCAB decode of all 2B latents to NB parameters (dentate_gyrus gene count), and the
    decode alone at every VAE precision with its roofline (fp32-MFMA peak for the parity path, the 16-bit peak for fp16 / bf16).
    The chain's decode runs in the reference's arithmetic class (fp16 operands = TF32's mantissa, experiments/scripts/inference.py:26)
    when the DiT runs a 16-bit policy, exact fp32 otherwise."""
    vae = make_vae(n_genes, device)
    vae.precision = chain_prec = "fp16" if m.precision in ("bf16", "fp16") else "fp32"
    B = min(wl["B"], 4096)  # the headline's own batch: (2B, G) fp32 mu + theta outputs are 2 x 557 MB at B = 4 096 (round 5 capped this at 1 024 cells)
    w2 = dict(wl); w2["B"] = B
    z2, cond2, scales = make_inputs(w2, B, device, seed=7)
    genes = torch.arange(n_genes, device=device).repeat(2 * B, 1)
    lib = torch.full((2 * B, 1), 3000.0, device=device)
    steps = w2["evals"] + 1 if w2["method"] == "euler" else w2["evals"] // 2 + 1
    def once():
        z = m.sample_ode_cfg(z2, cond2, scales, steps, w2["method"])
        return vae.decode(z, genes, lib)
    once(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); once(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    dt = statistics.median(ts)
    rec = {"cells_per_gpu": B, "n_genes": n_genes, "cells_per_s": B / dt, "decode_precision": chain_prec,
           "note": "sampling + MCAB decode of the 2B latents to (mu, theta); decode rates count decoded rows"}
    flops_row = n_genes * MCAB_DECODE_FLOPS_PER_GENE + 3.3e6
    for prec in ("fp32", "fp16", "bf16"):
        vae.precision = prec
        dd, lat = throughput_and_latency(lambda: vae.decode(z2, genes, lib), calls=8)
        ach = 2 * B * flops_row / dd / 1e12
        rec[f"decode_only_{prec}"] = {"rows_per_s": 2 * B / dd, "ms": 1e3 * dd, "ms_single_synchronised_call": 1e3 * lat,
                                      "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK[prec] / 1e12, "unit": "TFLOP/s",
                                                   "frac": ach / (PEAK[prec] / 1e12),
                                                   "hbm_out_GBps": 2 * B * n_genes * 8 / dd / 1e9}}
    rec["decode_only_cells_per_s"] = rec[f"decode_only_{chain_prec}"]["rows_per_s"]
    return rec


def mcab_roofline(device, rows=8192, cells=4096, n_genes=17002, S=6147, calls=6):
    """`roofline`-style blocks for the MCAB kernels at the shapes profiles/r6_mcab_* were taken at (decode 8 192 rows x 17 002
    genes, encode 4 096 cells x 6 147 tokens: the dentate_gyrus sizes at the north-star batch), per precision policy: algorithmic
    FLOPs per launch / mean launch duration from HIP events around each kernel on its launch stream (scldm_vae_kernel_timing), the
    same figure `rocprofv3 --kernel-trace --stats` of tests/perf/mcab_profile.py gives; `traffic` = HBM bytes per launch from the
    committed PMC passes (profiles/pmc_mcab.json: FETCH_SIZE doubled per the guide + WRITE_SIZE, KB -> bytes).
    Algorithmic work: dec_gene_kernel 23 104 FLOP per decoded gene; enc_pool_kernel 6 528 FLOP per pooled token; the
    16-token trunks 3.3 / 1.5 MFLOP per cell.  Algorithmic HBM bytes: decode = logits out 4 B + theta out 4 B + gene id in 8 B per
    gene (tables are L2-resident), encode = count 4 B + gene id 8 B per token."""
    vae = make_vae(n_genes, device)
    g = torch.Generator().manual_seed(11)
    z = torch.randn(rows, 16, 16, generator=g).to(device)
    genes = torch.arange(n_genes, device=device).repeat(rows, 1)
    lib = torch.full((rows, 1), 3000.0, device=device)
    sgenes = torch.stack([torch.randperm(n_genes, generator=g)[:S] for _ in range(8)]).repeat(cells // 8, 1).to(device)
    counts = torch.poisson(torch.full((cells, S), 1.5), generator=g).to(device)
    pmc = {}
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_mcab.json")) as f:
            pmc = json.load(f)
    except Exception:
        pass
    work = {  # kernel -> (algorithmic FLOPs per launch, algorithmic HBM bytes per launch, bound)
        "dec_gene": (rows * n_genes * MCAB_DECODE_FLOPS_PER_GENE, rows * n_genes * 16.0),
        "dec_cell": (rows * 3.3e6, rows * (16 * 16 * 4 + 48 * 64 * 4.0)),
        "dec_finalize": (rows * n_genes * 4.0, rows * n_genes * 8.0),
        "enc_pool": (cells * S * MCAB_ENCODE_FLOPS_PER_GENE, cells * S * 12.0),
        "enc_cell": (cells * 1.5e6, cells * (16 * 32 * 4 + 16 * 16 * 4.0)),
    }
    rec = {"decode_rows": rows, "encode_cells": cells, "n_genes": n_genes, "tokens_per_cell": S}
    for prec in ("fp32", "fp16", "bf16"):
        vae.precision = prec
        vae.decode(z, genes, lib); vae.encode(counts, sgenes); torch.cuda.synchronize()
        vae.kernel_timing(True)
        for _ in range(calls):
            vae.decode(z, genes, lib)
            vae.encode(counts, sgenes)
        t = vae.kernel_timing()
        vae.kernel_timing(False)
        out = {}
        for k, (n, ms) in t.items():
            if not n:
                continue
            fl, by = work[k]
            us = 1e3 * ms / n
            hbm_bound = k == "dec_finalize"
            peak = 8000.0 if hbm_bound else PEAK[prec] / 1e12   # (the trunks' Linears take the policy's operand type too)
            ach = by / (us * 1e-6) / 1e9 if hbm_bound else fl / (us * 1e-6) / 1e12
            traffic = (pmc.get(prec, {}).get(k, {}) or {}).get("hbm_bytes_per_launch")
            out[k] = {"bound": "hbm" if hbm_bound else "mfma", "achieved": ach, "peak": peak, "unit": "GB/s" if hbm_bound else "TFLOP/s",
                      "frac": ach / peak, "launches": n, "avg_launch_us": us, "algorithmic_flops_per_launch": fl,
                      "algorithmic_hbm_bytes_per_launch": by, "traffic": traffic,
                      "traffic_ratio": (traffic / by) if traffic else None}
        rec[prec] = out
    rec["dominant_kernel"] = "dec_gene_kernel"
    rec["traffic_source"] = "profiles/pmc_mcab.json" if pmc else None
    return rec


ARITH = {
    "bf16": "bf16 operands (8 mantissa bits: narrower than the reference's TF32), v_mfma_f32_32x32x16_bf16, fp32 accumulate / LN / softmax / residual",
    "bf16x3": "operands split into hi + lo bf16, 3 x v_mfma_f32_32x32x16_bf16 per k-step, fp32 accumulate / LN / softmax / residual",
    "fp16": "fp16 operands (10 mantissa bits = TF32, the reference's set_float32_matmul_precision('high') class), v_mfma_f32_32x32x16_f16, "
            "fp32 accumulate / LN / softmax / residual",
}


MCAB_ENCODE_FLOPS_PER_GENE = 6_528    # 41.6 MFLOP per cell at S = 6 147 incl. the 16-token trunk (1.5 MFLOP)


def encode_record(device, batches=(1024, 4096)):
    """MCAB pooling + encoder trunk (TransformerVAE.encode) on synthetic counts, both VAE precisions, with a roofline against the
    fp32-MFMA / bf16-MFMA peak: the dentate_gyrus size (G = 17 002 vocabulary, S = 6 147 tokens per cell) and the hlca size
    (27 997 / 10 186).  Algorithmic FLOPs per cell: S x 6 528 + 1.5e6.  Keys without a batch suffix are 1 024 cells (the
    per-GPU shard of configs[3]); `_b4096` is the north-star batch (the 16-token trunk is a fixed ~0.12 ms per call up to ~4 096 cells)."""
    rec = {}
    for name, n_genes, S in (("dentate", 17002, 6147), ("hlca", 27997, 10186)):
        vae = make_vae(n_genes, device)
        for cells in batches:
            g = torch.Generator().manual_seed(11)
            genes = torch.stack([torch.randperm(n_genes, generator=g)[:S] for _ in range(8)]).repeat(cells // 8, 1).to(device)
            counts = torch.poisson(torch.full((cells, S), 1.5), generator=g).to(device)
            flops = cells * (S * MCAB_ENCODE_FLOPS_PER_GENE + 1.5e6)
            for prec in ("fp32", "fp16", "bf16"):
                vae.precision = prec
                # throughput: 20 calls back to back between two synchronisations (the headline's protocol; a sub-millisecond call
                # timed alone is mostly launch ramp, host enqueue and drain: reported beside it as a latency)
                dt, lat = throughput_and_latency(lambda: vae.encode(counts, genes))
                ach = flops / dt / 1e12
                key = f"{name}_{prec}" + ("" if cells == 1024 else f"_b{cells}")
                rec[key] = {"cells": cells, "tokens_per_cell": S, "cells_per_s": cells / dt, "ms": 1e3 * dt,
                            "ms_single_synchronised_call": 1e3 * lat, "cells_per_s_single_call": cells / lat,
                            "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK[prec] / 1e12, "unit": "TFLOP/s",
                                         "frac": ach / (PEAK[prec] / 1e12), "hbm_in_GBps": cells * S * 12 / dt / 1e9}}
            del genes, counts
        del vae
    return rec


def precision_path(wl, device, prec, steps, warmup, ref_cells=64):
    """The same workload at another operand precision: throughput, fused-kernel roofline against ITS peak, and the error of every
    fast precision against the exact-fp32 path (itself pinned to the reference's golden vectors at ~6e-7 by tests/) on `ref_cells`
    cells x 8 Euler evaluations: scale-relative max error and the elementwise relative error floored at 1 % of max|ref|.
      parity_path          = 'bf16x3' (<= 1e-4 vs exact fp32)
      reference_class_path = 'fp16'   (TF32's mantissa: the arithmetic the reference itself computes in; ~1e-3 vs exact fp32, as
                                       the reference's own TF32 mode is - tests/test_gpu_dit.py asserts <= 1.5 x a TF32-operand oracle)"""
    m3 = make_model(wl, prec, device)
    # timed exactly like the headline: the same number of steps after the same number of warm-up steps
    dt, blocks, _, (z2, cond2, scales, B) = time_workload(m3, wl, device, steps, warmup, False, 1, 0, time_blocks=True)
    dt /= steps
    n_fwd = (2 + n_passes(wl)) * B
    rec = {"precision": prec, "cells_per_s": B / dt, "ms_per_step": 1e3 * dt, "steps": steps, "warmup": warmup, "arithmetic": ARITH[prec],
           "dit_fwd_mfma_frac": executed_flops_per_eval(wl, B, cond2) * wl["evals"] / dt / PEAK[prec]}
    if blocks and blocks[0] > 0:
        rec["roofline"] = fused_kernel_roofline(m3, blocks, n_fwd, prec, wl, wl["evals"])
    zs, cs, ss = make_inputs(wl, ref_cells, device, seed=31)
    outs = {}
    for pr in ("fp32", "bf16x3", "fp16", "bf16"):
        m3.precision = pr
        outs[pr] = m3.sample_ode_cfg(zs, cs, ss, 9, "euler").double()
    ref = outs["fp32"]
    for pr in ("bf16x3", "fp16", "bf16"):
        d = (outs[pr] - ref).abs()
        rec[f"err_{pr}_vs_fp32"] = {"max_abs_over_max_ref": float(d.max() / ref.abs().max()),
                                    "max_rel_floor_1pct": float((d / ref.abs().clamp_min(0.01 * float(ref.abs().max()))).max())}
    rec["err_note"] = f"{ref_cells} cells x 8 Euler CFG evaluations, against the exact-fp32 path on the same device"
    return rec


def cpu_baseline(m, wl, target_s=3.0):
    """The CPU oracle (plain-torch fp32 restatement of the reference, validated against it in tests/) timed on this box's host
    cores over a bounded sample of the SAME workload: the full number of CFG evaluations on a reduced cell count (chosen from a
    short probe so that one solve takes about `target_s` seconds), warm-up 1 solve, median of 3 - measured, not extrapolated.
    Thread count is FIXED (round 5: the per-box calibration picked 16 threads on one box and 32 on the next and the number swung
    2x between rounds): `value` / `cores` are the 16-thread figure, `value_32_threads` the same sample on 32 threads (both capped at
    the host's thread count).  Larger OpenMP teams on these 1.5k-row GEMMs are pathological (256 threads: seconds per cell-evaluation)."""
    from oracle.dit import DiTConfig, dit_forward_with_cfg
    from oracle.transport import sample_ode_fixed
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    cfg = DiTConfig(class_vocab_sizes=wl["vocab"], condition_strategy=wl["strategy"])
    n_evals = wl["evals"]
    steps = n_evals + 1 if wl["method"] == "euler" else n_evals // 2 + 1

    def solve(cells, n_steps):
        z2, cond2, scales = make_inputs(wl, cells, "cpu", seed=99)
        t0 = time.perf_counter()
        sample_ode_fixed(z2, lambda x, t: dit_forward_with_cfg(sd, cfg, x, t, cond2, scales), n_steps, wl["method"])
        return time.perf_counter() - t0

    all_cores = os.cpu_count() or 1
    t16, t32 = min(all_cores, 16), min(all_cores, 32)
    torch.set_num_threads(t16)
    solve(2, 2)                                         # first touch
    per_cell_eval = solve(32, 3) / (32 * 2)             # seconds per cell per evaluation on the 16-thread team
    cells = int(min(512, max(8, target_s / (per_cell_eval * n_evals))))
    res = {}
    for nt in sorted({t16, t32}):
        torch.set_num_threads(nt)
        solve(cells, steps)                             # warm-up solve
        times = [solve(cells, steps) for _ in range(3)]
        res[nt] = (statistics.median(times), times)
    torch.set_num_threads(t16)
    cpu_model = platform.processor() or ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), cpu_model)
    except OSError:
        pass
    med, times = res[t16]
    rec = {"value": cells / med, "unit": "cells/s", "cores": t16, "host_cores": all_cores, "kind": "port", "cpu": cpu_model,
           "sample": f"{cells} cells x all {n_evals} CFG evaluations ({wl['method']}), fp32 torch CPU ops on a fixed team of {t16} threads, "
                     f"median of 3 solves ({', '.join(f'{t:.2f}' for t in times)} s) after one warm-up solve; measured, not extrapolated"}
    if t32 != t16:
        rec["value_32_threads"] = cells / res[t32][0]
    return rec


def mfma_ceiling_record():
    """What the matrix pipe itself sustains on THIS box under its power budget: a register-only bf16 MFMA loop (no memory traffic,
    scldm_mfma_sustained_tflops) on zero, uniform and N(0, 1) operand fragments.  Same instruction stream; the part clocks to its power
    budget, so realistic operands run at about half the nominal 2.5 PFLOP/s.  The `roofline.frac` of this file keeps the nominal
    denominator; `frac_of_sustained_normal` prices the fused kernel against what bare MFMAs reach on the same device in the same run."""
    import ctypes as C
    from scldm_amd import _lib
    L = _lib.lib()
    rec = {}
    for fill, name in ((0, "zero"), (1, "uniform"), (2, "normal")):
        v = C.c_double()
        _lib.check(L.scldm_mfma_sustained_tflops(fill, 12000, C.byref(v)), "scldm_mfma_sustained_tflops")
        rec[name + "_tflops"] = v.value
    v = C.c_double()
    _lib.check(L.scldm_mfma_sustained_tflops(2 | 4, 12000, C.byref(v)), "scldm_mfma_sustained_tflops")
    rec["normal_fp16_tflops"] = v.value      # the same loop on v_mfma_f32_32x32x16_f16: the fp16 policy's ceiling on this box
    rec["note"] = ("register-only v_mfma_f32_32x32x16_bf16 loop, 2 waves per SIMD, ~0.6 s per fill; identical instruction stream - the "
                   "difference between fills is clock (power budget)")
    return rec


def default_sampler_record(m, wl, device, cells=4096):
    """The reference's DEFAULT sampler (LatentDiffusion.sample -> Sampler.sample_ode() with no arguments: adaptive dopri5, atol = rtol =
    1e-5; src/scldm/models.py:793, transport/transport.py:324-331): host-driven Dormand-Prince steps over the fused forward_with_cfg.
    cells/s = requested cells / wall time of one solve (median of 3 after one warm-up solve); the number of CFG evaluations the
    controller took is reported beside it."""
    B = min(cells, wl["B"])
    w2 = dict(wl); w2["B"] = B
    z2, cond2, scales = make_inputs(w2, B, device, seed=77)
    from scldm_amd.transport import Sampler, create_transport
    fn = Sampler(create_transport()).sample_ode()              # exactly the reference's call
    model_fn = lambda x, t, **kw: m.forward_with_cfg(x, t, **kw, cfg_scale=scales)
    fn(z2, model_fn, condition=cond2)
    sync(device)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        fn(z2, model_fn, condition=cond2)
        sync(device)
        ts.append(time.perf_counter() - t0)
    dt = statistics.median(ts)
    st = getattr(fn, "last_stats", {}) or {}
    ev = st.get("evaluations")
    rec = {"sampler": "dopri5 (atol = rtol = 1e-5, 50 save points): Sampler.sample_ode() defaults", "cells_per_gpu": B, "cells_per_s": B / dt,
           "ms_per_solve": 1e3 * dt, "cfg_evaluations": ev, "accepted_steps": len(st.get("accepted_steps", [])),
           "rejected_steps": len(st.get("rejected_steps", [])), "precision": m.precision}
    if ev:
        n_fwd = (2 + n_passes(wl)) * B
        rec["dit_fwd_mfma_frac"] = n_fwd * FLOPS_TRUNK_PER_SAMPLE_FWD * ev / dt / PEAK[m.precision]
    return rec


def vae_training_record(device, batches=(32, 512), n_genes=17002, S=6147):
    """BASELINE configs[0] on the GPU: one VAE optimisation step = TransformerVAE.forward -> -log_nb_positive(...).sum(1).mean() ->
    HIP backward -> fused AdamW (src/scldm/models.py:243-249, vae.py:29-56) at the dentate_gyrus shape.  FLOPs per cell: forward =
    encode (S x 6 528 + 1.5e6) + decode (G x 23 104 + 3.3e6), backward counted as 2 x forward."""
    from scldm_amd.distributions import log_nb_positive
    rec = {}
    for B in batches:
        vae = make_vae(n_genes, device).train()
        g = torch.Generator().manual_seed(5)
        counts = torch.poisson(torch.full((B, n_genes), 0.5), generator=g).to(device)
        genes = torch.arange(n_genes, device=device).repeat(B, 1)
        gs = torch.stack([torch.sort(torch.randperm(n_genes, generator=g)[:S]).values for _ in range(B)]).to(device)
        cs = counts.gather(1, gs)
        lib = counts.sum(1, keepdim=True)
        opt = make_optimizer(vae.parameters(), 1e-3, max_grad_norm=10.0)     # the trainer's gradient_clip_val (training/default.yaml:15-16)

        def step(ev=None):
            opt.zero_grad(set_to_none=True)
            if ev: ev[0].record()
            params, _ = vae(counts, genes, lib, cs, gs)
            loss = (-log_nb_positive(counts, params["mu"], params["theta"])).sum(1).mean()
            if ev: ev[1].record()
            loss.backward()
            if ev: ev[2].record()
            opt.step()
            return loss
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        n = 20 if B <= 64 else 5
        import gc
        gc.collect()
        gc.disable()         # (a full collection of the bench process is ~70 ms: one of them inside ten 2.5 ms steps read 6.3 ms per step)
        try:
            t0 = time.perf_counter()
            for _ in range(n):
                loss = step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
        finally:
            gc.enable()
        # device time of the forward (+ loss) and of the backward, medians of separately recorded steps
        fwd, bwd = [], []
        for _ in range(5):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            step(ev)
            torch.cuda.synchronize()
            fwd.append(ev[0].elapsed_time(ev[1]))
            bwd.append(ev[1].elapsed_time(ev[2]))
        fwd_ms, bwd_ms = statistics.median(fwd), statistics.median(bwd)
        flops = 3 * B * (S * MCAB_ENCODE_FLOPS_PER_GENE + 1.5e6 + n_genes * MCAB_DECODE_FLOPS_PER_GENE + 3.3e6)
        rec[f"b{B}"] = {"cells": B, "n_genes": n_genes, "tokens_per_cell": S, "ms_per_step": 1e3 * dt, "cells_per_s": B / dt,
                        "forward_and_loss_ms": fwd_ms, "backward_ms": bwd_ms, "backward_over_forward": bwd_ms / fwd_ms,
                        "tflops": flops / dt / 1e12, "frac_of_fp32_mfma_peak": flops / dt / PEAK["fp32"], "final_loss": float(loss.detach()),
                        "gradient_clip": "global norm 10.0 inside the optimizer launch", "last_grad_norm": float(opt.last_grad_norm)}
        del vae, opt
        torch.cuda.empty_cache()
    return rec


def synthetic_vocabulary_encoder(vocab, strategy, seed=3):
    """Size-factor statistics of a made-up vocabulary encoder with the attributes LatentDiffusion._sample_log_size_factors reads
    (src/scldm/models.py:473-597): per-label (independent) or per joint class mean / std of the log library size."""
    import random
    from types import SimpleNamespace
    rnd = random.Random(seed)
    names = sorted(vocab)
    if strategy == "joint" and len(names) > 1:
        import itertools
        toks = {"_".join(map(str, idx)): "c" + "_".join(map(str, idx)) for idx in itertools.product(*[range(vocab[k]) for k in names])}
        mu = {c: rnd.uniform(6.5, 9.0) for c in toks.values()}
        sd = {c: rnd.uniform(0.1, 0.5) for c in toks.values()}
        return SimpleNamespace(joint_key="joint", joint_components=names, joint_idx_2_classes=toks,
                               mu_size_factor={"joint": mu}, sd_size_factor={"joint": sd}, size_factor_condition_key=None)
    k = names[0]
    return SimpleNamespace(size_factor_condition_key=k, mu_size_factor={k: {i: rnd.uniform(6.5, 9.0) for i in range(vocab[k])}},
                           sd_size_factor={k: {i: rnd.uniform(0.1, 0.5) for i in range(vocab[k])}})


def generation_end_to_end(wl_name, n_genes, device, precision, reps=5):
    """What one `predict_step` of the reference delivers (src/scldm/models.py:707-819 + _utils.py:192-197), end to end on the device
    path: size factors drawn on device (SizeFactorSampler) -> noise -> fused CFG ODE -> MCAB decode (fp16 operands beside a 16-bit DiT, else fp32) with the negative-binomial
    draw fused in -> CSR assembly on device -> host arrays (indptr / indices / data of the 2B generated rows + the 2B latents).
    cells/s = requested cells / wall time of the whole chain (median of `reps` after one warm-up), with the device time of each stage."""
    from scldm_amd.datamodule import dense_to_csr, to_host
    from scldm_amd.sampling import SizeFactorSampler, sample_latents
    wl = dict(WORKLOADS[wl_name])
    B = wl["B"]
    m = make_model(wl, precision, device)
    vae = make_vae(n_genes, device)
    # the decode runs in the reference's arithmetic class (TF32 mantissa = fp16 operands) beside a 16-bit DiT, exact fp32 otherwise
    vae.precision = "fp16" if precision in ("bf16", "fp16") else "fp32"
    smp = SizeFactorSampler(synthetic_vocabulary_encoder(wl["vocab"], wl["strategy"]), wl["strategy"], device)
    g = torch.Generator().manual_seed(21)
    cond = {k: torch.randint(0, v, (B,), generator=g).to(device) for k, v in wl["vocab"].items()}
    scales = {k: wl["scale"] for k in wl["vocab"]}
    genes2 = torch.arange(n_genes, device=device).repeat(2 * B, 1)
    steps = wl["evals"] + 1 if wl["method"] == "euler" else wl["evals"] // 2 + 1

    def once(ev=None):
        rec = (lambda i: ev[i].record()) if ev else (lambda i: None)
        rec(0)
        sf = smp.sample(cond, B)
        z0 = torch.randn((B, 16, 16), device=device)
        rec(1)
        z = sample_latents(m, z0, cond, scales, steps, wl["method"])
        rec(2)
        lib = torch.exp(sf).view(-1, 1)
        counts = vae.decode_sample(z, genes2, torch.cat([lib, lib], dim=0))
        rec(3)
        indptr, indices, data = dense_to_csr(counts)
        rec(4)
        host = to_host(indptr, indices, data, z)     # pinned staging, one synchronisation (scldm_amd.datamodule.to_host)
        rec(5)
        return host
    once()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        host = once()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    dt = statistics.median(ts)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    once(ev)
    torch.cuda.synchronize()
    stage = [ev[i].elapsed_time(ev[i + 1]) for i in range(5)]
    nnz = int(host[2].numel())
    # the same chain over K consecutive batches as the two-stream pipeline (scldm_amd.sampling.generate_cells_stream: batch i's decode /
    # draw / CSR / host copies beside batch i + 1's ODE) - the reference's prediction loop is a loop over batches (models.py:707-764)
    from scldm_amd.sampling import generate_cells_stream
    K = 8
    genes1 = genes2[:B]
    def run_stream(merge=1):
        return sum(int(out[2].numel()) for out in generate_cells_stream(m, vae, [cond] * K, scales, genes1, steps, wl["method"], size_factor_sampler=smp,
                                                                        merge_batches=merge))
    def time_stream(merge):
        run_stream(merge)
        torch.cuda.synchronize()
        tp = []
        for _ in range(3):
            t0 = time.perf_counter()
            run_stream(merge)
            torch.cuda.synchronize()
            tp.append((time.perf_counter() - t0) / K)
        return statistics.median(tp)
    dtp, dtm = time_stream(1), time_stream(4)
    pipelined = {"batches": K, "cells_per_s": B / dtp, "ms_per_batch": 1e3 * dtp,
                 "path": "generate_cells_stream: ODE of batch i+1 on the caller's stream || decode_sample + dense_to_csr + to_host of batch i on a second stream",
                 "four_batches_per_solve": {"cells_per_s": B / dtm, "ms_per_batch": 1e3 * dtm,
                                            "note": "merge_batches=4: four consecutive batches share one solve (bit-identical arrays per batch; a cell's trajectory does not depend on its batch)"}}
    return {"workload": wl_name, "pipelined": pipelined, "cells": B, "generated_rows": 2 * B, "n_genes": n_genes, "cfg_evaluations": wl["evals"], "method": wl["method"],
            "dit_precision": precision, "decode_precision": vae.precision, "cells_per_s": B / dt, "ms": 1e3 * dt, "ms_each": [round(1e3 * t, 3) for t in ts],
            "stage_ms": {"size_factors_and_noise": stage[0], "ode": stage[1], "decode_and_nb_draw": stage[2], "csr_assembly": stage[3],
                         "device_to_host": stage[4]},
            "nnz_fraction": nnz / (2 * B * n_genes), "host_bytes": sum(int(t.numel()) * t.element_size() for t in host),
            "path": "SizeFactorSampler.sample -> sample_latents (scldm_sample_ode) -> TransformerVAE.decode_sample -> dense_to_csr -> to_host (pinned)"}


# --------------------------------------------------------------------------------------------------------------------
# the stdout line: the contract's keys + the scalars a reader needs, <= 6 KB; everything else goes to profiles/bench_last.json
# --------------------------------------------------------------------------------------------------------------------
def _dig(d, *path, default=None):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d


def compact_line(result):
    keep = ("metric", "value", "unit", "n_gpus", "rccl_ranks", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "dit_fwd_tflops_per_gpu", "dit_fwd_mfma_frac", "roofline", "cpu_baseline", "per_rank_ms_per_step",
            "allgather_ms", "cross_rank_check", "strong_scaling", "train_tflops_per_gpu", "final_loss", "data_parallel")
    out = {k: result[k] for k in keep if k in result}
    if isinstance(out.get("cpu_baseline"), dict) and len(out["cpu_baseline"].get("sample", "")) > 420:
        out["cpu_baseline"] = dict(out["cpu_baseline"], sample=out["cpu_baseline"]["sample"][:417] + "...")
    sc = {}
    def put(key, *path):
        v = _dig(result, *path)
        if v is not None:
            sc[key] = round(v, 6) if isinstance(v, float) else v
    # the same workload at the reference's arithmetic class (fp16 = TF32's mantissa) and at the <= 1e-4 parity precision (bf16x3)
    put("fp16_cells_per_s", "reference_class_path", "cells_per_s"); put("fp16_frac", "reference_class_path", "roofline", "frac")
    put("fp16_err_vs_fp32", "reference_class_path", "err_fp16_vs_fp32", "max_abs_over_max_ref")
    put("bf16x3_cells_per_s", "parity_path", "cells_per_s"); put("bf16x3_frac", "parity_path", "roofline", "frac")
    put("bf16x3_err_vs_fp32", "parity_path", "err_bf16x3_vs_fp32", "max_abs_over_max_ref")
    put("bf16_err_vs_fp32", "parity_path", "err_bf16_vs_fp32", "max_abs_over_max_ref")
    put("bf16_cells_per_s", "throughput_path", "cells_per_s"); put("bf16_frac", "throughput_path", "roofline", "frac")
    for w in result.get("other_workloads", []) or []:
        tag = {"dentate_b512_euler50": "b512", "dentate_b128_euler50": "b128", "parse1m_b1024_euler100": "b1024", "hlca_b2048_heun100": "hlca_b2048_heun100",
               "parse1m_b8192_euler100_strong": "b8192"}.get(w["workload"], w["workload"])
        sc[f"cells_per_s_{tag}"] = round(w["cells_per_s"], 1)
        sc[f"frac_{tag}"] = round(w["dit_fwd_mfma_frac"], 4)
    put("dopri5_cells_per_s", "default_sampler", "cells_per_s"); put("dopri5_frac", "default_sampler", "dit_fwd_mfma_frac")
    put("dopri5_cfg_evaluations", "default_sampler", "cfg_evaluations")
    put("with_decode_cells_per_s", "with_vae_decode", "cells_per_s")
    put("decode_fp32_rows_per_s", "with_vae_decode", "decode_only_fp32", "rows_per_s"); put("decode_fp32_frac", "with_vae_decode", "decode_only_fp32", "roofline", "frac")
    put("decode_fp16_rows_per_s", "with_vae_decode", "decode_only_fp16", "rows_per_s"); put("decode_fp16_frac", "with_vae_decode", "decode_only_fp16", "roofline", "frac")
    put("decode_bf16_rows_per_s", "with_vae_decode", "decode_only_bf16", "rows_per_s"); put("decode_bf16_frac", "with_vae_decode", "decode_only_bf16", "roofline", "frac")
    put("encode_fp32_cells_per_s", "with_vae_encode", "dentate_fp32", "cells_per_s"); put("encode_fp32_frac", "with_vae_encode", "dentate_fp32", "roofline", "frac")
    put("encode_fp16_cells_per_s", "with_vae_encode", "dentate_fp16", "cells_per_s"); put("encode_fp16_frac", "with_vae_encode", "dentate_fp16", "roofline", "frac")
    put("encode_bf16_cells_per_s", "with_vae_encode", "dentate_bf16", "cells_per_s"); put("encode_bf16_frac", "with_vae_encode", "dentate_bf16", "roofline", "frac")
    for prec in ("fp32", "fp16"):
        put(f"mcab_dec_gene_{prec}_frac", "mcab_roofline", prec, "dec_gene", "frac"); put(f"mcab_dec_gene_{prec}_us", "mcab_roofline", prec, "dec_gene", "avg_launch_us")
        put(f"mcab_enc_pool_{prec}_frac", "mcab_roofline", prec, "enc_pool", "frac"); put(f"mcab_enc_pool_{prec}_us", "mcab_roofline", prec, "enc_pool", "avg_launch_us")
    put("train_e2e_ms", "training_step_end_to_end", "ms_per_step"); put("train_autograd_graph_ms", "training_step_autograd_graph", "ms_per_step")
    put("e2e_cells_per_s_dentate512", "generation_end_to_end", "dentate_b512_euler50", "cells_per_s")
    put("e2e_ms_dentate512", "generation_end_to_end", "dentate_b512_euler50", "ms")
    put("e2e_pipelined_cells_per_s_dentate512", "generation_end_to_end", "dentate_b512_euler50", "pipelined", "cells_per_s")
    put("e2e_merged4_cells_per_s_dentate512", "generation_end_to_end", "dentate_b512_euler50", "pipelined", "four_batches_per_solve", "cells_per_s")
    put("e2e_cells_per_s_dentate128", "generation_end_to_end", "dentate_b128_euler50", "cells_per_s")
    put("e2e_pipelined_cells_per_s_dentate128", "generation_end_to_end", "dentate_b128_euler50", "pipelined", "cells_per_s")
    put("e2e_merged4_cells_per_s_dentate128", "generation_end_to_end", "dentate_b128_euler50", "pipelined", "four_batches_per_solve", "cells_per_s")
    put("e2e_pipelined_cells_per_s_parse1m1024", "generation_end_to_end", "parse1m_b1024_euler100", "pipelined", "cells_per_s")
    put("e2e_cells_per_s_parse1m1024", "generation_end_to_end", "parse1m_b1024_euler100", "cells_per_s")
    put("e2e_ms_parse1m1024", "generation_end_to_end", "parse1m_b1024_euler100", "ms")
    put("train_ms", "training_step", "ms_per_step"); put("train_cells_per_s", "training_step", "cells_per_s"); put("train_tflops", "training_step", "tflops")
    put("train_torch_adamw_ms", "training_step_torch_adamw", "ms_per_step"); put("train_eager_ms", "training_step_eager", "ms_per_step")
    put("train_fp16_ms", "training_step_fp16", "ms_per_step"); put("train_b256_ms", "training_step_b256", "ms_per_step")
    put("train_ditl_b1024_tflops", "training_step_ditl_b1024", "tflops"); put("train_ditl_b256_tflops", "training_step_ditl", "tflops")
    put("ditl_sampling_tflops", "ditl_sampling", "tflops")
    put("vae_train_b32_ms", "vae_training_step", "b32", "ms_per_step"); put("vae_train_b512_ms", "vae_training_step", "b512", "ms_per_step")
    put("guidance1_direct_cells_per_s", "guidance1_direct", "cells_per_s")
    put("mfma_sustained_normal_tflops", "mfma_sustained_ceiling", "normal_tflops")
    put("mfma_sustained_normal_fp16_tflops", "mfma_sustained_ceiling", "normal_fp16_tflops")
    put("fp16_frac_of_sustained", "reference_class_path", "roofline", "frac_of_sustained_normal")
    put("strong_scaling_cells_per_s", "strong_scaling", "cells_per_s"); put("strong_scaling_ms", "strong_scaling", "ms_per_step")
    out.update(sc)
    out["details"] = "profiles/bench_last.json (every record behind these scalars; written by this run)"
    return out


def kernel_source_sha256():
    """fingerprint of the fused layer kernel's sources (the PMC summary records the one it was collected on)"""
    import hashlib
    h = hashlib.sha256()
    for f in ("dit_forward.hpp", "common.hpp"):
        with open(os.path.join(ROOT, "scldm_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


_T0 = time.perf_counter()


def note(msg):
    """progress to stderr (the JSON line on stdout stays the only stdout line of rank 0)"""
    print(f"[bench +{time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def emit(result, rank):
    if rank == 0:
        sys.stdout.flush()
        try:  # RCCL prints its banner through C stdio: flush that buffer first so the JSON line is the last line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        try:   # the long per-workload records: next to the profiles, not on stdout (the driver keeps an 8 KB tail)
            if os.environ.get("BENCH_FAKE") == "1":
                raise OSError("BENCH_FAKE run: no details file")
            os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
            with open(os.path.join(ROOT, "profiles", "bench_last.json"), "w") as f:
                json.dump(result, f, indent=1)
        except OSError as e:
            note(f"profiles/bench_last.json not written: {e}")
        line = json.dumps(compact_line(result))
        if len(line) > 6144:
            note(f"stdout line is {len(line)} bytes (> 6 KB)")
        print(line, flush=True)  # the one JSON line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="dentate_b4096_euler100", choices=sorted(WORKLOADS) + sorted(TRAIN_WORKLOADS))
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "bf16x3", "fp32"])
    ap.add_argument("--batch", type=int, default=0, help="override per-GPU batch")
    ap.add_argument("--global-cells", type=int, default=0, help="override the GLOBAL cell count of the strong-scaling leg (N > 1)")
    ap.add_argument("--evals", type=int, default=0, help="override the number of CFG evaluations (profiling only; not a valid bench line)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the short extra-workload measurements")
    args = ap.parse_args()
    fake = os.environ.get("BENCH_FAKE") == "1"   # CPU / gloo test of the launcher and harness (tests/test_bench_launcher.py)

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            # nothing in this process has touched a GPU yet (device_count() does not initialise one on this platform)
            if not fake and args.gpus > torch.cuda.device_count():
                raise SystemExit(f"bench.py --gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible on this node")
            raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started as one of WORLD_SIZE={os.environ['WORLD_SIZE']} ranks: "
                         "launch as many ranks as GPUs (python -m torch.distributed.run --nproc-per-node N bench.py --gpus N)")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"  # the env switch exercises the RCCL path on one GPU
    if fake:
        device = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
        if local_rank >= torch.cuda.device_count():
            raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if fake:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    rccl_ranks = dist.get_world_size() if dist_on else 1

    if args.workload in TRAIN_WORKLOADS:   # SURVEY 8a row T1 / BASELINE configs[4]: not the headline metric, its own line
        wl = dict(TRAIN_WORKLOADS[args.workload])
        if args.batch:
            wl["B"] = args.batch
        dt, loss = time_training(wl, args.precision, device, args.steps, max(args.warmup, 1), dist_on, world)
        value = world * wl["B"] / (dt / args.steps)
        result = {"metric": "training cells/sec (flow-matching step: forward + backward + gradient all-reduce + AdamW)", "value": value,
                  "unit": "cells/s", "n_gpus": world, "rccl_ranks": rccl_ranks, "steps": args.steps, "warmup": max(args.warmup, 1),
                  "ms_per_step": 1e3 * dt / args.steps,
                  "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
                  "config": {"workload": args.workload, "cells_per_gpu": wl["B"], "global_cells": world * wl["B"],
                             "class_vocab_sizes": wl["vocab"], "condition_strategy": wl["strategy"], "optimizer": "scldm_amd.optim.AdamW (torch.optim.AdamW arithmetic, one launch)",
                             "parallelism": f"data-parallel x{world}, bucketed in-place gradient all-reduce overlapped with the backward" if dist_on else "single GPU"},
                  "train_tflops_per_gpu": 3 * dit_flops(**{k: v for k, v in wl.get("shape", {}).items() if k != "n_head"}) * wl["B"]
                  / (dt / args.steps) / 1e12, "final_loss": loss, "data_parallel": time_training.info}
        if "shape" in wl:
            result["config"]["dit_shape"] = wl["shape"]
        if dist_on:
            dist.barrier()
            dist.destroy_process_group()
        emit(result, rank)
        return

    wl = dict(WORKLOADS[args.workload])
    if args.batch:
        wl["B"] = args.batch
    if args.evals:
        wl["evals"] = args.evals
    m = make_model(wl, args.precision, device)
    dt, blocks, out, (z2, cond2, scales, B) = time_workload(m, wl, device, args.steps, args.warmup, dist_on, world, rank, time_blocks=True)
    ms_per_step = 1e3 * dt / args.steps
    strong = bool(wl.get("strong"))
    cells = wl["B"] if strong else world * B
    value = cells / (dt / args.steps)
    n_fwd = (2 + n_passes(wl)) * B
    evals = wl["evals"]
    result = {
        "metric": "cells/sec (whole node) @100 Euler steps; DiT-fwd MFMA util % of peak",
        "value": value, "unit": "cells/s", "n_gpus": world, "rccl_ranks": rccl_ranks, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": args.precision, "data": "synthetic",
        "config": {"workload": args.workload, "cells_per_gpu": B, "global_cells": cells, "cfg_evaluations": evals,
                   "method": wl["method"], "guidance_scale": wl["scale"], "condition_strategy": wl["strategy"],
                   "class_vocab_sizes": wl["vocab"], "sample_forwards_per_evaluation_per_gpu": n_fwd,
                   "gathered_rows": int(out.shape[0]) if out is not None else None,
                   "parallelism": f"batch-sharded x{world}, one all-gather of latents" if dist_on else "single GPU"},
    }
    if not fake:
        peak = PEAK[args.precision]
        step_s = dt / args.steps
        ex = executed_flops_per_eval(wl, B, cond2) * evals
        result["dit_fwd_tflops_per_gpu"] = ex / step_s / 1e12
        result["dit_fwd_mfma_frac"] = ex / step_s / peak                          # FLOPs the path executes (adaLN per conditioning row)
        result["dit_fwd_mfma_frac_algorithmic"] = n_fwd * FLOPS_PER_SAMPLE_FWD * evals / step_s / peak   # BASELINE.md section 3 formula
        if blocks and blocks[0] > 0:
            rl = fused_kernel_roofline(m, blocks, n_fwd, args.precision, wl, evals)
            traffic, traffic_src = None, None
            pmc = os.path.join(ROOT, "profiles", "pmc_dit_forward_kernel.json")
            if args.precision == "bf16" and args.workload == "dentate_b4096_euler100" and not args.batch and os.path.exists(pmc):
                # HBM bytes per launch of this kernel from the committed rocprofv3 PMC passes of this same workload
                # (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate passes; see the file's note)
                with open(pmc) as f:
                    pj = json.load(f)
                sha = kernel_source_sha256()
                if pj.get("layers_per_launch", rl["layers_per_launch"]) != rl["layers_per_launch"]:
                    traffic_src = "stale: profiles/pmc_dit_forward_kernel.json was collected with another number of layers per launch"
                elif pj.get("kernel_source_sha256") not in (None, sha):
                    traffic_src = "stale: the kernel sources changed after profiles/pmc_dit_forward_kernel.json was collected (tools/pmc_summary.py)"
                else:
                    traffic, traffic_src = pj.get("hbm_bytes_per_launch"), "profiles/pmc_dit_forward_kernel.json"
            rl["traffic"], rl["traffic_source"] = traffic, traffic_src
            rl["traffic_ratio"] = (traffic / rl["algorithmic_hbm_bytes_per_launch"]) if traffic else None
            result["roofline"] = rl
    note(f"main workload done: {value:.0f} cells/s")
    if rank == 0 and not fake and "roofline" in result and not args.no_extra:   # (--no-extra: profiling runs keep their kernel list clean)
        try:
            ceil = mfma_ceiling_record()
            result["mfma_sustained_ceiling"] = ceil
            result["roofline"]["frac_of_sustained_normal"] = result["roofline"]["achieved"] / ceil["normal_tflops"]
        except Exception as e:   # a measurement aid: never takes the headline line down
            result["mfma_sustained_ceiling"] = {"error": repr(e)}
    if rank == 0 and not dist_on and not fake:
        if not args.no_extra:
            if args.precision != "bf16x3":
                result["parity_path"] = precision_path(wl, device, "bf16x3", args.steps, args.warmup)
                note("parity path done")
            if args.precision != "fp16":
                result["reference_class_path"] = precision_path(wl, device, "fp16", args.steps, args.warmup)
                c16 = _dig(result, "mfma_sustained_ceiling", "normal_fp16_tflops")
                a16 = _dig(result, "reference_class_path", "roofline", "achieved")
                if c16 and a16:   # the fp16 policy against what a bare fp16 MFMA loop sustains on this box in this run
                    result["reference_class_path"]["roofline"]["frac_of_sustained_normal"] = a16 / c16
                note("reference-class (fp16) path done")
            if args.precision != "bf16":
                result["throughput_path"] = precision_path(wl, device, "bf16", args.steps, args.warmup)
                note("bf16 throughput path done")
            extra = []
            for name in ("dentate_b512_euler50", "dentate_b128_euler50", "parse1m_b1024_euler100", "hlca_b2048_heun100", "parse1m_b8192_euler100_strong"):
                if name == args.workload:
                    continue
                w2 = dict(WORKLOADS[name])
                m2 = make_model(w2, args.precision, device)
                # median of >= 5 individually synchronised steps after >= 2 warm-up steps (single-step timings carry +-10 %)
                n_t, n_w = max(5, min(args.steps, 7)), max(2, args.warmup)
                ts, (_, c2, _, B2) = time_steps_each(m2, w2, device, n_t, n_w)
                d2 = statistics.median(ts)
                extra.append({"workload": name, "cells_per_s": B2 / d2, "ms_per_step": 1e3 * d2, "steps": n_t, "warmup": n_w,
                              "spread": (max(ts) - min(ts)) / d2, "ms_each": [round(1e3 * t, 3) for t in ts],
                              "dit_fwd_mfma_frac": executed_flops_per_eval(w2, B2, c2) * w2["evals"] / d2 / PEAK[args.precision]})
                del m2
            result["other_workloads"] = extra
            note("other workloads done")
            if wl["scale"] == 1.0 and n_passes(wl) == 1:
                # the product's opt-in shortcut for guidance exactly 1.0 (SCLDM_OPT_CFG1_DIRECT): guided rows = conditional output, 2B
                # instead of 3B sample-forwards per evaluation.  Reported beside the headline, never as it.
                m.guidance1_direct = True
                ts, _ = time_steps_each(m, wl, device, 3, 1)
                m.guidance1_direct = False
                dg = statistics.median(ts)
                result["guidance1_direct"] = {"cells_per_s": wl["B"] / dg, "ms_per_step": 1e3 * dg, "sample_forwards_per_evaluation_per_gpu": 2 * wl["B"],
                                              "note": "opt-in (DiT.guidance1_direct): at guidance scale exactly 1.0 u2 + 1.0 * (c2 - u2) == c2 up to one "
                                                      "fp32 rounding, so the second half's unconditional forward is skipped; within 1e-6 of the default path "
                                                      "(tests/test_gpu_dit.py); the headline above runs all 3B sample-forwards"}
                note("guidance-1 direct done")
            result["default_sampler"] = default_sampler_record(m, wl, device)
            note("default (dopri5) sampler done")
            result["with_vae_decode"] = decode_inclusive(m, wl, device)
            note("decode done")
            result["with_vae_encode"] = encode_record(device)
            note("encode done")
            try:
                result["mcab_roofline"] = mcab_roofline(device)
            except Exception as e:  # noqa: BLE001
                result["mcab_roofline"] = {"error": repr(e)}
            note("mcab roofline done")
            try:   # configs[1] as the reference's predict_step delivers it, and the configs[3] per-GPU shard
                result["generation_end_to_end"] = {
                    "dentate_b128_euler50": generation_end_to_end("dentate_b128_euler50", 17002, device, args.precision),     # the reference's generation batch size (generation.yaml:16)
                    "dentate_b512_euler50": generation_end_to_end("dentate_b512_euler50", 17002, device, args.precision),
                    "parse1m_b1024_euler100": generation_end_to_end("parse1m_b1024_euler100", 2000, device, args.precision)}
            except Exception as e:   # an extra: never takes the headline line down
                result["generation_end_to_end"] = {"error": repr(e)}
            torch.cuda.empty_cache()
            note("end-to-end generation done")
            tw = dict(TRAIN_WORKLOADS["replogle_train_b1024"])
            torch.cuda.empty_cache()
            tprec = "bf16" if args.precision in ("bf16", "fp16") else "fp32"
            fused_ok = tprec == "bf16"
            try:
                dtt, _ = time_training(tw, tprec, device, 20, 5, False, 1, graphed="fused" if fused_ok else True)
            except Exception as e:   # the one-call step must not take the line down: fall back to the graphed autograd step, and say so
                note(f"FusedTrainStep failed ({e!r}): timing GraphedTrainStep")
                fused_ok = False
                torch.cuda.synchronize()
                dtt, _ = time_training(tw, tprec, device, 20, 5, False, 1, graphed=True)
            fused_modes = None
            if fused_ok:   # the same one-call step issued eagerly (no graph: ~60 launches from C per step): faster where the step is device-bound
                dte_f, _ = time_training(tw, tprec, device, 20, 5, False, 1, graphed="fused_eager")
                fused_modes = {"hip_graph_replay_ms": 1e3 * dtt / 20, "eager_one_c_call_ms": 1e3 * dte_f / 20}
                dtt = min(dtt, dte_f)
            dtt /= 2
            result["training_step"] = {"workload": "replogle_train_b1024", "cells_per_s": tw["B"] / (dtt / 10), "ms_per_step": 1e3 * dtt / 10,
                                       "tflops": 3 * FLOPS_PER_SAMPLE_FWD * tw["B"] / (dtt / 10) / 1e12, "dtype": tprec,
                                       "launch": ("FusedTrainStep: scldm_dit_train_step (batch preparation, forward, loss, backward, AdamW + EMA hook: one C call), "
                                                  "the faster of a HIP-graph replay and the eager call (fused_step_modes)" if fused_ok else
                                                  "GraphedTrainStep: the whole step (training_losses forward, HIP backward, AdamW) replayed as one HIP graph"
                                                  if time_training.graphed else "eager train_step (graph capture failed: see stderr)"),
                                       "path": ("fused: REC forward + dit_backward_kernel + batched bf16 wgrad, activation record 32 KB per cell "
                                                "per layer (DESIGN 4.5)") if tprec == "bf16" else "generic GEMM path (DESIGN 4.6)"}
            if fused_modes:
                result["training_step"]["fused_step_modes"] = fused_modes
            if tprec == "bf16":
                # the same step at the reference's own training precision class (fp16 operands = TF32's mantissa, loss-scaled backward)
                torch.cuda.empty_cache()
                dth, _ = time_training(tw, "fp16", device, 20, 5, False, 1, graphed="fused" if fused_ok else True)
                dth /= 2
                result["training_step_fp16"] = {"workload": "replogle_train_b1024", "cells_per_s": tw["B"] / (dth / 10), "ms_per_step": 1e3 * dth / 10,
                                                "tflops": 3 * FLOPS_PER_SAMPLE_FWD * tw["B"] / (dth / 10) / 1e12, "dtype": "fp16",
                                                "path": "fused route, fp16 operands (10 mantissa bits = the reference's TF32 training arithmetic), "
                                                        "device-side loss scaling of the backward"}
            torch.cuda.empty_cache()
            dte, _ = time_training(tw, tprec, device, 10, 5, False, 1)
            result["training_step_eager"] = {"workload": "replogle_train_b1024", "ms_per_step": 1e3 * dte / 10, "dtype": tprec,
                                             "launch": "scldm_amd.training.train_step: eager (~100 kernel launches + autograd + optimizer Python per step)"}
            d256, _ = time_training(dict(tw, B=256), tprec, device, 20, 5, False, 1, graphed="fused" if fused_ok else True)   # the small-batch step
            dgr, _ = time_training(tw, tprec, device, 20, 5, False, 1, graphed=True)
            result["training_step_autograd_graph"] = {"workload": "replogle_train_b1024", "ms_per_step": 1e3 * dgr / 20, "dtype": tprec,
                                                      "launch": "GraphedTrainStep: Transport.training_losses -> autograd -> optimizer.step() captured as one HIP graph (round 5's form)"}
            if fused_ok:
                try:
                    result["training_step_end_to_end"] = train_end_to_end(tw, tprec, device)
                except Exception as e:  # noqa: BLE001
                    result["training_step_end_to_end"] = {"error": repr(e)}
            torch.cuda.empty_cache()
            dto, _ = time_training(tw, tprec, device, 10, 5, False, 1, optimizer="torch")
            result["training_step_torch_adamw"] = {"workload": "replogle_train_b1024", "ms_per_step": 1e3 * dto / 10, "dtype": tprec,
                                                   "optimizer": "torch.optim.AdamW(fused=True) instead of scldm_amd.optim.AdamW"}
            result["training_step"]["optimizer"] = "scldm_amd.optim.AdamW (torch.optim.AdamW's arithmetic and state_dict, one HIP launch per step)"
            result["training_step_b256"] = {"workload": "replogle_train_b1024 at 256 cells", "cells_per_s": 256 / (d256 / 20), "ms_per_step": 1e3 * d256 / 20,
                                            "dtype": tprec}
            note("training step done")
            try:
                result["vae_training_step"] = vae_training_record(device)
            except Exception as e:   # an extra: never takes the headline line down
                result["vae_training_step"] = {"error": repr(e)}
            note("VAE training step done")
            if tprec == "bf16":   # BASELINE configs[4] names a DiT-L denoiser: the same step on that shape (generic path, bf16 arrays + bgemm)
                try:
                    torch.cuda.empty_cache()
                    for key, name in (("training_step_ditl", "replogle_train_ditl_b256"), ("training_step_ditl_b1024", "replogle_train_ditl_b1024")):
                        tl = dict(TRAIN_WORKLOADS[name])
                        dtl, _ = time_training(tl, "bf16", device, 4, 2, False, 1)
                        result[key] = {"workload": name, "cells_per_s": tl["B"] / (dtl / 4), "ms_per_step": 1e3 * dtl / 4,
                                       "tflops": 3 * dit_flops(n_embed=1024, n_layer=24) * tl["B"] / (dtl / 4) / 1e12, "dtype": "bf16",
                                       "path": "generic: bf16 operand arrays, LDS-DMA 256-tile GEMMs (bgemm8_kernel) / bgemm_kernel, matrix-core attention, "
                                               "batched weight gradients beside the chain (DESIGN 4.6, HISTORY 4.4b-c)"}
                        torch.cuda.empty_cache()
                    # the same DiT-L shape sampling (generic GEMM route under forward_with_cfg; 20 Euler CFG evaluations, 256 cells)
                    wL = dict(vocab={"cell_line": 4, "gene": 2024}, strategy="joint", B=256, evals=20, method="euler", scale=1.0,
                              shape=dict(n_embed=1024, n_layer=24, n_head=16))
                    mL = make_model(wL, "bf16", device)
                    tsL, _ = time_steps_each(mL, wL, device, 3, 1)
                    dL = statistics.median(tsL)
                    result["ditl_sampling"] = {"cells": 256, "cfg_evaluations": 20, "cells_per_s": 256 / dL, "ms_per_solve": 1e3 * dL,
                                               "cells_per_s_at_100_evaluations": 256 / (dL * 5),
                                               "tflops": 3 * 256 * dit_flops(n_embed=1024, n_layer=24) * 20 / dL / 1e12, "dtype": "bf16"}
                    del mL
                    torch.cuda.empty_cache()
                except Exception as e:   # an extra: never takes the headline line down
                    result["training_step_ditl"] = {"error": repr(e)}
                note("DiT-L training step done")
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(m, wl)
            note("cpu baseline done")
    if dist_on:
        result["per_rank_ms_per_step"] = [1e3 * v / args.steps for v in getattr(time_workload, "per_rank_s", [])]
        result["allgather_ms"] = 1e3 * time_allgather(out, B, device, world)
        result["cross_rank_check"] = cross_rank_check(m, wl, out, B, device, world, rank)
    if dist_on and not args.no_extra and args.workload == "dentate_b4096_euler100":
        # N > 1: the strong-scaling leg north_star names (parse1m, 8192 cells global = 8192 / N per GPU), every rank takes part
        w2 = dict(WORKLOADS["parse1m_b8192_euler100_strong"])
        if args.global_cells:
            w2["B"] = args.global_cells
        m2 = make_model(w2, args.precision, device)
        d2, _, out2, (_, _, _, B2) = time_workload(m2, w2, device, 1, 1, dist_on, world, rank, time_blocks=False)
        result["strong_scaling"] = {"workload": "parse1m_b8192_euler100_strong", "global_cells": w2["B"], "cells_per_gpu": B2,
                                    "cells_per_gpu_by_rank": [shard_cells(w2, world, r)[0] for r in range(world)],
                                    "gathered_rows": int(out2.shape[0]) if out2 is not None else None,
                                    "cells_per_s": w2["B"] / d2, "ms_per_step": 1e3 * d2, "scaling": "strong",
                                    "cross_rank_check": cross_rank_check(m2, w2, out2, B2, device, world, rank)}
        if not fake:
            # BASELINE configs[4] under the same launch: the data-parallel training step (bucketed gradient all-reduce overlapped with
            # the backward) on the base shape and on the DiT-L shape, 1 024 / 256 cells per GPU
            del m2
            torch.cuda.empty_cache()
            for key, name, st_, wu_ in (("training_step", "replogle_train_b1024", 10, 5), ("training_step_ditl", "replogle_train_ditl_b256", 4, 2)):
                tw = dict(TRAIN_WORKLOADS[name])
                dtt, _ = time_training(tw, "bf16", device, st_, wu_, dist_on, world)
                fl = 3 * dit_flops(**{k: v for k, v in tw.get("shape", {}).items() if k != "n_head"})
                result[key] = {"workload": name, "cells_per_s": world * tw["B"] / (dtt / st_), "ms_per_step": 1e3 * dtt / st_,
                               "tflops_per_gpu": fl * tw["B"] / (dtt / st_) / 1e12, "dtype": "bf16", "data_parallel": time_training.info}
                torch.cuda.empty_cache()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    emit(result, rank)


if __name__ == "__main__":
    main()
