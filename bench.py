#!/usr/bin/env python3
"""bench.py - cells/sec of CFG flow-matching sampling with the MI355X DiT path (one JSON line on rank 0).

step     = one full sampling pass of the per-GPU batch: noise (already resident in HBM) -> final guided latents,
           n_evals DiT-with-CFG evaluations (3 sample-forwards per requested cell per evaluation), fused in one
           C call (scldm_sample_ode); for N > 1 followed by the single RCCL all-gather of the generated latents.
value    = whole-job requested cells per second = N * B_per_gpu / (time per step), max over ranks.
roofline = the fused DiT block kernel (dominant): algorithmic FLOPs per launch / mean launch duration measured
           with HIP events on the launch stream inside the timed region, against the dense MFMA peak.
cpu_baseline = the CPU oracle ("port" of the reference algorithm) on this box's host cores, bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
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
    # configs[2]: 100 Heun steps = 200 evaluations, guidance 2.0
    "hlca_b2048_heun100": dict(vocab={"cell_type": 50}, strategy="mutually_exclusive", B=2048, evals=200, method="heun", scale=2.0),
    # configs[3]: 8192 cells over 8 GPUs = 1024 per GPU, joint conditioning
    "parse1m_b1024_euler100": dict(vocab={"cell_type": 18, "cytokine": 91}, strategy="joint", B=1024, evals=100, method="euler", scale=1.0),
}
TRAIN_WORKLOADS = {
    # BASELINE.json configs[4] restated on the reference's own DiT shape (ldm_base.yaml; "DiT-L" is not a reference config):
    # replogle labels (cell_line 4, gene 2024, joint), cfg_dropout_prob 0.8, per-GPU batch 1024, AdamW, data-parallel
    "replogle_train_b1024": dict(vocab={"cell_line": 4, "gene": 2024}, strategy="joint", B=1024),
    # the same step on the DiT-L shape configs[4] names (1024 wide, 24 layers, 16 heads; not a reference config):
    # outside the fused family, so it runs on the generic GEMM-based HIP path
    "replogle_train_ditl_b256": dict(vocab={"cell_line": 4, "gene": 2024}, strategy="joint", B=256,
                                     shape=dict(n_embed=1024, n_layer=24, n_head=16)),
}


def dit_flops(n_embed=256, n_layer=8, n_embed_input=16, seq_len=16, multiple_of=4):
    """Algorithmic FLOPs (2 x MAC, GEMMs + attention contractions) of one DiT sample-forward (BASELINE.md section 3 formula)."""
    D, S = n_embed, seq_len
    H = multiple_of * ((int(2 * (4 * D) / 3) + multiple_of - 1) // multiple_of)
    block = 2 * D * 6 * D + S * (2 * D * 3 * D + 2 * D * D + 6 * D * H) + 2 * (2 * S * S * D)
    return n_layer * block + 2 * 256 * D + 2 * D * D + S * 2 * n_embed_input * D + 2 * D * 2 * D + S * 2 * D * n_embed_input
FLOPS_PER_SAMPLE_FWD = 210_763_776
FLOPS_PER_SAMPLE_BLOCK = 26_247_168 - 2 * 256 * 1536  # fused block kernel: everything of a block except the adaLN projection
PEAK = {"bf16": 2.5e15, "fp32": 157.3e12}     # dense MFMA peaks, MI355X_MICROARCH.md:41-42


def make_model(wl, precision, device, seed=0):
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


def run_steps(m, wl, z2, cond2, scales, steps, dist_on, world):
    out = None
    for _ in range(steps):
        out = m.sample_ode_cfg(z2, cond2, scales, wl["evals"] + 1 if wl["method"] == "euler" else wl["evals"] // 2 + 1, wl["method"])
        if dist_on:
            import torch.distributed as dist
            gathered = torch.empty((world * out.shape[0],) + tuple(out.shape[1:]), device=out.device, dtype=out.dtype)
            dist.all_gather_into_tensor(gathered, out)  # the single collective of the path: generated latents, 1 KB per cell
            out = gathered
    return out


def time_workload(m, wl, device, steps, warmup, dist_on, world, rank, time_blocks):
    z2, cond2, scales = make_inputs(wl, wl["B"], device, seed=1234 + rank)
    run_steps(m, wl, z2, cond2, scales, warmup, dist_on, world)
    if time_blocks:
        m.block_timing(True)
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(m, wl, z2, cond2, scales, steps, dist_on, world)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    dt = time.perf_counter() - t0
    blocks = m.block_timing() if time_blocks else None
    if time_blocks:
        m.block_timing(False)
    if dist_on:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt, blocks


def time_training(wl, precision, device, steps, warmup, dist_on, world):
    """One step = Transport.training_losses forward + HIP backward + gradient all-reduce (N > 1) + fused AdamW on the per-GPU
    batch of synthetic latents (standing in for frozen-VAE output).  Returns seconds for `steps` steps (max over ranks)."""
    from scldm_amd.training import train_step
    from scldm_amd.transport import create_transport
    m = make_model(wl, precision, device).train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, fused=True)
    tr = create_transport("Linear", "velocity", "velocity", 1e-5, 1e-5)
    g = torch.Generator().manual_seed(3)
    x1 = torch.randn(wl["B"], 16, 16, generator=g).to(device)
    cond = {k: torch.randint(0, v, (wl["B"],), generator=g).to(device) for k, v in wl["vocab"].items()}
    for _ in range(warmup):
        train_step(m, tr, opt, x1, cond)
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = train_step(m, tr, opt, x1, cond)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist_on:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt, float(loss)


def decode_inclusive(m, wl, device, n_genes=17002):
    """Same sampling pass followed by the MCAB decode of all 2B latents to NB parameters (dentate_gyrus gene count)."""
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
    vae = vae.to(device).eval()
    vae.precision = m.precision          # bf16 run: bf16-operand decode as well
    B = min(wl["B"], 1024)  # (2B, G) fp32 mu + theta outputs: 2 x 139 MB at B=1024
    w2 = dict(wl); w2["B"] = B
    z2, cond2, scales = make_inputs(w2, B, device, seed=7)
    genes = torch.arange(n_genes, device=device).repeat(2 * B, 1)
    lib = torch.full((2 * B, 1), 3000.0, device=device)
    steps = w2["evals"] + 1 if w2["method"] == "euler" else w2["evals"] // 2 + 1
    def once():
        z = m.sample_ode_cfg(z2, cond2, scales, steps, w2["method"])
        return vae.decode(z, genes, lib)
    once(); torch.cuda.synchronize()
    t0 = time.perf_counter(); once(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    t1 = time.perf_counter(); vae.decode(z2, genes, lib); torch.cuda.synchronize(); dd = time.perf_counter() - t1
    return {"cells_per_gpu": B, "n_genes": n_genes, "cells_per_s": B / dt, "decode_only_cells_per_s": 2 * B / dd,
            "note": "sampling + MCAB decode of the 2B latents to (mu, theta); decode rate counts decoded rows"}


def cpu_baseline(m, wl, budget_cells=256, evals=3):
    """The CPU oracle (plain-torch fp32 restatement of the reference, validated against it in tests/) timed on this
    box's host cores over a bounded sample: `budget_cells` cells x `evals` Euler evaluations with CFG."""
    from oracle.dit import DiTConfig, dit_forward_with_cfg
    from oracle.transport import sample_ode_fixed
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    cfg = DiTConfig(class_vocab_sizes=wl["vocab"], condition_strategy=wl["strategy"])
    z2, cond2, scales = make_inputs(wl, budget_cells, "cpu", seed=99)
    cores = min(os.cpu_count() or 1, 16)  # small GEMMs: more threads than this only adds contention (measured)
    torch.set_num_threads(cores)
    f = lambda x, t: dit_forward_with_cfg(sd, cfg, x, t, cond2, scales)
    sample_ode_fixed(z2[:8].repeat(1, 1, 1), lambda x, t: dit_forward_with_cfg(sd, cfg, x, t, {k: v[:8] for k, v in cond2.items()}, scales), 2, "euler")
    t0 = time.perf_counter()
    sample_ode_fixed(z2, f, evals + 1, "euler")
    dt = time.perf_counter() - t0
    per_eval = dt / evals
    n_evals_full = wl["evals"]
    return {"value": budget_cells / (per_eval * n_evals_full), "unit": "cells/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{budget_cells} cells x {evals} of {n_evals_full} CFG evaluations (fp32, torch CPU ops), per-evaluation time "
                      f"{per_eval:.3f}s scaled to {n_evals_full} evaluations"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="dentate_b4096_euler100", choices=sorted(WORKLOADS) + sorted(TRAIN_WORKLOADS))
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=0, help="override per-GPU batch")
    ap.add_argument("--evals", type=int, default=0, help="override the number of CFG evaluations (profiling only; not a valid bench line)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the short extra-workload measurements")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"  # the env switch exercises the RCCL path on one GPU
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=device)
    if args.workload in TRAIN_WORKLOADS:   # SURVEY 8a row T1 / BASELINE configs[4]: not the headline metric, its own line
        wl = dict(TRAIN_WORKLOADS[args.workload])
        if args.batch:
            wl["B"] = args.batch
        dt, loss = time_training(wl, args.precision, device, args.steps, max(args.warmup, 1), dist_on, world)
        value = world * wl["B"] / (dt / args.steps)
        result = {"metric": "training cells/sec (flow-matching step: forward + backward + gradient all-reduce + AdamW)", "value": value,
                  "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 1), "ms_per_step": 1e3 * dt / args.steps,
                  "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
                  "config": {"workload": args.workload, "cells_per_gpu": wl["B"], "global_cells": world * wl["B"],
                             "class_vocab_sizes": wl["vocab"], "condition_strategy": wl["strategy"], "optimizer": "AdamW (fused)",
                             "parallelism": f"data-parallel x{world}, one flat-bucket all-reduce of gradients" if dist_on else "single GPU"},
                  "train_tflops_per_gpu": 3 * dit_flops(**{k: v for k, v in wl.get("shape", {}).items() if k != "n_head"}) * wl["B"]
                  / (dt / args.steps) / 1e12, "final_loss": loss}
        if "shape" in wl:
            result["config"]["dit_shape"] = wl["shape"]
        if dist_on:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            sys.stdout.flush()
            try:
                import ctypes
                ctypes.CDLL(None).fflush(None)
            except Exception:
                pass
            print(json.dumps(result), flush=True)
        return
    wl = dict(WORKLOADS[args.workload])
    if args.batch:
        wl["B"] = args.batch
    if args.evals:
        wl["evals"] = args.evals
    m = make_model(wl, args.precision, device)
    dt, blocks = time_workload(m, wl, device, args.steps, args.warmup, dist_on, world, rank, time_blocks=True)
    ms_per_step = 1e3 * dt / args.steps
    cells = world * wl["B"]
    value = cells / (dt / args.steps)
    n_fwd = (2 + n_passes(wl)) * wl["B"]
    evals = wl["evals"]
    dit_flops_step = n_fwd * FLOPS_PER_SAMPLE_FWD * evals
    result = {
        "metric": "cells/sec (whole node) @100 Euler steps; DiT-fwd MFMA util % of peak",
        "value": value, "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.precision, "data": "synthetic",
        "config": {"workload": args.workload, "cells_per_gpu": wl["B"], "global_cells": cells, "cfg_evaluations": evals,
                   "method": wl["method"], "guidance_scale": wl["scale"], "condition_strategy": wl["strategy"],
                   "class_vocab_sizes": wl["vocab"], "sample_forwards_per_evaluation_per_gpu": n_fwd,
                   "parallelism": f"batch-sharded x{world}, one all-gather of latents" if dist_on else "single GPU"},
        "dit_fwd_tflops_per_gpu": dit_flops_step / (dt / args.steps) / 1e12,
        "dit_fwd_mfma_frac": dit_flops_step / (dt / args.steps) / PEAK[args.precision],
    }
    if blocks and blocks[0] > 0:
        n_launch, tot_ms = blocks
        avg_s = tot_ms / n_launch / 1e3
        lpl = m.layers_per_launch()                      # DiT layers one fused launch runs (2 unless SCLDM_LPL=1)
        flops_launch = n_fwd * FLOPS_PER_SAMPLE_BLOCK * lpl
        ach = flops_launch / avg_s / 1e12
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_dit_forward_kernel.json")
        if args.precision == "bf16" and args.workload == "dentate_b4096_euler100" and not args.batch and os.path.exists(pmc):
            # HBM bytes per launch of this kernel from the committed rocprofv3 PMC passes of this same workload
            # (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate passes; see the file's note)
            with open(pmc) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch")
            traffic_src = "profiles/pmc_dit_forward_kernel.json"
        result["roofline"] = {"bound": "mfma", "kernel": "dit_forward_kernel", "achieved": ach, "peak": PEAK[args.precision] / 1e12,
                              "unit": "TFLOP/s", "frac": ach / (PEAK[args.precision] / 1e12), "traffic": traffic,
                              "traffic_source": traffic_src,
                              "launches": n_launch, "avg_launch_us": avg_s * 1e6, "layers_per_launch": lpl,
                              "algorithmic_flops_per_launch": flops_launch}
    if rank == 0 and not dist_on:
        if not args.no_extra:
            extra = []
            for name in ("dentate_b512_euler50", "parse1m_b1024_euler100", "hlca_b2048_heun100"):
                if name == args.workload:
                    continue
                w2 = dict(WORKLOADS[name])
                m2 = make_model(w2, args.precision, device)
                d2, _ = time_workload(m2, w2, device, 1, 1, False, 1, 0, time_blocks=False)
                nf2 = (2 + n_passes(w2)) * w2["B"]
                extra.append({"workload": name, "cells_per_s": w2["B"] / d2, "ms_per_step": 1e3 * d2,
                              "dit_fwd_mfma_frac": nf2 * FLOPS_PER_SAMPLE_FWD * w2["evals"] / d2 / PEAK[args.precision]})
                del m2
            result["other_workloads"] = extra
            result["with_vae_decode"] = decode_inclusive(m, wl, device)
            tw = dict(TRAIN_WORKLOADS["replogle_train_b1024"])
            torch.cuda.empty_cache()
            dtt, _ = time_training(tw, args.precision, device, 10, 5, False, 1)
            result["training_step"] = {"workload": "replogle_train_b1024", "cells_per_s": tw["B"] / (dtt / 10), "ms_per_step": 1e3 * dtt / 10,
                                       "tflops": 3 * FLOPS_PER_SAMPLE_FWD * tw["B"] / (dtt / 10) / 1e12, "dtype": args.precision}
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(m, wl)
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        try:  # RCCL prints its banner through C stdio: flush that buffer first so the JSON line is the last line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(result), flush=True)  # the one JSON line


if __name__ == "__main__":
    main()
