#!/usr/bin/env python3
"""bf16-operand TransformerVAE.encode, one synchronised call at a time (median of 9), at 1 024 and 4 096 cells of the dentate_gyrus
shape; SCLDM_ENC_WAVES (4 | 6 | 8 waves per cell in the pooling kernel) is read by the library once per process."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from bench import make_vae
n_genes, S = 17002, 6147
dev = torch.device("cuda")
vae = make_vae(n_genes, dev)
for prec in ("bf16", "fp32"):
    vae.precision = prec
    for B in (1024, 4096):
        g = torch.Generator().manual_seed(11)
        genes = torch.stack([torch.randperm(n_genes, generator=g)[:S] for _ in range(8)]).repeat(B // 8, 1).to(dev)
        counts = torch.poisson(torch.full((B, S), 1.5), generator=g).to(dev)
        for _ in range(3):
            z = vae.encode(counts, genes)
        torch.cuda.synchronize()
        ts = []
        for _ in range(9):
            t0 = time.perf_counter(); z = vae.encode(counts, genes); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        dt = statistics.median(ts)
        print(f"SCLDM_ENC_WAVES={os.environ.get('SCLDM_ENC_WAVES', '4')} {prec} B={B}: {1e3 * dt:.3f} ms = {B / dt / 1e6:.3f} M cells/s  (z checksum {float(z.double().sum()):.6f})", flush=True)
