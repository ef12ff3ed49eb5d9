#!/usr/bin/env python3
"""HBM roofline of scldm_tokenize_expressed next to the NumPy restatement of the reference's tokenizer.
Algorithmic bytes per cell: 4*G (counts in) + 12*S (int64 gene + fp32 count per window slot out) + 8 (num_expressed, library size)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from scldm_amd.datamodule import tokenize_cells_expressed
from oracle.tokenize import tokenize_expressed

for name, G, S in (("dentate_gyrus", 17002, 6147), ("hlca", 27997, 10186)):
    for N in (256, 4096):
        gen = torch.Generator(device="cuda").manual_seed(1)
        counts = (torch.poisson(torch.full((N, G), 0.8, device="cuda"), generator=gen) * (torch.rand((N, G), device="cuda", generator=gen) < 0.25)).float()
        gid = torch.randperm(G, device="cuda", generator=gen) + 1
        for _ in range(3):
            tokenize_cells_expressed(counts, gid, S, check=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            tokenize_cells_expressed(counts, gid, S, check=False)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        byts = N * (4 * G + 12 * S + 8)
        line = f"{name} N={N} G={G} S={S}: {us:.1f} us/batch, {N / us * 1e6:.3e} cells/s, {byts / us / 1e3:.0f} GB/s algorithmic ({byts / us / 1e3 / 8000:.1%} of 8 TB/s)"
        if N == 256:
            c, g = counts.cpu().numpy(), gid.cpu().numpy()
            t0 = time.perf_counter()
            tokenize_expressed(c, g, S)
            line += f"; NumPy restatement on the host: {(time.perf_counter() - t0) * 1e3:.1f} ms/batch"
        print(line)
