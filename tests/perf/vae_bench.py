#!/usr/bin/env python3
"""Throughput of TransformerVAE.encode / decode (MCAB kernels, fp32-exact) at the dentate_gyrus sizes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_abi_cpu import _build_vae
from oracle.weights import make_state_dict

n_genes, G, S = 17002, 17002, 6147
vae = _build_vae(n_genes)
vae.load_state_dict(make_state_dict({k: tuple(v.shape) for k, v in vae.state_dict().items()}, 7), strict=True)
vae = vae.cuda().eval()
for B in (256, 1024):
    gen = torch.Generator(device="cuda").manual_seed(B)
    counts = torch.poisson(torch.full((B, S), 1.5, device="cuda"), generator=gen) + 1
    genes = torch.stack([torch.randperm(n_genes, device="cuda", generator=gen)[:S] + 1 for _ in range(4)]).repeat(B // 4, 1)
    allg = torch.arange(1, G + 1, device="cuda").unsqueeze(0).expand(B, G).contiguous()
    lib = counts.sum(1, keepdim=True)
    with torch.no_grad():
        for _ in range(2):
            z = vae.encode(counts, genes, counts, genes)
            nb = vae.decode(z, allg, lib)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            z = vae.encode(counts, genes, counts, genes)
        torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 5
        t0 = time.perf_counter()
        for _ in range(5):
            nb = vae.decode(z, allg, lib)
        torch.cuda.synchronize(); td = (time.perf_counter() - t0) / 5
        vae.precision = "bf16"
        for _ in range(2):
            zb = vae.encode(counts, genes, counts, genes)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            zb = vae.encode(counts, genes, counts, genes)
        torch.cuda.synchronize(); teb = (time.perf_counter() - t0) / 5
        print(f"B={B}: bf16-operand encode {teb*1e3:.2f} ms = {B/teb:.0f} cells/s")
        for _ in range(2):
            nb = vae.decode(z, allg, lib)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            nb = vae.decode(z, allg, lib)
        torch.cuda.synchronize(); tdb = (time.perf_counter() - t0) / 5
        vae.precision = "fp32"
    print(f"B={B}: bf16-operand decode {tdb*1e3:.2f} ms = {B/tdb:.0f} cells/s")
    print(f"B={B}: encode (S={S}) {te*1e3:.2f} ms = {B/te:.0f} cells/s ({41.6e6*B/te/1e12:.1f} TFLOP/s); decode (G={G}) {td*1e3:.2f} ms = {B/td:.0f} cells/s ({396.4e6*B/td/1e12:.1f} TFLOP/s)")
