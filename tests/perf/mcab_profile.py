#!/usr/bin/env python3
"""MCAB encode OR decode at the bench shapes for rocprofv3 (kernel trace or one PMC pass): the bench's own synthetic VAE and inputs,
a few calls, nothing else.   usage: mcab_profile.py decode|decode_sample|encode [precision=fp32] [rows=8192|cells=4096] [G=17002] [S=6147] [calls=4]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

mode = sys.argv[1] if len(sys.argv) > 1 else "decode"
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
n = int(sys.argv[3]) if len(sys.argv) > 3 else (4096 if mode == "encode" else 8192)
G = int(sys.argv[4]) if len(sys.argv) > 4 else 17002
S = int(sys.argv[5]) if len(sys.argv) > 5 else 6147
calls = int(sys.argv[6]) if len(sys.argv) > 6 else 4
dev = torch.device("cuda:0")
vae = bench.make_vae(G, dev)
vae.precision = prec
g = torch.Generator().manual_seed(11)
with torch.no_grad():
    if mode == "encode":
        genes = torch.stack([torch.randperm(G, generator=g)[:S] for _ in range(8)]).repeat(n // 8, 1).to(dev)
        counts = torch.poisson(torch.full((n, S), 1.5), generator=g).to(dev)
        for _ in range(calls):
            out = vae.encode(counts, genes)
    else:
        z = torch.randn(n, 16, 16, generator=g).to(dev)
        genes = torch.arange(G, device=dev).repeat(n, 1)
        lib = torch.full((n, 1), 3000.0, device=dev)
        for _ in range(calls):
            out = vae.decode_sample(z, genes, lib, seed=1) if mode == "decode_sample" else vae.decode(z, genes, lib).mu
torch.cuda.synchronize()
print("done", mode, prec, n, float(out.float().sum()))
