#!/usr/bin/env python3
"""Short MCAB encode / decode run for rocprofv3 (kernel trace or PMC): a few calls at one size, nothing else.
usage: vae_profile.py [G] [S] [B] [precision]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_abi_cpu import _build_vae
from oracle.weights import make_state_dict

G = int(sys.argv[1]) if len(sys.argv) > 1 else 17002
S = int(sys.argv[2]) if len(sys.argv) > 2 else 6147
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
prec = sys.argv[4] if len(sys.argv) > 4 else "fp32"
vae = _build_vae(G)
vae.load_state_dict(make_state_dict({k: tuple(v.shape) for k, v in vae.state_dict().items()}, 7), strict=True)
vae = vae.cuda().eval()
vae.precision = prec
gen = torch.Generator(device="cuda").manual_seed(B)
counts = torch.poisson(torch.full((B, S), 1.5, device="cuda"), generator=gen) + 1
genes = (torch.arange(S, device="cuda").unsqueeze(0) * 2 + torch.arange(B, device="cuda").unsqueeze(1)) % G + 1
allg = torch.arange(1, G + 1, device="cuda").unsqueeze(0).expand(B, G).contiguous()
lib = counts.sum(1, keepdim=True)
with torch.no_grad():
    for _ in range(4):
        z = vae.encode(counts, genes)
        nb = vae.decode(z, allg, lib)
torch.cuda.synchronize()
print("done", float(nb.mu.sum()))
