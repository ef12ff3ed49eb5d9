#!/usr/bin/env python3
"""The serial end-to-end generation chain of bench.generation_end_to_end for rocprofv3: usage gen_e2e_profile.py [workload] [n_genes] [iterations=4]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from scldm_amd.datamodule import dense_to_csr, to_host
from scldm_amd.sampling import SizeFactorSampler, sample_latents
wl_name = sys.argv[1] if len(sys.argv) > 1 else "dentate_b512_euler50"
n_genes = int(sys.argv[2]) if len(sys.argv) > 2 else 17002
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda", 0)
wl = dict(bench.WORKLOADS[wl_name]); B = wl["B"]
m = bench.make_model(wl, "bf16", dev)
vae = bench.make_vae(n_genes, dev); vae.precision = "fp16"
smp = SizeFactorSampler(bench.synthetic_vocabulary_encoder(wl["vocab"], wl["strategy"]), wl["strategy"], dev)
g = torch.Generator().manual_seed(21)
cond = {k: torch.randint(0, v, (B,), generator=g).to(dev) for k, v in wl["vocab"].items()}
scales = {k: wl["scale"] for k in wl["vocab"]}
genes2 = torch.arange(n_genes, device=dev).repeat(2 * B, 1)
steps = wl["evals"] + 1
def once():
    sf = smp.sample(cond, B)
    z = sample_latents(m, torch.randn((B, 16, 16), device=dev), cond, scales, steps, wl["method"])
    lib = torch.exp(sf).view(-1, 1)
    return to_host(*dense_to_csr(vae.decode_sample(z, genes2, torch.cat([lib, lib]))), z)
once(); torch.cuda.synchronize()
for _ in range(iters):
    t0 = time.perf_counter(); once(); torch.cuda.synchronize()
    print(f"{wl_name}: {1e3 * (time.perf_counter() - t0):.3f} ms")
