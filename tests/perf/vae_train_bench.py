"""VAE training step timing (forward + NB loss + HIP backward + AdamW) at the dentate shape.  usage: vae_train_bench.py [B ...]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from test_abi_cpu import _build_vae
from scldm_amd.distributions import log_nb_positive
G, S = 17002, 6147
for B in [int(a) for a in sys.argv[1:]] or [32, 128]:
    vae = _build_vae(G).cuda().train()
    rng = np.random.default_rng(5)
    counts = torch.from_numpy(rng.poisson(0.5, (B, G)).astype(np.float32)).cuda()
    genes = torch.arange(G).repeat(B, 1).cuda()
    gs = torch.from_numpy(np.stack([np.sort(rng.permutation(G)[:S]) for _ in range(B)])).cuda()
    cs = counts.gather(1, gs)
    lib = counts.sum(1, keepdim=True)
    opt = torch.optim.AdamW(vae.parameters(), lr=1e-3, fused=True)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    def step(rec=False):
        opt.zero_grad(set_to_none=True)
        if rec: ev[0].record()
        params, z = vae(counts, genes, lib, cs, gs)
        loss = (-log_nb_positive(counts, params["mu"], params["theta"])).sum(1).mean()
        if rec: ev[1].record()
        loss.backward()
        if rec: ev[2].record()
        opt.step()
        if rec: ev[3].record()
        return loss
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n): loss = step(True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"B={B}: {1e3*dt:.2f} ms/step ({B/dt:.0f} cells/s)  last step device: fwd+loss {ev[0].elapsed_time(ev[1]):.2f} bwd {ev[1].elapsed_time(ev[2]):.2f} opt {ev[2].elapsed_time(ev[3]):.2f} ms  loss {float(loss):.1f}")
