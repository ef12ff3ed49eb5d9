#!/usr/bin/env python3
"""Time MMDLoss (three fused kernel-mean terms) at a generation-evaluation size; VALU-bound: pair-feature updates per second."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from scldm_amd import evaluations as ev
from oracle.evaluations import mmd as oracle_mmd

nx, ny, D = 2048, 2048, 17002
gen = torch.Generator(device="cuda").manual_seed(0)
x = (torch.poisson(torch.full((nx, D), 0.9, device="cuda"), generator=gen) * (torch.rand((nx, D), device="cuda", generator=gen) < 0.2)).float()
y = (torch.poisson(torch.full((ny, D), 1.1, device="cuda"), generator=gen) * (torch.rand((ny, D), device="cuda", generator=gen) < 0.2)).float()
for name, k in (("rbf", ev.RBFKernel(1e-3)), ("braycurtis", ev.BrayCurtisKernel()), ("tanimoto", ev.TanimotoKernel()), ("ruzicka", ev.RuzickaKernel())):
    loss = ev.MMDLoss(k)
    loss(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    v = loss(x, y)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    upd = (nx * nx + ny * ny + nx * ny) * D
    print(f"MMD {name}: {dt * 1e3:.1f} ms for {nx}+{ny} cells x {D} genes ({upd / dt / 1e12:.2f} T pair-feature updates/s), value {float(v):.6f}")
xs, ys = x[:128].cpu(), y[:128].cpu()
t0 = time.perf_counter()
oracle_mmd("braycurtis", xs, ys)
dt = time.perf_counter() - t0
print(f"CPU restatement (braycurtis, 128+128 cells, {torch.get_num_threads()} threads): {dt * 1e3:.0f} ms = {(3 * 128 * 128 * D) / dt / 1e12:.4f} T updates/s")
