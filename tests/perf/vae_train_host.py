"""Host (enqueue) time of one VAE training step by phase, against the step's wall time: is a batch-32 step host- or device-bound?
usage: vae_train_host.py [B]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from test_abi_cpu import _build_vae
from scldm_amd.distributions import log_nb_positive
G, S = 17002, 6147
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
vae = _build_vae(G).cuda().train()
rng = np.random.default_rng(5)
counts = torch.from_numpy(rng.poisson(0.5, (B, G)).astype(np.float32)).cuda()
genes = torch.arange(G).repeat(B, 1).cuda()
gs = torch.from_numpy(np.stack([np.sort(rng.permutation(G)[:S]) for _ in range(B)])).cuda()
cs = counts.gather(1, gs)
lib = counts.sum(1, keepdim=True)
opt = torch.optim.AdamW(vae.parameters(), lr=1e-3, fused=True)
ph = np.zeros(5)
def step(rec):
    t = [time.perf_counter()]
    opt.zero_grad(set_to_none=True); t.append(time.perf_counter())
    params, z = vae(counts, genes, lib, cs, gs); t.append(time.perf_counter())
    loss = (-log_nb_positive(counts, params["mu"], params["theta"])).sum(1).mean(); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    if rec: ph[:] += np.diff(t)
for _ in range(5): step(False)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n): step(True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B}: host enqueue {1e3*(t1-t0)/n:.2f} ms/step (zero_grad {1e3*ph[0]/n:.2f}, forward {1e3*ph[1]/n:.2f}, loss {1e3*ph[2]/n:.2f}, backward {1e3*ph[3]/n:.2f}, "
      f"optimizer {1e3*ph[4]/n:.2f}); wall {1e3*(t2-t0)/n:.2f} ms/step")
