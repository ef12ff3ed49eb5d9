#!/usr/bin/env python3
"""FusedTrainStep (scldm_dit_train_step as a HIP graph; optionally with the frozen VAE encode inside) for rocprofv3 / wall timing.
usage: train_fused_profile.py [cells=1024] [steps=30] [precision=bf16] [graph=1] [encode=0]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from scldm_amd.ema import EMA
from scldm_amd.training import FusedTrainStep
from scldm_amd.transport import create_transport

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"
graph = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
encode = bool(int(sys.argv[5])) if len(sys.argv) > 5 else False
dev = torch.device("cuda:0")
wl = dict(bench.TRAIN_WORKLOADS["replogle_train_b1024"], B=B)
m = bench.make_model(wl, prec, dev).train()
opt = bench.make_optimizer([p for p in m.parameters() if p.requires_grad], 1e-4, "native")
tr = create_transport("Linear", "velocity", "velocity", 1e-5, 1e-5)
ema = EMA(model=m, beta=0.9999, update_every=10, update_after_step=10_000)
g = torch.Generator().manual_seed(3)
x1 = torch.randn(B, 16, 16, generator=g).to(dev)
cond = {k: torch.randint(0, v, (B,), generator=g).to(dev) for k, v in wl["vocab"].items()}
kw = {}
if encode:
    G, S = 17002, 6147
    vae = bench.make_vae(G, dev)
    vae.precision = "fp16"
    genes = torch.stack([torch.randperm(G, generator=g)[:S] for _ in range(8)]).repeat(B // 8, 1).to(dev)
    counts = torch.poisson(torch.full((B, S), 1.5), generator=g).to(dev)
    kw = dict(vae=vae, encode_shape=(B, S))
fs = FusedTrainStep(m, tr, opt, B, list(wl["vocab"]), ema=ema, seed=7, graph=graph, **kw)
def step():
    loss = fs(condition=cond, counts_subset=counts, genes_subset=genes) if encode else fs(x1, cond)
    ema.update()
    return loss
for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = step()
torch.cuda.synchronize()
print(f"{B} cells {prec} graph={int(graph)} encode={int(encode)}: {1e3 * (time.perf_counter() - t0) / steps:.3f} ms/step   loss {float(loss):.4f}")
