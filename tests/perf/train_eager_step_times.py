#!/usr/bin/env python3
"""Wall time of each of the first N eager train_step() calls (synchronised per step): where a warm-up cost hides.
usage: train_eager_step_times.py [steps=40] [optimizer=native|torch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench
from scldm_amd.training import train_step
from scldm_amd.transport import create_transport
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
kind = sys.argv[2] if len(sys.argv) > 2 else "native"
dev = torch.device("cuda:0")
wl = dict(bench.TRAIN_WORKLOADS["replogle_train_b1024"])
m = bench.make_model(wl, "bf16", dev).train()
opt = bench.make_optimizer(m.parameters(), 1e-4, kind)
tr = create_transport("Linear", "velocity", "velocity", 1e-5, 1e-5)
g = torch.Generator().manual_seed(3)
x1 = torch.randn(wl["B"], 16, 16, generator=g).to(dev)
cond = {k: torch.randint(0, v, (wl["B"],), generator=g).to(dev) for k, v in wl["vocab"].items()}
ts = []
for i in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    train_step(m, tr, opt, x1, cond)
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print(kind, " ".join(f"{t:.2f}" for t in ts))
