#!/usr/bin/env python3
"""What the per-step host hand-off of FusedTrainStep costs on the device: the full call (batch copies into the static inputs + the 16-byte
hyper copy + launch) against the bare launch / graph replay on the same static inputs.  usage: train_fused_head.py [cells=1024] [steps=200]"""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from scldm_amd.ema import EMA
from scldm_amd.training import FusedTrainStep
from scldm_amd.transport import create_transport

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
wl = dict(bench.TRAIN_WORKLOADS["replogle_train_b1024"], B=B)
g = torch.Generator().manual_seed(3)
x1 = torch.randn(B, 16, 16, generator=g).to(dev)
cond = {k: torch.randint(0, v, (B,), generator=g).to(dev) for k, v in wl["vocab"].items()}
tr = create_transport("Linear", "velocity", "velocity", 1e-5, 1e-5)
for graph in (True, False):
    m = bench.make_model(wl, "bf16", dev).train()
    opt = bench.make_optimizer([p for p in m.parameters() if p.requires_grad], 1e-4, "native")
    ema = EMA(model=m, beta=0.9999, update_every=10, update_after_step=10_000)
    fs = FusedTrainStep(m, tr, opt, B, list(wl["vocab"]), ema=ema, seed=7, graph=graph)
    def full():
        fs(x1, cond); ema.update()
    def bare():
        fs.graph.replay() if graph else fs._launch()
    def hyper_only():
        opt.refresh_hyper(); bare(); ema.update()
    for name, fn in (("full call", full), ("bare launch", bare), ("hyper copy + launch", hyper_only), ("full call", full), ("bare launch", bare)):
        for _ in range(10):
            fn()
        gc.collect(); gc.freeze()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        print(f"{B} cells graph={int(graph)} {name:22s} {1e3 * (time.perf_counter() - t0) / steps:.3f} ms/step")
    del fs
