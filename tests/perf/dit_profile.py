#!/usr/bin/env python3
"""Short fused-DiT sampling run for rocprofv3 (kernel trace or PMC): the default bench workload, few evaluations.
usage: dit_profile.py [precision] [evals] [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
evals = int(sys.argv[2]) if len(sys.argv) > 2 else 6
wl = dict(bench.WORKLOADS["dentate_b4096_euler100"])
if len(sys.argv) > 3:
    wl["B"] = int(sys.argv[3])
wl["evals"] = evals
dev = torch.device("cuda", 0)
m = bench.make_model(wl, prec, dev)
z2, cond2, scales = bench.make_inputs(wl, wl["B"], dev, seed=1234)
for _ in range(2):
    out = m.sample_ode_cfg(z2, cond2, scales, evals + 1, "euler")
torch.cuda.synchronize()
print("done", float(out.abs().mean()))
