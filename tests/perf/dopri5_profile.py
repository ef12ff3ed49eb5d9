#!/usr/bin/env python3
"""The reference's default sampler (dopri5) at the bench shape for rocprofv3 / wall timing: usage dopri5_profile.py [cells=4096] [solves=2]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from scldm_amd.transport import Sampler, create_transport
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
wl = dict(bench.WORKLOADS["dentate_b4096_euler100"], B=B)
m = bench.make_model(wl, "bf16", dev)
z2, cond2, scales = bench.make_inputs(wl, B, dev, seed=77)
fn = Sampler(create_transport()).sample_ode()
model_fn = lambda x, t, **kw: m.forward_with_cfg(x, t, **kw, cfg_scale=scales)
fn(z2, model_fn, condition=cond2)
torch.cuda.synchronize()
for _ in range(n):
    t0 = time.perf_counter()
    fn(z2, model_fn, condition=cond2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = fn.last_stats
    print(f"{B} cells: {1e3 * dt:.2f} ms per solve, {st['evaluations']} evaluations, {len(st['accepted_steps'])} accepted / {len(st['rejected_steps'])} rejected steps, "
          f"{1e3 * dt / st['evaluations']:.3f} ms per evaluation")
