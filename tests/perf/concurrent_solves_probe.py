#!/usr/bin/env python3
"""Two fused solves of two different batches at once (two native handles over the SAME parameters, two streams) against one after the
other: what running consecutive batches of the prediction loop concurrently could give at small batch sizes.
usage: concurrent_solves_probe.py [cells=128] [evals=50]"""
import copy, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from scldm_amd.sampling import sample_latents
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
evals = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda", 0)
wl = dict(bench.WORKLOADS["dentate_b128_euler50"], B=B, evals=evals)
m0 = bench.make_model(wl, "bf16", dev)
m1 = copy.copy(m0)                       # same parameters and sub-modules, its own native handle (the copy drops the handle)
assert m1.pos_embed is m0.pos_embed
g = torch.Generator().manual_seed(3)
conds = [{k: torch.randint(0, v, (B,), generator=g).to(dev) for k, v in wl["vocab"].items()} for _ in range(2)]
z0s = [torch.randn(B, 16, 16, generator=g).to(dev) for _ in range(2)]
scales = {k: wl["scale"] for k in wl["vocab"]}
s1 = torch.cuda.Stream()
def sequential():
    return [sample_latents(m0, z0s[i], conds[i], scales, evals + 1, "euler") for i in range(2)]
def concurrent():
    a = sample_latents(m0, z0s[0], conds[0], scales, evals + 1, "euler")
    s1.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s1):
        b = sample_latents(m1, z0s[1], conds[1], scales, evals + 1, "euler")
    torch.cuda.current_stream().wait_stream(s1)
    return [a, b]
ref = sequential(); torch.cuda.synchronize()
got = concurrent(); torch.cuda.synchronize()
print("bit-identical:", all(torch.equal(x, y) for x, y in zip(ref, got)))
for name, fn in (("sequential", sequential), ("concurrent", concurrent), ("sequential", sequential), ("concurrent", concurrent)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"{B} cells x {evals}: {name:10s} {1e3 * dt:7.3f} ms for two batches = {2 * B / dt:8.0f} cells/s")
