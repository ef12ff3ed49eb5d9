"""Host-side vs device time of one training step (bench workload replogle_train_b1024)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench  # noqa: E402
from scldm_amd.training import train_step  # noqa: E402
from scldm_amd.transport import create_transport  # noqa: E402

wl = dict(bench.TRAIN_WORKLOADS["replogle_train_b1024"])
dev = torch.device("cuda:0")
m = bench.make_model(wl, "bf16", dev).train()
opt = torch.optim.AdamW(m.parameters(), lr=1e-4, fused=True)
tr = create_transport("Linear", "velocity", "velocity", 1e-5, 1e-5)
g = torch.Generator().manual_seed(3)
x1 = torch.randn(wl["B"], 16, 16, generator=g).to(dev)
cond = {k: torch.randint(0, v, (wl["B"],), generator=g).to(dev) for k, v in wl["vocab"].items()}
for _ in range(5):
    train_step(m, tr, opt, x1, cond)
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N):
    train_step(m, tr, opt, x1, cond)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"per step: host enqueue {1e3 * t_host / N:.3f} ms, wall {1e3 * t_all / N:.3f} ms")
if len(sys.argv) > 1:
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(N):
        train_step(m, tr, opt, x1, cond)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
