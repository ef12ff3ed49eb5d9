import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from scldm_amd.training import train_step
from scldm_amd.transport import create_transport
dev = torch.device("cuda:0")
wl = dict(bench.TRAIN_WORKLOADS["replogle_train_ditl_b1024"]); wl["B"] = 512
m = bench.make_model(wl, "bf16", dev).train()
with torch.no_grad():
    for p in m.parameters():
        if p.dim() == 2: p.mul_(0.5)
opt = torch.optim.AdamW(m.parameters(), lr=2e-5, fused=True)
tr = create_transport("Linear", "velocity", "velocity", 1e-5, 1e-5)
g = torch.Generator().manual_seed(3)
x1 = torch.randn(wl["B"], 16, 16, generator=g).to(dev)
cond = {k: torch.randint(0, v, (wl["B"],), generator=g).to(dev) for k, v in wl["vocab"].items()}
losses = []
for i in range(60):
    losses.append(float(train_step(m, tr, opt, x1, cond)))
print("first 5", [round(l, 4) for l in losses[:5]]); print("last 5", [round(l, 4) for l in losses[-5:]])
assert all(l == l and l < 1e4 for l in losses), "non-finite loss"
assert sum(losses[-10:]) / 10 < sum(losses[:10]) / 10, "loss did not decrease"
print("soak OK: mean first 10", sum(losses[:10]) / 10, "mean last 10", sum(losses[-10:]) / 10)
