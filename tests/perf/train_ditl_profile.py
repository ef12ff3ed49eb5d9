"""Profile target: a few training steps of the DiT-L workload (generic GEMM-based path)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench
dev = torch.device("cuda:0")
wl = dict(bench.TRAIN_WORKLOADS["replogle_train_ditl_b256"])
if len(sys.argv) > 1:
    wl["B"] = int(sys.argv[1])
dt, loss = bench.time_training(wl, "bf16", dev, 5, 2, False, 1)
print(f"ms/step {1e3 * dt / 5:.2f}  cells/s {wl['B'] / (dt / 5):.0f}")
