#!/usr/bin/env python3
"""Run the fused training step several times on the same inputs and report which gradient tensors differ between runs (a race
detector for dit_backward_kernel / wgrad), and their error against the first run.  usage: tools/train_determinism.py [cells] [precision] [layers]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_train as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
layers = int(sys.argv[3]) if len(sys.argv) > 3 else 8
vocab = {"cell_line": 4, "gene": 2024}
m, sd, cfg = T.build(vocab, "joint", layers, 81)
m.precision = prec
gen = torch.Generator().manual_seed(9)
x1, x0 = torch.randn(n, 16, 16, generator=gen), torch.randn(n, 16, 16, generator=gen)
t = torch.rand(n, generator=gen)
cond = {k: torch.randint(0, v + 1, (n,), generator=gen) for k, v in vocab.items()}
runs = []
for _ in range(4):
    T.hip_training_step(m, x1, x0, t, cond)
    torch.cuda.synchronize()
    runs.append({k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
bad = {}
for i in range(1, 4):
    for k in runs[0]:
        d = runs[i][k] != runs[0][k]
        if bool(d.any()):
            bad.setdefault(k, []).append((int(d.sum()), float((runs[i][k] - runs[0][k]).norm() / runs[0][k].norm())))
print(f"{os.environ.get('SCLDM_LIB', 'tree')[-16:]} {n} cells {prec}: {len(bad)} of {len(runs[0])} gradient tensors differ between runs")
for k, v in sorted(bad.items())[:40]:
    print("   ", k, v)
