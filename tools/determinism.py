#!/usr/bin/env python3
"""Run the same DiT forward several times and report where (sample / token / channel) results differ."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import _random_dit  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
layers = int(sys.argv[3]) if len(sys.argv) > 3 else 8
m = _random_dit(n_layer=layers).cuda()
m.precision = prec
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(n, 16, 16, device="cuda", generator=g)
t = torch.rand(n, device="cuda", generator=g)
lab = torch.randint(0, 14, (n,), device="cuda", generator=g)
ys = [m(x, t, {"clusters": lab}).clone() for _ in range(4)]
torch.cuda.synchronize()
for i in range(1, 4):
    d = (ys[i] != ys[0])
    ns = int(d.any(dim=(1, 2)).sum())
    print(f"run {i}: differing elements {int(d.sum())} in {ns} samples; max abs diff {float((ys[i]-ys[0]).abs().max()):.3e}")
    if ns:
        bad = d.any(dim=(1, 2)).nonzero().flatten()[:12].tolist()
        print("   first bad samples:", bad, " sample%4:", [b % 4 for b in bad])
        s0 = bad[0]
        print("   tokens with diffs in sample", s0, d[s0].any(dim=1).nonzero().flatten().tolist(), " channels:", d[s0].any(dim=0).nonzero().flatten().tolist())
