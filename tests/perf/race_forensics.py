#!/usr/bin/env python3
"""For a nondeterministically corrupted sample: which (t, label) reproduces the bad output?"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import _random_dit
from oracle.dit import DiTConfig, dit_forward

n, layers = 12288, 1
m = _random_dit(n_layer=layers)
sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
cfg = DiTConfig(n_layer=layers, class_vocab_sizes={"clusters": 14})
m = m.cuda(); m.precision = "bf16"
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(n, 16, 16, device="cuda", generator=g)
t = torch.rand(n, device="cuda", generator=g)
lab = torch.randint(0, 14, (n,), device="cuda", generator=g)
ys = [m(x, t, {"clusters": lab}).clone() for _ in range(6)]
torch.cuda.synchronize()
stack = torch.stack(ys)                      # (6, n, 16, 16)
med = stack.median(dim=0).values             # majority value per element = presumably correct
bad_runs = [(r, int(b)) for r in range(6) for b in (stack[r] != med).any(dim=(1, 2)).nonzero().flatten()[:3]]
print("bad (run, sample):", bad_runs[:8])
xc, tc, lc = x.cpu(), t.cpu(), lab.cpu()
for r, b in bad_runs[:4]:
    ref_all = dit_forward(sd, cfg, xc[b:b + 1].expand(n, -1, -1).contiguous(), tc, {"clusters": lc})  # x_b with every sample's conditioning
    err = (ref_all - stack[r, b].cpu()).abs().amax(dim=(1, 2))
    good = (ref_all[b] - med[b].cpu()).abs().max()
    j = int(err.argmin())
    print(f"run {r} sample {b} (tile {b // 4}, local {b % 4}): correct-output err vs oracle {float(good):.3e}; bad output best matches conditioning of sample {j} "
          f"(tile {j // 4}, local {j % 4}) err {float(err[j]):.3e}; err with own conditioning {float(err[b]):.3e}; t_b={float(tc[b]):.4f} t_j={float(tc[j]):.4f}")
