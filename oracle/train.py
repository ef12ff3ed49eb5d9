"""Gradients of the flow-matching training loss through the oracle DiT.  TEST INFRASTRUCTURE ONLY.

The reference obtains them from torch autograd over its nn.Modules (Transport.training_losses,
src/scldm/transport/transport.py:110-150, then `loss.mean().backward()` in the Lightning step,
src/scldm/models.py:443-470).  The oracle forward (oracle/dit.py) is plain differentiable torch, so the
same autograd over it is the CPU restatement; it is pinned against gradient digests generated from the
reference (tests/golden/make_golden.py: gen_train -> tests/golden/train_*.npz).
"""
from __future__ import annotations

import numpy as np
import torch

from .dit import DiTConfig, dit_forward
from .transport import training_losses

GRAD_SAMPLE = 64
FROZEN = ("pos_embed",)  # nn.Parameter(requires_grad=False), src/scldm/nnets.py:248


def grad_digest(g) -> np.ndarray:
    """[sum, l2, GRAD_SAMPLE strided entries] of a gradient tensor (same recipe as make_golden.grad_digest)."""
    f = torch.as_tensor(g).detach().cpu().reshape(-1).to(torch.float64).numpy()
    stride = max(1, f.size // GRAD_SAMPLE)
    smp = f[::stride][:GRAD_SAMPLE]
    smp = np.pad(smp, (0, GRAD_SAMPLE - smp.size))
    return np.concatenate([[f.sum(), np.sqrt((f * f).sum())], smp])


def training_grads(sd: dict, cfg: DiTConfig, x1: torch.Tensor, x0: torch.Tensor, t: torch.Tensor, condition: dict,
                   want_dx: bool = False):
    """loss_b, pred and d mean(loss) / d parameter for every trainable state_dict entry.
    `condition` holds the labels the model actually sees (null tokens already substituted)."""
    p = {k: (v.clone().requires_grad_(k not in FROZEN)) for k, v in sd.items()}
    x1 = x1.clone().requires_grad_(want_dx)
    out = training_losses(lambda xt, tt: dit_forward(p, cfg, xt, tt, condition), x1, x0, t)
    out["loss"].mean().backward()
    grads = {k: v.grad for k, v in p.items() if v.grad is not None}
    return out["loss"].detach(), out["pred"].detach(), grads, (x1.grad if want_dx else None)
