"""CPU restatement of the generation-evaluation MMD kernels.  TEST INFRASTRUCTURE ONLY.

src/scldm/evaluations.py: RBFKernel :10-21, BrayCurtisKernel :24-37, TanimotoKernel :40-53, RuzickaKernel :56-69,
MMDLoss :72-82.  Written pair-by-pair in float64-free plain torch fp32 (row blocks instead of the (Bx,By,D) broadcast, same
formulas).  Pinned against matrices and MMD values produced by the reference classes (tests/golden/mmd_*.npz).
`wasserstein` (:85-108) delegates to third-party POT (`ot.emd2` / `ot.sinkhorn2`, unpinned `pot` in pyproject): not restated.
"""
from __future__ import annotations

import torch


def kernel_matrix(kind: str, x: torch.Tensor, y: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    if kind == "rbf":
        xn, yn = (x ** 2).sum(1, keepdim=True), (y ** 2).sum(1, keepdim=True)
        return torch.exp(-scale * (xn - 2 * x @ y.T + yn.T))
    rows = []
    for i in range(x.shape[0]):          # one row of pairs at a time: (By, D) temporaries only
        xi = x[i:i + 1]
        if kind == "braycurtis":
            rows.append(1 - (xi - y).abs().sum(1) / ((xi + y).abs().sum(1) + 1e-8))
        elif kind == "tanimoto":
            rows.append((xi * y).sum(1) / ((xi + y - xi * y).sum(1) + 1e-8))
        elif kind == "ruzicka":
            rows.append(torch.minimum(xi, y).sum(1) / (torch.maximum(xi, y).sum(1) + 1e-8))
        else:
            raise ValueError(kind)
    return torch.stack(rows)


def mmd(kind: str, x: torch.Tensor, y: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    k = lambda a, b: kernel_matrix(kind, a, b, scale)
    return k(x, x).mean() + k(y, y).mean() - 2 * k(x, y).mean()
