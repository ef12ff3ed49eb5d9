"""CPU restatement of the generation-evaluation MMD kernels.  TEST INFRASTRUCTURE ONLY.

src/scldm/evaluations.py: RBFKernel :10-21, BrayCurtisKernel :24-37, TanimotoKernel :40-53, RuzickaKernel :56-69,
MMDLoss :72-82.  Written pair-by-pair in float64-free plain torch fp32 (row blocks instead of the (Bx,By,D) broadcast, same
formulas).  Pinned against matrices and MMD values produced by the reference classes (tests/golden/mmd_*.npz).
`wasserstein` (:85-108) delegates to third-party POT (`ot.emd2` / `ot.sinkhorn2`; `pot` is unpinned in pyproject.toml and not
installed here): PARITY UNPINNED.  `wasserstein_sinkhorn` below restates the PUBLISHED algorithm behind the call the reference
makes (models.py:47-48: method="sinkhorn", power 1 / 2, reg 0.05): Sinkhorn-Knopp matrix scaling (Cuturi, NIPS 2013) with the
iteration order, the every-10th-iteration marginal-error test (stopThr 1e-9) and the keep-previous-scalings exit on a singular
update that POT's `sinkhorn_knopp` documents; anchored on closed-form cases in tests/test_wasserstein.py and on an independent
SciPy solve (L-BFGS on the semi-dual, `linear_sum_assignment` for the reg -> 0 limit) of the optimisation problem the call is
published to solve: the VALUE is third-party-checked, POT's iteration count / stopping details stay unpinned.
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


def wasserstein_sinkhorn(x0: torch.Tensor, x1: torch.Tensor, reg: float = 0.05, power: int = 2, num_iter_max: int = 10_000_000,
                         stop_thr: float = 1e-9, dtype=torch.float64):
    """evaluations.py:85-108 with method="sinkhorn": returns (distance, iterations, status) - status 0 converged, 1 iteration
    limit, 2 singular update (previous scalings kept).  float64 by default (the checker), `dtype=torch.float32` mimics the reference."""
    assert power in (1, 2)
    x0, x1 = x0.to(dtype), x1.to(dtype)
    n, m = x0.shape[0], x1.shape[0]
    a = torch.full((n,), 1.0 / n, dtype=dtype)
    b = torch.full((m,), 1.0 / m, dtype=dtype)
    M = torch.cdist(x0, x1)
    if power == 2:
        M = M ** 2
    K = torch.exp(M / (-reg))
    u, v = a.clone(), b.clone()
    status, it = 1, 0
    while it < num_iter_max:
        ktu = K.T @ u
        v_new = b / ktu
        u_new = a / (K @ v_new)
        if bool((ktu == 0).any()) or not bool(torch.isfinite(u_new).all()) or not bool(torch.isfinite(v_new).all()):
            status = 2
            break
        u, v = u_new, v_new
        if it % 10 == 0:
            err = torch.linalg.norm(v * (K.T @ u) - b)
            if float(err) < stop_thr:
                it += 1
                status = 0
                break
        it += 1
    cost = float((u[:, None] * K * v[None, :] * M).sum())
    return (cost ** 0.5 if power == 2 else cost), it, status
