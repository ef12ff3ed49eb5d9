"""Entropic OT (Sinkhorn) for the Wasserstein generation metrics (reference src/scldm/evaluations.py:85-108 -> third-party POT,
not vendored: parity unpinned).  CPU part: the oracle's restatement against closed-form cases and against an independent SciPy
solve of the same published optimisation problem.  GPU part: the HIP iteration
against the oracle."""
import math

import numpy as np
import pytest
import torch

from oracle.evaluations import wasserstein_sinkhorn


def test_oracle_identical_clouds_and_translation():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(40, 5, generator=g)
    d, it, status = wasserstein_sinkhorn(x, x.clone(), reg=0.05, power=2, num_iter_max=3000)
    assert status in (0, 1) and d < 0.35                 # entropic blur only: far below the cloud's diameter (~3)
    shift = torch.tensor([10.0, 0, 0, 0, 0])
    d2, _, st2 = wasserstein_sinkhorn(x, x + shift, reg=5.0, power=2, num_iter_max=3000)
    assert st2 in (0, 1) and abs(d2 - 10.0) < 0.5        # W2 between a cloud and its translate is the shift length
    d1, _, st1 = wasserstein_sinkhorn(x, x + shift, reg=1.0, power=1, num_iter_max=3000)
    assert st1 in (0, 1) and abs(d1 - 10.0) < 0.5


def test_oracle_two_point_closed_form():
    """n = m = 2 on a line: the optimal plan is the monotone matching; with small reg the entropic cost approaches it."""
    x0 = torch.tensor([[0.0], [1.0]])
    x1 = torch.tensor([[0.5], [3.0]])
    d, _, status = wasserstein_sinkhorn(x0, x1, reg=0.05, power=1, num_iter_max=5000)
    assert status in (0, 1) and abs(d - 0.5 * (0.5 + 2.0)) < 1e-3
    d2, _, _ = wasserstein_sinkhorn(x0, x1, reg=0.05, power=2, num_iter_max=5000)
    assert abs(d2 - math.sqrt(0.5 * (0.25 + 4.0))) < 1e-3
    d3, it3, st3 = wasserstein_sinkhorn(x0, x1, reg=0.2, power=1)      # larger reg: converges to stopThr in a few dozen iterations
    assert st3 == 0 and it3 < 200 and 1.25 <= d3 < 1.35


def test_oracle_singular_update_keeps_previous_scalings():
    g = torch.Generator().manual_seed(1)
    x0, x1 = torch.randn(8, 3, generator=g) * 50, torch.randn(8, 3, generator=g) * 50
    d, it, status = wasserstein_sinkhorn(x0, x1, reg=1e-3, power=2, dtype=torch.float32)   # exp(-M/reg) underflows: K has zero columns
    assert status == 2 and it == 0 and (math.isnan(d) or d >= 0)


@pytest.mark.parametrize("n,m,D,power,reg,scale", [(7, 9, 3, 2, 0.05, 0.3), (12, 12, 4, 1, 0.05, 0.5), (20, 11, 6, 2, 0.5, 1.0)])
def test_oracle_value_equals_an_independent_solve_of_the_entropic_problem(n, m, D, power, reg, scale):
    """POT is absent; the PROBLEM `ot.sinkhorn2` solves is published (Cuturi 2013): min_P <P, M> - reg H(P) over the couplings of
    (a, b), returned as <P*, M>.  SciPy solves its concave semi-dual here with L-BFGS in float64 (no Sinkhorn iteration involved):
    the restated iteration's value must be that optimum's transport cost; and for n = m uniform weights the reg -> 0 limit is the
    assignment problem `scipy.optimize.linear_sum_assignment` solves exactly (`ot.emd2`'s value)."""
    from scipy.optimize import linear_sum_assignment, minimize
    from scipy.special import logsumexp
    g = torch.Generator().manual_seed(n * 100 + m)
    x0, x1 = torch.randn(n, D, generator=g) * scale, torch.randn(m, D, generator=g) * scale + 0.2
    M = torch.cdist(x0.double(), x1.double()).numpy() ** power
    a, b = np.full(n, 1.0 / n), np.full(m, 1.0 / m)

    def neg_semidual(f):
        lse = logsumexp((f[:, None] - M) / reg, axis=0)                      # (m,)
        gdual = reg * (np.log(b) - lse)
        P = np.exp((f[:, None] + gdual[None, :] - M) / reg)                   # columns sum to b by construction
        return -(a @ f + b @ gdual), -(a - P.sum(1))
    r = minimize(neg_semidual, np.zeros(n), jac=True, method="L-BFGS-B", options={"maxiter": 20000, "ftol": 1e-16, "gtol": 1e-12})
    f = r.x
    gdual = reg * (np.log(b) - logsumexp((f[:, None] - M) / reg, axis=0))
    P = np.exp((f[:, None] + gdual[None, :] - M) / reg)
    assert np.abs(P.sum(1) - a).max() < 1e-8 and np.abs(P.sum(0) - b).max() < 1e-12
    cost = float((P * M).sum())
    d, it, status = wasserstein_sinkhorn(x0, x1, reg=reg, power=power)
    assert status == 0
    assert abs(d - (math.sqrt(cost) if power == 2 else cost)) < 2e-6 * max(1.0, d)
    if n == m:
        ri, ci = linear_sum_assignment(M)
        emd = float(M[ri, ci].mean())
        d_small, _, st = wasserstein_sinkhorn(x0, x1, reg=0.002 * float(M.max()), power=power, num_iter_max=200_000)
        val = d_small ** 2 if power == 2 else d_small
        assert st in (0, 1) and emd - 1e-9 <= val <= emd * 1.05               # entropic plans are feasible couplings: cost >= the optimum


@pytest.mark.gpu
@pytest.mark.parametrize("n,m,D,power,reg", [(64, 64, 8, 2, 1.0), (100, 37, 16, 1, 0.5), (257, 300, 50, 2, 5.0), (33, 65, 3, 1, 0.05)])
def test_gpu_sinkhorn_matches_oracle(n, m, D, power, reg):
    from scldm_amd.evaluations import wasserstein
    g = torch.Generator().manual_seed(n * 7 + m)
    x0 = torch.randn(n, D, generator=g)
    x1 = torch.randn(m, D, generator=g) * 1.3 + 0.4
    ref, it_ref, st_ref = wasserstein_sinkhorn(x0, x1, reg=reg, power=power, num_iter_max=3000)
    got = wasserstein(x0.cuda(), x1.cuda(), method="sinkhorn", reg=reg, power=power, num_iter_max=3000)
    stats = wasserstein.last_stats
    print(f"[parity] sinkhorn n={n} m={m} D={D} power={power} reg={reg}: hip {got:.7f} ({stats}) oracle {ref:.7f} ({it_ref} its, status {st_ref})")
    assert stats["status"] == 0 and stats["iterations"] <= 3000   # stop_thr or the fp32 floor, not the iteration limit
    assert abs(got - ref) <= 1e-4 * abs(ref) + 1e-6


@pytest.mark.gpu
def test_gpu_sinkhorn_properties_and_errors():
    from scldm_amd.evaluations import wasserstein
    g = torch.Generator().manual_seed(5)
    x = torch.randn(500, 64, generator=g).cuda()
    y = (torch.randn(400, 64, generator=g) + 0.5).cuda()
    dxy = wasserstein(x, y, method="sinkhorn", reg=10.0, power=2)
    dyx = wasserstein(y, x, method="sinkhorn", reg=10.0, power=2)
    assert abs(dxy - dyx) < 1e-4 * dxy and dxy > 0
    shift = torch.zeros(64, device="cuda"); shift[0] = 30.0
    assert abs(wasserstein(x, x + shift, method="sinkhorn", reg=20.0, power=2) - 30.0) < 1.0
    with pytest.warns(RuntimeWarning):                      # reg far too small for the cost scale: POT's "numerical errors" exit
        wasserstein(x * 100, y * 100, method="sinkhorn", reg=0.05, power=2)
    assert wasserstein.last_stats["status"] == 2
    with pytest.raises(NotImplementedError):
        wasserstein(x, y, method="emd")
    with pytest.raises(ValueError):
        wasserstein(x, y, method="other")
    with pytest.raises(RuntimeError, match="CUDA"):
        wasserstein(x.cpu(), y.cpu(), method="sinkhorn")
