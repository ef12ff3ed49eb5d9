"""Entropic OT (Sinkhorn) for the Wasserstein generation metrics (reference src/scldm/evaluations.py:85-108 -> third-party POT,
not vendored: parity unpinned).  CPU part: the oracle's restatement against closed-form cases.  GPU part: the HIP iteration
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
