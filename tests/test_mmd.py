"""MMD evaluation kernels: oracle vs the reference classes' outputs (CPU), HIP kernels vs both (GPU)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, max_abs_rel
from oracle.evaluations import kernel_matrix, mmd

CASES = [("rbf", "rbf", "z", 1.0), ("rbf_s", "rbf", "z", 0.37), ("braycurtis", "braycurtis", "l", 1.0),
         ("braycurtis_signed", "braycurtis", "z", 1.0), ("tanimoto", "tanimoto", "c", 1.0), ("ruzicka", "ruzicka", "l", 1.0),
         ("ruzicka_signed", "ruzicka", "z", 1.0)]
TOL = 1e-4


def _mmd_tol(g, tag):
    return TOL * float(np.abs(g[f"k_{tag}"]).mean())    # MMD is a difference of kernel means: tolerance on that scale


@pytest.mark.parametrize("name", ["mmd_small", "mmd_counts"])
@pytest.mark.parametrize("tag,kind,data,scale", CASES)
def test_oracle_matches_reference(name, tag, kind, data, scale):
    g = load_golden(name)
    x, y = torch.from_numpy(g[data + "x"]), torch.from_numpy(g[data + "y"])
    assert max_abs_rel(kernel_matrix(kind, x, y, scale), g[f"k_{tag}"]) < 2e-5
    assert abs(float(mmd(kind, x, y, scale)) - float(g[f"mmd_{tag}"])) <= _mmd_tol(g, tag)


def _hip_kernel(kind, scale):
    from scldm_amd import evaluations as ev
    return {"rbf": ev.RBFKernel(scale), "braycurtis": ev.BrayCurtisKernel(), "tanimoto": ev.TanimotoKernel(), "ruzicka": ev.RuzickaKernel()}[kind]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mmd_small", "mmd_counts"])
@pytest.mark.parametrize("tag,kind,data,scale", CASES)
def test_hip_matches_reference_golden(name, tag, kind, data, scale):
    from scldm_amd.evaluations import MMDLoss
    g = load_golden(name)
    x, y = torch.from_numpy(g[data + "x"]).cuda(), torch.from_numpy(g[data + "y"]).cuda()
    k = _hip_kernel(kind, scale)
    assert max_abs_rel(k(x, y).cpu(), g[f"k_{tag}"]) < TOL
    assert abs(float(MMDLoss(k)(x, y)) - float(g[f"mmd_{tag}"])) <= _mmd_tol(g, tag)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["rbf", "braycurtis", "tanimoto", "ruzicka"])
def test_hip_properties_at_evaluation_size(kind):
    """2 048 x 1 536 cells x 17 002 genes (the reference would allocate a 214 GB broadcast tensor per term): the fused sums
    equal the mean of the explicitly written matrix, k(x, x) is symmetric with the right diagonal, MMD(x, x) = 0, and a
    random 64 x 64 block agrees with the oracle."""
    from scldm_amd.evaluations import MMDLoss
    gen = torch.Generator(device="cuda").manual_seed(11)
    nx, ny, D = 2048, 1536, 17002
    x = (torch.poisson(torch.full((nx, D), 0.9, device="cuda"), generator=gen) * (torch.rand((nx, D), device="cuda", generator=gen) < 0.2)).float()
    y = (torch.poisson(torch.full((ny, D), 1.1, device="cuda"), generator=gen) * (torch.rand((ny, D), device="cuda", generator=gen) < 0.2)).float()
    if kind != "tanimoto":
        x, y = torch.log1p(x / x.sum(1, keepdim=True).clamp(min=1) * 1e4), torch.log1p(y / y.sum(1, keepdim=True).clamp(min=1) * 1e4)
    scale = 1e-3 if kind == "rbf" else 1.0
    k = _hip_kernel(kind, scale)
    kxy = k(x, y)
    assert abs(float(k.mean(x, y)) - float(kxy.double().mean())) <= 1e-5 * float(kxy.abs().mean())
    kxx = k(x, x)
    assert max_abs_rel(kxx.T.cpu(), kxx.cpu()) < 1e-5
    if kind != "tanimoto":
        assert float((kxx.diagonal() - 1).abs().max()) < 1e-4      # k(x, x) = 1 for RBF / Bray-Curtis / Ruzicka
    assert abs(float(MMDLoss(k)(x, x))) <= 1e-5 * float(kxx.abs().mean())
    bi, bj = 640, 1024
    ref = kernel_matrix(kind, x[bi:bi + 64].cpu(), y[bj:bj + 64].cpu(), scale)
    assert max_abs_rel(kxy[bi:bi + 64, bj:bj + 64].cpu(), ref) < TOL
