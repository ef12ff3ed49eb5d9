import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name: str) -> dict:
    with np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


def golden_json(g: dict, key: str):
    return json.loads(str(g[key]))


def rel_err(a, b, floor: float = 1e-2) -> float:
    """Elementwise relative error with a floor, max |a-b| / max(|b|, floor*max|b|): the outputs cross zero and the fp32
    reference itself carries ~1e-6*max|b| of rounding noise, so elements below `floor` of the largest are measured against
    that floor (1 % of max|b| by default)."""
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    den = torch.clamp(b.abs(), min=floor * float(b.abs().max()))
    return float(((a - b).abs() / den).max())


def max_abs_rel(a, b) -> float:
    """max|a-b| / max|b| (scale-relative max error)."""
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).abs().max() / b.abs().max())


def check_err(a, b, tol: float, what: str = "", floor_tol: float | None = None) -> float:
    """The parity assertion of the GPU tests.  north_star's gate is "<= 1e-4 max rel-err": asserted on the scale-relative
    max error, with the floored ELEMENTWISE relative error (rel_err) computed, printed beside it (pytest -s / -rP show
    it) and asserted against `floor_tol` (default 10 x tol: an element at 1 % of the output scale may carry ten times the
    relative error of the largest one)."""
    e_scale, e_elem = max_abs_rel(a, b), rel_err(a, b)
    print(f"[parity] {what}: max|a-b|/max|b| = {e_scale:.3e}   elementwise rel (floor 1% of max) = {e_elem:.3e}   tol {tol:g}")
    assert e_scale < tol, f"{what}: scale-relative error {e_scale:.3e} >= {tol:g}"
    ft = (10 * tol if tol <= 1e-3 else float("inf")) if floor_tol is None else floor_tol   # bf16-class tolerances: reported only
    assert e_elem < ft, f"{what}: floored elementwise relative error {e_elem:.3e} >= {ft:g}"
    return e_scale


@pytest.fixture(scope="session")
def golden():
    return load_golden
