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


def rel_err(a, b, floor: float = 1e-3) -> float:
    """max |a-b| / max(|b|, floor*max|b|): relative error with a floor, since outputs cross zero."""
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    den = torch.clamp(b.abs(), min=floor * float(b.abs().max()))
    return float(((a - b).abs() / den).max())


def max_abs_rel(a, b) -> float:
    """max|a-b| / max|b| (scale-relative max error)."""
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).abs().max() / b.abs().max())


@pytest.fixture(scope="session")
def golden():
    return load_golden
