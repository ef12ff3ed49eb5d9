"""Seeded, portable weight recipe shared by the golden generator and the tests.

The reference zero-initialises adaLN / final layers (src/scldm/nnets.py:480-492)
and never initialises Encoder.pos_embed (nnets.py:103-106), so golden vectors use
randomised values for EVERY tensor.  Values come from numpy's PCG64 so they are
identical across torch versions; tensors are filled in sorted-key order.
"""
from __future__ import annotations

import numpy as np
import torch


def _is_ln_weight(key: str) -> bool:
    leaf = key.split(".")
    return len(leaf) >= 2 and leaf[-1] == "weight" and leaf[-2].startswith("ln_")


def make_state_dict(shapes: dict[str, tuple[int, ...]], seed: int, std: float = 0.05,
                    dtype: torch.dtype = torch.float32) -> dict[str, torch.Tensor]:
    rng = np.random.default_rng(seed)
    out: dict[str, torch.Tensor] = {}
    for key in sorted(shapes):
        shape = tuple(shapes[key])
        v = rng.standard_normal(shape).astype(np.float64)
        if _is_ln_weight(key):
            v = 1.0 + std * v          # LayerNorm gains around 1
        elif key.endswith("theta.weight"):
            v = 0.3 * v                # NB inverse-dispersion log-embedding
        elif key.endswith("inducing_points") or key.endswith("gene_embedding.weight"):
            v = 1.0 * v                # reference uses randn / nn.Embedding default N(0,1)
        else:
            v = std * v
        out[key] = torch.from_numpy(v).to(dtype)
    return out


def shapes_of(module: torch.nn.Module) -> dict[str, tuple[int, ...]]:
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}
