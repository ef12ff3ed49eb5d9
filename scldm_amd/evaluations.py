"""Generation-evaluation MMD metrics - drop-in for the classes of `scldm.evaluations` (src/scldm/evaluations.py:10-82;
SURVEY.md section 8f row N4).  Same class names, constructor arguments and call signatures: `kernel(x, y)` returns the
(Bx, By) kernel matrix, `MMDLoss(kernel)(x, y)` = mean k(x,x) + mean k(y,y) - 2 mean k(x,y).

The reference kernels broadcast to (Bx, By, D) tensors (for count matrices that is Bx*By*G floats per term); here one
fused HIP kernel per term streams D through LDS and keeps the pair accumulators in registers (scldm_mmd_kernel_sum);
MMDLoss never materialises a kernel matrix.  No CPU fallback.  `wasserstein` (third-party POT) is not provided.
"""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn

from . import _lib

_KIND = {"rbf": 0, "braycurtis": 1, "tanimoto": 2, "ruzicka": 3}


def _pair_sum(x: torch.Tensor, y: torch.Tensor, kind: int, scale: float, want_matrix: bool):
    if x.device.type != "cuda" or y.device != x.device:
        raise RuntimeError("scldm_amd.evaluations works on CUDA (ROCm) tensors; there is no CPU path")
    if x.dim() != 2 or y.dim() != 2 or x.shape[1] != y.shape[1]:
        raise ValueError(f"expected x (Bx,D) and y (By,D), got {tuple(x.shape)} and {tuple(y.shape)}")
    x, y = x.float().contiguous(), y.float().contiguous()
    nx, ny, D = x.shape[0], y.shape[0], x.shape[1]
    L = _lib.lib()
    ws = torch.empty(max(L.scldm_mmd_workspace_bytes(nx, ny), 256), dtype=torch.uint8, device=x.device)
    out = torch.empty(1, dtype=torch.float64, device=x.device)
    kmat = torch.empty((nx, ny), dtype=torch.float32, device=x.device) if want_matrix else None
    with torch.cuda.device(x.device):
        _lib.check(L.scldm_mmd_kernel_sum(x.data_ptr(), nx, y.data_ptr(), ny, D, kind, float(scale), out.data_ptr(),
                                          kmat.data_ptr() if kmat is not None else None, ws.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream), "scldm_mmd_kernel_sum")
    return out, kmat


class _PairKernel(nn.Module):
    kind = -1
    scale = 1.0

    def forward(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        return _pair_sum(x, y, self.kind, self.scale, True)[1]

    def mean(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        """mean_ij k(x_i, y_j) as a 0-d fp32 tensor, without the matrix."""
        return (_pair_sum(x, y, self.kind, self.scale, False)[0][0] / (x.shape[0] * y.shape[0])).float()


class RBFKernel(_PairKernel):
    kind = _KIND["rbf"]

    def __init__(self, scale: float = 1.0):
        super().__init__()
        self.scale = scale


class BrayCurtisKernel(_PairKernel):
    kind = _KIND["braycurtis"]


class TanimotoKernel(_PairKernel):
    kind = _KIND["tanimoto"]


class RuzickaKernel(_PairKernel):
    kind = _KIND["ruzicka"]


class MMDLoss(nn.Module):
    def __init__(self, kernel):
        super().__init__()
        self.kernel = kernel

    def forward(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        k = self.kernel
        if isinstance(k, _PairKernel):
            return k.mean(x, x) + k.mean(y, y) - 2 * k.mean(x, y)
        return k(x, x).mean() + k(y, y).mean() - 2 * k(x, y).mean()   # any other callable kernel, as in the reference
