"""Losses with the reference's API (src/scldm/distributions.py): `log_nb_positive` as ONE HIP kernel forward and one backward."""
from __future__ import annotations

import torch

from . import _lib


class _LogNB(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mu, theta, eps):
        x, mu, theta = x.contiguous().float(), mu.contiguous().float(), theta.contiguous().float()
        out = torch.empty_like(mu)
        with torch.cuda.device(mu.device):
            _lib.check(_lib.lib().scldm_nb_loglik(x.data_ptr(), mu.data_ptr(), theta.data_ptr(), float(eps), out.data_ptr(), mu.numel(),
                                                  torch.cuda.current_stream().cuda_stream), "scldm_nb_loglik")
        ctx.save_for_backward(x, mu, theta)
        ctx.eps = float(eps)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        x, mu, theta = ctx.saved_tensors
        gout = gout.contiguous().float()
        dmu = torch.empty_like(mu) if ctx.needs_input_grad[1] else None
        dth = torch.empty_like(theta) if ctx.needs_input_grad[2] else None
        with torch.cuda.device(mu.device):
            _lib.check(_lib.lib().scldm_nb_loglik_bwd(x.data_ptr(), mu.data_ptr(), theta.data_ptr(), gout.data_ptr(), ctx.eps,
                                                      dmu.data_ptr() if dmu is not None else None, dth.data_ptr() if dth is not None else None,
                                                      mu.numel(), torch.cuda.current_stream().cuda_stream), "scldm_nb_loglik_bwd")
        return None, dmu, dth, None


def log_nb_positive(x: torch.Tensor, mu: torch.Tensor, theta: torch.Tensor, eps: float = 1e-8, log_fn=None, lgamma_fn=None) -> torch.Tensor:
    """Elementwise negative-binomial log-likelihood (distributions.py:6-42; `log_fn` / `lgamma_fn` overrides are not supported).
    x, mu, theta: same-shape CUDA (ROCm) fp32 tensors; differentiable w.r.t. mu and theta."""
    if log_fn is not None or lgamma_fn is not None:
        raise NotImplementedError("log_fn / lgamma_fn overrides have no caller in the reference")
    if not (mu.is_cuda and theta.is_cuda and x.is_cuda):
        raise RuntimeError("log_nb_positive runs on the MI355X HIP path; there is no CPU fallback")
    if x.shape != mu.shape or theta.shape != mu.shape:
        x, mu, theta = torch.broadcast_tensors(x, mu, theta)
    return _LogNB.apply(x, mu, theta, eps)
