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


def _eager_log_nb(x, mu, theta, eps, log_fn, lgamma_fn):
    """The formula of distributions.py:33-42 in torch ops (custom log / lgamma callables)."""
    log_theta_mu_eps = log_fn(theta + mu + eps)
    return (theta * (log_fn(theta + eps) - log_theta_mu_eps) + x * (log_fn(mu + eps) - log_theta_mu_eps)
            + lgamma_fn(x + theta) - lgamma_fn(theta) - lgamma_fn(x + 1))


def log_nb_positive(x: torch.Tensor, mu: torch.Tensor, theta: torch.Tensor, eps: float = 1e-8, log_fn=torch.log,
                    lgamma_fn=torch.lgamma) -> torch.Tensor:
    """Elementwise negative-binomial log-likelihood (distributions.py:6-42, same signature and defaults).
    x, mu, theta: CUDA (ROCm) fp32 tensors; differentiable w.r.t. mu and theta.  With the default `log_fn` / `lgamma_fn`
    (torch.log / torch.lgamma, or None) this is one fused HIP kernel forward and one backward; other callables are applied through
    the reference's formula in torch ops on the same device.  The HIP path has no CPU implementation: CPU tensors raise."""
    log_fn = torch.log if log_fn is None else log_fn
    lgamma_fn = torch.lgamma if lgamma_fn is None else lgamma_fn
    if not (mu.is_cuda and theta.is_cuda and x.is_cuda):
        raise RuntimeError("log_nb_positive runs on the MI355X HIP path; there is no CPU fallback")
    if log_fn is not torch.log or lgamma_fn is not torch.lgamma:
        return _eager_log_nb(x, mu, theta, eps, log_fn, lgamma_fn)
    if x.shape != mu.shape or theta.shape != mu.shape:
        x, mu, theta = torch.broadcast_tensors(x, mu, theta)
    return _LogNB.apply(x, mu, theta, eps)
