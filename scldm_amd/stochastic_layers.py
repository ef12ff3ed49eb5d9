"""Output head of the VAE with the reference's API (src/scldm/stochastic_layers.py:76-116), shared- and unshared-theta variants."""
from __future__ import annotations

import torch
import torch.nn as nn


class NegativeBinomialTransformerLayer(nn.Module):
    """Parameter container with the reference's two variants (stochastic_layers.py:89-96):
      shared_theta=True  (vae_base.yaml `negative_binomial_shared_theta`): `theta` Embedding(n_genes+1, 1) initialised to ones and
                         `params` Linear(n_embed, 1); theta = exp(theta[genes]);
      shared_theta=False (`negative_binomial_unshared_theta`): `theta` is None and `params` is Linear(n_embed, 2); theta = exp of the
                         head's second output, per cell and gene (stochastic_layers.py:109-111).
    mu = softmax_G(params(h)[..., 0] / t) * library_size.  Both are computed inside scldm_vae_decode (dec_gene_kernel); the training
    backward (TransformerVAE.forward under autograd) is built for the shared-theta head only."""

    def __init__(self, *, n_genes: int, shared_theta: bool = False, n_embed: int | None = None, norm_layer: str = "layernorm",
                 layernorm_eps: float = 1e-8, eps_: float = 1e-6, t: float = 1.0):
        super().__init__()
        self.shared_theta = shared_theta
        if shared_theta:
            self.theta = nn.Embedding(n_genes + 1, 1)
            nn.init.ones_(self.theta.weight)
            self.params = nn.Linear(n_embed, 1, bias=True)
        else:
            self.theta = None
            self.params = nn.Linear(n_embed, 2, bias=True)
        self.eps_ = eps_
        self.t = t

    def forward(self, *a, **k):  # pragma: no cover - guard only
        raise RuntimeError("NegativeBinomialTransformerLayer is fused into scldm_amd.vae.TransformerVAE.decode")


class NegativeBinomial(torch.distributions.Distribution):
    """Minimal stand-in for scvi.distributions.NegativeBinomial(mu=, theta=) as used by the reference
    (vae.py:87; models.py:819 calls .sample()): holds mu / theta, samples with the Gamma-Poisson mixture."""

    arg_constraints = {}

    def __init__(self, mu: torch.Tensor, theta: torch.Tensor, validate_args=False):
        self.mu, self.theta = mu, theta
        super().__init__(batch_shape=mu.shape, validate_args=validate_args)

    @property
    def mean(self):
        return self.mu

    @torch.no_grad()
    def sample(self, sample_shape=torch.Size(), seed: int | None = None):
        """counts ~ Poisson(Gamma(concentration=theta, rate=theta/mu)) (the parameterisation scvi-tools uses), drawn by the HIP
        kernel behind scldm_nb_sample (Philox4x32-10 + Marsaglia-Tsang + PTRS).  `seed` defaults to a draw from torch's global
        generator, so torch.manual_seed makes it reproducible.  Device tensors only: there is no CPU path."""
        import ctypes as C

        from . import _lib
        shape = self._extended_shape(sample_shape)
        mu = self.mu.expand(shape).contiguous().float()
        theta = self.theta.expand(shape).contiguous().float()
        if not mu.is_cuda:
            raise RuntimeError("NegativeBinomial.sample draws on the MI355X (scldm_nb_sample); mu / theta must be CUDA (ROCm) tensors")
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (), dtype=torch.int64).item())
        out = torch.empty_like(mu)
        with torch.cuda.device(mu.device):
            _lib.check(_lib.lib().scldm_nb_sample(mu.data_ptr(), theta.data_ptr(), out.data_ptr(), mu.numel(), C.c_uint64(seed),
                                                  torch.cuda.current_stream().cuda_stream), "scldm_nb_sample")
        return out
