"""Gradients of the VAE training loss through the oracle TransformerVAE.  TEST INFRASTRUCTURE ONLY.

The reference obtains them from torch autograd over `TransformerVAE.forward` (src/scldm/vae.py:29-56) and
`VAE.loss` = -log_nb_positive(counts, mu, theta).sum(dim=1).mean() (src/scldm/models.py:231-249,
src/scldm/distributions.py:6-42).  The oracle forward (oracle/vae.py) is plain differentiable torch, so autograd over it is the
CPU restatement; it is pinned against gradient digests of the reference itself (tests/golden/make_golden.py: gen_vae_train ->
tests/golden/vae_train_*.npz).
"""
from __future__ import annotations

import torch

from . import vae as ov

FROZEN = ("encoder.pos_embed",)   # nn.Parameter(requires_grad=False), src/scldm/nnets.py:103-106


def log_nb_positive(x: torch.Tensor, mu: torch.Tensor, theta: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    """src/scldm/distributions.py:6-42."""
    log_theta_mu_eps = torch.log(theta + mu + eps)
    return (theta * (torch.log(theta + eps) - log_theta_mu_eps) + x * (torch.log(mu + eps) - log_theta_mu_eps)
            + torch.lgamma(x + theta) - torch.lgamma(theta) - torch.lgamma(x + 1))


def vae_forward(sd: dict, cfg: ov.VAEConfig, genes, library_size, counts_subset, genes_subset):
    """TransformerVAE.forward, src/scldm/vae.py:29-56: the encoder reads the SUBSET (:37-40), the decoder all `genes`."""
    z = ov.encode(sd, cfg, counts_subset, genes_subset)
    mu, theta = ov.decode(sd, cfg, z, genes, library_size)
    return mu, theta, z


def vae_training_grads(sd: dict, cfg: ov.VAEConfig, counts, genes, library_size, counts_subset, genes_subset, z_weight=None):
    """loss, (mu, theta, z) and d loss / d parameter for every trainable state_dict entry.  z_weight (B, 16, n_lat): adds
    sum(z * z_weight) to the loss (exercises the gradient path through the returned latent)."""
    p = {k: v.clone().requires_grad_(k not in FROZEN) for k, v in sd.items()}
    mu, theta, z = vae_forward(p, cfg, genes, library_size, counts_subset, genes_subset)
    recon = -log_nb_positive(counts, mu, theta)
    loss = recon.sum(dim=1).mean()
    if z_weight is not None:
        loss = loss + (z * z_weight).sum()
    loss.backward()
    grads = {k: v.grad for k, v in p.items() if v.grad is not None}
    return loss.detach(), (mu.detach(), theta.detach(), z.detach()), grads
