"""CPU restatement of the TransformerVAE encode/decode path (MCAB + NB head).  TEST INFRASTRUCTURE ONLY.

State-dict keys are the reference's, with the `TransformerVAE` prefixes
`encoder.`, `decoder.`, `input_layer.`, `decoder_head.` (src/scldm/vae.py:15-27).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

from .dit import layer_norm, linear, matmul, mlp, silu  # noqa: F401


@dataclass
class VAEConfig:
    """experiments/configs/model/vae_base.yaml:8-41 (shared_embedding, shared_theta, log1p)."""
    n_genes: int
    n_embed: int = 32
    n_inducing_points: int = 16
    n_embed_latent: int = 16
    n_layer: int = 8
    n_head: int = 8
    n_head_cross: int = 4
    multiple_of: int = 4
    layernorm_eps: float = 1e-8
    positional_encoding: bool = True
    nb_temperature: float = 1.0


def _ln(sd, prefix, x, eps):
    return layer_norm(x, eps, sd[f"{prefix}.weight"], sd[f"{prefix}.bias"])


def cross_attention(sd: dict, prefix: str, x: torch.Tensor, q: torch.Tensor, n_head: int) -> torch.Tensor:
    """CrossAttention.forward, src/scldm/layers.py:248-264 (k,v split order :252; bias=False)."""
    B, S, D = x.shape
    M = q.shape[1]
    hd = D // n_head
    kv = linear(x, sd[f"{prefix}.c_attn.weight"], sd.get(f"{prefix}.c_attn.bias"))
    k, v = kv.split(D, dim=-1)
    qq = linear(q, sd[f"{prefix}.c_attn_q.weight"], sd.get(f"{prefix}.c_attn_q.bias"))
    k = k.view(B, S, n_head, hd).transpose(1, 2)
    v = v.view(B, S, n_head, hd).transpose(1, 2)
    qq = qq.view(B, M, n_head, hd).transpose(1, 2)
    # (through dit.matmul: under `matmul_operand_bits(10)` both attention products round their operands like every Linear - the
    # reference's TF32 arithmetic class, experiments/scripts/inference.py:26; exact otherwise)
    s = matmul(qq, k.transpose(-1, -2)) / math.sqrt(hd)
    p = torch.softmax(s, dim=-1)
    y = matmul(p, v).transpose(1, 2).reshape(B, M, D)
    return linear(y, sd[f"{prefix}.c_proj.weight"], sd.get(f"{prefix}.c_proj.bias"))


def mcab(sd: dict, prefix: str, x: torch.Tensor, q: torch.Tensor, n_head: int, eps: float) -> torch.Tensor:
    """CrossAttentionBlock.forward, non-adaLN branch, src/scldm/layers.py:325-330.
    The residual is taken from q, not x (:327)."""
    att = cross_attention(sd, f"{prefix}.attn", _ln(sd, f"{prefix}.ln_1", x, eps), _ln(sd, f"{prefix}.ln_1q", q, eps), n_head)
    y = q + att
    return y + mlp(sd, f"{prefix}.mlp", _ln(sd, f"{prefix}.ln_2", y, eps))


def plain_block(sd: dict, prefix: str, x: torch.Tensor, n_head: int, eps: float) -> torch.Tensor:
    """Block.forward, non-adaLN branch (affine LN, bias=False), src/scldm/layers.py:222-226."""
    from .dit import self_attention
    x = x + self_attention(sd, f"{prefix}.attn", _ln(sd, f"{prefix}.ln_1", x, eps), n_head, bias=False)
    return x + mlp(sd, f"{prefix}.mlp", _ln(sd, f"{prefix}.ln_2", x, eps))


def input_layer(sd: dict, counts: torch.Tensor, genes: torch.Tensor) -> torch.Tensor:
    """InputTransformerVAE.forward with agg_func=log1p, src/scldm/layers.py:28-31,111-118."""
    emb = sd["input_layer.gene_embedding.weight"][genes.long()]
    return emb * torch.log1p(counts.to(emb.dtype)).unsqueeze(-1)


def encoder(sd: dict, cfg: VAEConfig, x: torch.Tensor) -> torch.Tensor:
    """Encoder.forward, src/scldm/nnets.py:137-144."""
    B = x.shape[0]
    q = sd["encoder.ca_layer.inducing_points"].expand(B, -1, -1)
    h = mcab(sd, "encoder.ca_layer", x, q, cfg.n_head_cross, cfg.layernorm_eps)
    if cfg.positional_encoding:
        h = h + sd["encoder.pos_embed"]
    for i in range(cfg.n_layer):
        h = plain_block(sd, f"encoder.encoder_layers.{i}", h, cfg.n_head, cfg.layernorm_eps)
    h = linear(h, sd["encoder.encoder_latent_input.0.weight"], sd.get("encoder.encoder_latent_input.0.bias"))
    return layer_norm(h, cfg.layernorm_eps)


def decoder(sd: dict, cfg: VAEConfig, z: torch.Tensor, gene_emb: torch.Tensor) -> torch.Tensor:
    """Decoder.forward with shared_embedding (gene_embedding = Identity), use_adaln=False,
    src/scldm/nnets.py:200-208."""
    h = layer_norm(z, cfg.layernorm_eps)
    h = linear(h, sd["decoder.decoder_latent_input.1.weight"], sd.get("decoder.decoder_latent_input.1.bias"))
    for i in range(cfg.n_layer):
        h = plain_block(sd, f"decoder.decoder_layers.{i}", h, cfg.n_head, cfg.layernorm_eps)
    return mcab(sd, "decoder.decoder_cross_attention", h, gene_emb, cfg.n_head_cross, cfg.layernorm_eps)


def nb_head(sd: dict, cfg: VAEConfig, h: torch.Tensor, genes: torch.Tensor, library_size: torch.Tensor):
    """NegativeBinomialTransformerLayer.forward, both theta variants, src/scldm/stochastic_layers.py:102-116."""
    if "decoder_head.theta.weight" in sd:
        mu = linear(h, sd["decoder_head.params.weight"], sd["decoder_head.params.bias"]).squeeze(-1)
        theta = torch.exp(sd["decoder_head.theta.weight"][genes.long()]).squeeze(-1)
    else:   # shared_theta=False: params is Linear(n_embed, 2), theta = exp of its second output (stochastic_layers.py:94-96,109-112)
        mu, theta = torch.chunk(linear(h, sd["decoder_head.params.weight"], sd["decoder_head.params.bias"]), 2, dim=-1)
        mu, theta = mu.squeeze(-1), torch.exp(theta).squeeze(-1)
    mu = torch.softmax(mu / cfg.nb_temperature, dim=1) * library_size
    return mu, theta


def encode(sd: dict, cfg: VAEConfig, counts: torch.Tensor, genes: torch.Tensor) -> torch.Tensor:
    """TransformerVAE.encode, src/scldm/vae.py:58-69."""
    return encoder(sd, cfg, input_layer(sd, counts, genes))


def decode(sd: dict, cfg: VAEConfig, z: torch.Tensor, genes: torch.Tensor, library_size: torch.Tensor):
    """TransformerVAE.decode -> (mu, theta), src/scldm/vae.py:71-87."""
    dt = sd["input_layer.gene_embedding.weight"].dtype
    gene_emb = sd["input_layer.gene_embedding.weight"][genes.long()]
    h = decoder(sd, cfg, z.to(dt), gene_emb)
    return nb_head(sd, cfg, h, genes, library_size.to(dt))
