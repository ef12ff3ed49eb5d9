"""Parameter containers with the reference's module tree / state_dict keys (src/scldm/layers.py).

These classes only HOLD parameters under the names reference checkpoints use
(`blocks.i.attn.c_attn.weight`, `blocks.i.mlp.w1.weight`, `blocks.i.adaln_modulation.1.weight`, ...).
Their arithmetic runs inside the fused gfx950 kernels launched by the owning network
(`scldm_amd.nnets.DiT`, ...), so calling a container's `forward` on its own is an error.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn


def swiglu_hidden(n_embed: int, multiple_of: int) -> int:
    """Hidden width of the reference's SwiGLU MLP (layers.py:165-167): 684 at 256, 88 at 32."""
    h = int(2 * (4 * n_embed) / 3)
    return multiple_of * ((h + multiple_of - 1) // multiple_of)


def _no_dropout(dropout: float, who: str) -> None:
    """The fused kernels have no dropout stage: every reference config sets dropout 0.0 (ldm_base.yaml:21, vae_base.yaml:14).  A
    non-zero rate would silently train a different model than the reference (resid_dropout, layers.py:140-157,246-262)."""
    if float(dropout) != 0.0:
        raise NotImplementedError(f"{who}: dropout={dropout} is not supported by the fused HIP kernels (the reference configs use 0.0)")


class _Fused(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover - guard only
        raise RuntimeError(f"{type(self).__name__} is a parameter container: its math is fused into the owning "
                           "network's HIP kernels (scldm_amd.nnets); call the network instead.")


class SelfAttention(_Fused):
    """c_attn (q|k|v, split order q,k,v) and c_proj - layers.py:121-141."""

    def __init__(self, n_embed: int, n_head: int, dropout: float, bias: bool):
        super().__init__()
        assert n_embed % n_head == 0
        _no_dropout(dropout, "SelfAttention")
        self.n_head, self.n_embed, self.dropout = n_head, n_embed, dropout
        self.c_attn = nn.Linear(n_embed, 3 * n_embed, bias=bias)
        self.c_proj = nn.Linear(n_embed, n_embed, bias=bias)


class CrossAttention(_Fused):
    """c_attn (k|v from x), c_attn_q, c_proj - layers.py:229-246."""

    def __init__(self, n_embed: int, n_head: int, dropout: float, bias: bool):
        super().__init__()
        _no_dropout(dropout, "CrossAttention")
        self.n_head, self.n_embed = n_head, n_embed
        self.c_attn = nn.Linear(n_embed, 2 * n_embed, bias=bias)
        self.c_attn_q = nn.Linear(n_embed, n_embed, bias=bias)
        self.c_proj = nn.Linear(n_embed, n_embed, bias=bias)


class MLP(_Fused):
    """SwiGLU: c_proj(silu(w1 x) * w2 x), no biases - layers.py:161-174."""

    def __init__(self, n_embed: int, multiple_of: int):
        super().__init__()
        hidden = swiglu_hidden(n_embed, multiple_of)
        self.w1 = nn.Linear(n_embed, hidden, bias=False)
        self.w2 = nn.Linear(n_embed, hidden, bias=False)
        self.c_proj = nn.Linear(hidden, n_embed, bias=False)


class Block(_Fused):
    """Pre-LN transformer block; adaLN-Zero variant when use_adaln - layers.py:177-206."""

    def __init__(self, n_embed: int, n_head: int, dropout: float, bias: bool, norm_layer: str, multiple_of: int,
                 layernorm_eps: float, use_adaln: bool = False, elementwise_affine: bool = True):
        super().__init__()
        if norm_layer != "layernorm":
            raise KeyError(norm_layer)
        self.ln_1 = nn.LayerNorm(n_embed, eps=layernorm_eps, elementwise_affine=elementwise_affine)
        self.ln_2 = nn.LayerNorm(n_embed, eps=layernorm_eps, elementwise_affine=elementwise_affine)
        self.attn = SelfAttention(n_embed, n_head, dropout, bias)
        self.mlp = MLP(n_embed, multiple_of)
        self.use_adaln = use_adaln
        if use_adaln:
            self.adaln_modulation = nn.Sequential(nn.SiLU(), nn.Linear(n_embed, 6 * n_embed, bias=True))


class CrossAttentionBlock(_Fused):
    """MCAB: learned inducing-point queries (encoder) or given queries (decoder) - layers.py:267-303."""

    def __init__(self, n_embed: int, n_inducing_points: int, n_head: int, dropout: float, bias: bool, norm_layer: str,
                 multiple_of: int, layernorm_eps: float, use_adaln: bool = False):
        super().__init__()
        if norm_layer != "layernorm":
            raise KeyError(norm_layer)
        if use_adaln:
            raise NotImplementedError("adaLN MCAB is not used by any reference config (vae_base.yaml: use_adaln false)")
        self.inducing_points = None if n_inducing_points == 0 else nn.Parameter(torch.randn(n_inducing_points, n_embed))
        self.ln_1 = nn.LayerNorm(n_embed, eps=layernorm_eps)
        self.ln_1q = nn.LayerNorm(n_embed, eps=layernorm_eps)
        self.attn = CrossAttention(n_embed, n_head, dropout, bias)
        self.ln_2 = nn.LayerNorm(n_embed, eps=layernorm_eps)
        self.mlp = MLP(n_embed, multiple_of)
        self.use_adaln = use_adaln

    def extra_repr(self):
        return "" if self.inducing_points is None else f"(inducing_points): Parameter(shape={tuple(self.inducing_points.shape)})"


class TimestepEmbedder(_Fused):
    """Sinusoid([cos|sin], cos first) -> Linear -> SiLU -> Linear - layers.py:339-364."""

    def __init__(self, hidden_size: int, frequency_embedding_size: int = 256):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(frequency_embedding_size, hidden_size), nn.SiLU(), nn.Linear(hidden_size, hidden_size))
        self.frequency_embedding_size = frequency_embedding_size


class FinalLayerDit(_Fused):
    """LN -> modulate(shift=chunk0, scale=chunk1) -> Linear - layers.py:388-401."""

    def __init__(self, n_embed: int, n_embed_input: int, bias: bool, layernorm_eps: float):
        super().__init__()
        self.norm_final = nn.LayerNorm(n_embed, elementwise_affine=False, eps=layernorm_eps)
        self.linear = nn.Linear(n_embed, n_embed_input, bias=bias)
        self.adaln_modulation = nn.Sequential(nn.SiLU(), nn.Linear(n_embed, 2 * n_embed, bias=bias))


class InputTransformerVAE(_Fused):
    """gene_embedding(genes) * log1p(counts) - layers.py:97-118 (agg_func log1p only)."""

    def __init__(self, n_genes: int, n_embed: int, agg_func: str = "log1p"):
        super().__init__()
        if agg_func != "log1p":
            raise NotImplementedError(f"agg_func={agg_func!r}: only 'log1p' is used by the reference configs (vae_base.yaml:39)")
        self.gene_embedding = nn.Embedding(n_genes + 1, n_embed)
        self.agg_func = agg_func


def sincos_pos_embed(embed_dim: int, seq_len: int) -> np.ndarray:
    """1-D sin|cos table (sin FIRST) - layers.py:367-385."""
    assert embed_dim % 2 == 0
    omega = 1.0 / (10000 ** (np.arange(embed_dim // 2, dtype=np.float32) / (embed_dim / 2.0)))
    ang = np.arange(seq_len, dtype=np.float32)[:, None] * omega[None, :]
    return np.concatenate([np.sin(ang), np.cos(ang)], axis=1)
