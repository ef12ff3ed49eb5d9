"""CPU restatement of the reference DiT denoiser (eval mode).  TEST INFRASTRUCTURE ONLY.

Functional style: every function takes the reference's `state_dict` (same key
names / shapes as `scldm.nnets.DiT`, src/scldm/nnets.py:216-271) plus a small
config, so reference checkpoints and the seeded golden weights drop in unchanged.
Works in fp32 or fp64 (dtype follows the weights).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch


@dataclass
class DiTConfig:
    """Mirror of `scldm.nnets.DiT.__init__` kwargs (src/scldm/nnets.py:219-234)."""
    n_embed: int = 256
    n_embed_input: int = 16
    n_layer: int = 8
    n_head: int = 8
    seq_len: int = 16
    multiple_of: int = 4
    layernorm_eps: float = 1e-8
    class_vocab_sizes: dict = field(default_factory=dict)
    condition_strategy: str = "mutually_exclusive"
    bias: bool = True
    frequency_embedding_size: int = 256

    @property
    def hidden_dim(self) -> int:
        return mlp_hidden_dim(self.n_embed, self.multiple_of)


def mlp_hidden_dim(n_embed: int, multiple_of: int) -> int:
    """SwiGLU hidden size, src/scldm/layers.py:165-167 (684 at n_embed=256, 88 at 32)."""
    h = int(2 * (n_embed * 4) / 3)
    return multiple_of * ((h + multiple_of - 1) // multiple_of)


def silu(x: torch.Tensor) -> torch.Tensor:
    return x * torch.sigmoid(x)


def layer_norm(x: torch.Tensor, eps: float, weight=None, bias=None) -> torch.Tensor:
    """nn.LayerNorm over the last dim (biased variance)."""
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    y = xc / torch.sqrt(var + eps)
    if weight is not None:
        y = y * weight
    if bias is not None:
        y = y + bias
    return y


# ---- matmul operand precision (the reference's `torch.set_float32_matmul_precision("high")`) ---------------------------------
# experiments/scripts/inference.py:26, train_ldm.py:18, train.py:18 select "high": every fp32 matmul (nn.Linear, the q k^T and
# p v products of eager flex_attention) rounds its OPERANDS to TF32 - 10 explicit mantissa bits, fp32 exponent - and accumulates
# exact products in fp32.  `matmul_operand_bits(10)` restates that arithmetic class here (round-to-nearest-even on the 13 dropped
# bits) so that tests can state a reduced-precision tolerance in terms of the REFERENCE's own arithmetic:
# err(product path) <= 1.5 x err(this mode), both measured against the exact-fp32 oracle on the same inputs.
_OPERAND_BITS: int | None = None


class matmul_operand_bits:
    """Context manager: round both operands of every matmul in this module to `bits` explicit mantissa bits (None = exact)."""

    def __init__(self, bits: int | None):
        self.bits = bits

    def __enter__(self):
        global _OPERAND_BITS
        self.prev, _OPERAND_BITS = _OPERAND_BITS, self.bits
        return self

    def __exit__(self, *exc):
        global _OPERAND_BITS
        _OPERAND_BITS = self.prev
        return False


def round_operand(x: torch.Tensor) -> torch.Tensor:
    if _OPERAND_BITS is None:
        return x
    assert x.dtype == torch.float32, "operand rounding is defined on fp32 tensors"
    drop = 23 - _OPERAND_BITS
    i = x.contiguous().view(torch.int32)
    i = (i + ((1 << (drop - 1)) - 1) + ((i >> drop) & 1)) & ~((1 << drop) - 1)      # round to nearest, ties to even
    return i.view(torch.float32)


class _RoundedMatmul(torch.autograd.Function):
    """a @ b with both operands rounded - in the forward AND in the two backward products (d a = r(g) r(b)^T, d b = r(a)^T r(g)):
    what a TF32 matmul library does to every GEMM of a training step, so that autograd over the oracle under
    `matmul_operand_bits(10)` is the reference's training arithmetic (train_ldm.py:18), gradients included."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return round_operand(a) @ round_operand(b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        rg = round_operand(g.contiguous())
        ga = (rg @ round_operand(b).transpose(-1, -2)).sum_to_size(a.shape)
        gb = (round_operand(a).transpose(-1, -2) @ rg).sum_to_size(b.shape)
        return ga, gb


def matmul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    if _OPERAND_BITS is not None and torch.is_grad_enabled() and (a.requires_grad or b.requires_grad):
        return _RoundedMatmul.apply(a, b)
    return round_operand(a) @ round_operand(b)


def linear(x: torch.Tensor, w: torch.Tensor, b=None) -> torch.Tensor:
    y = matmul(x, w.transpose(-1, -2))
    return y if b is None else y + b


def timestep_embedding(t: torch.Tensor, dim: int, max_period: int = 10000) -> torch.Tensor:
    """src/scldm/layers.py:351-360: [cos | sin] (cos FIRST), freqs = exp(-ln(P) k / half)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def t_embedder(sd: dict, cfg: DiTConfig, t: torch.Tensor) -> torch.Tensor:
    """TimestepEmbedder.forward, src/scldm/layers.py:362-364."""
    w0 = sd["t_embedder.mlp.0.weight"]
    e = timestep_embedding(t, cfg.frequency_embedding_size).to(w0.dtype)
    h = silu(linear(e, w0, sd["t_embedder.mlp.0.bias"]))
    return linear(h, sd["t_embedder.mlp.2.weight"], sd["t_embedder.mlp.2.bias"])


def condition_embedding(sd: dict, cfg: DiTConfig, condition: dict, selected_class: int = 0):
    """DiT._get_condition_embedding in EVAL mode (no dropout), src/scldm/nnets.py:380-456.

    mutually_exclusive (:389-426): classes sorted by name; the class at position
    `selected_class` of the *available* (present-in-condition) list keeps its labels,
    every other class is replaced by its null token (= vocab size).  The reference
    draws `selected_class` with torch.randint (:395); with one available class it is 0.
    joint (:428-456): every class uses its labels (eval: no mask); sum of embeddings.
    Returns (B, n_embed) or None.
    """
    names = sorted(cfg.class_vocab_sizes.keys())
    available = [n for n in names if n in condition]
    first = next(iter(condition.values()))
    bsz = first.shape[0]
    if cfg.condition_strategy == "joint":
        if not available:
            return torch.zeros(bsz, cfg.n_embed, dtype=sd["pos_embed"].dtype)
        out = 0
        for n in names:
            out = out + sd[f"class_embeddings.{n}.weight"][condition[n].long()]
        return out
    out = 0
    for n in names:
        null = cfg.class_vocab_sizes[n]
        if n in available and available.index(n) == selected_class:
            vals = condition[n].long()
        else:
            vals = torch.full((bsz,), null, dtype=torch.long)
        out = out + sd[f"class_embeddings.{n}.weight"][vals]
    return out


def self_attention(sd: dict, prefix: str, x: torch.Tensor, n_head: int, bias: bool = True) -> torch.Tensor:
    """SelfAttention.forward, src/scldm/layers.py:143-158 (q,k,v split order :147;
    flex_attention with no mask == softmax(q k^T / sqrt(hd)) v)."""
    B, S, D = x.shape
    hd = D // n_head
    qkv = linear(x, sd[f"{prefix}.c_attn.weight"], sd.get(f"{prefix}.c_attn.bias"))
    q, k, v = qkv.split(D, dim=2)
    q = q.view(B, S, n_head, hd).transpose(1, 2)
    k = k.view(B, S, n_head, hd).transpose(1, 2)
    v = v.view(B, S, n_head, hd).transpose(1, 2)
    s = matmul(q, k.transpose(-1, -2)) / math.sqrt(hd)
    p = torch.softmax(s, dim=-1)
    y = matmul(p, v).transpose(1, 2).reshape(B, S, D)
    return linear(y, sd[f"{prefix}.c_proj.weight"], sd.get(f"{prefix}.c_proj.bias"))


def mlp(sd: dict, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """MLP.forward (SwiGLU, no bias), src/scldm/layers.py:173-174."""
    return linear(silu(linear(x, sd[f"{prefix}.w1.weight"])) * linear(x, sd[f"{prefix}.w2.weight"]),
                  sd[f"{prefix}.c_proj.weight"])


def adaln_block(sd: dict, cfg: DiTConfig, i: int, x: torch.Tensor, c: torch.Tensor, taps: dict | None = None):
    """Block.forward, adaLN branch, src/scldm/layers.py:213-221.

    NOTE the argument swap: modulate(x, shift, scale) = x*(1+scale)+shift
    (:91-94) is called as modulate(ln(x), scale_attn, shift_attn), i.e. chunk 0 (named
    shift) ACTS AS the scale and chunk 1 as the shift.
    """
    p = f"blocks.{i}"
    m = linear(silu(c), sd[f"{p}.adaln_modulation.1.weight"], sd[f"{p}.adaln_modulation.1.bias"])
    a0, a1, a2, a3, a4, a5 = m.chunk(6, dim=-1)
    h1 = layer_norm(x, cfg.layernorm_eps) * (1 + a0) + a1
    att = self_attention(sd, f"{p}.attn", h1, cfg.n_head, cfg.bias)
    x = x + a2 * att
    h2 = layer_norm(x, cfg.layernorm_eps) * (1 + a3) + a4
    ml = mlp(sd, f"{p}.mlp", h2)
    if taps is not None:
        taps[f"block{i}.mod1"] = h1
        taps[f"block{i}.attn_out"] = att
        taps[f"block{i}.x_after_attn"] = x
        taps[f"block{i}.mod2"] = h2
        taps[f"block{i}.mlp_out"] = ml
    return x + a5 * ml


def final_layer(sd: dict, cfg: DiTConfig, x: torch.Tensor, c: torch.Tensor) -> torch.Tensor:
    """FinalLayerDit.forward, src/scldm/layers.py:397-401 (conventional order:
    shift = chunk 0, scale = chunk 1)."""
    m = linear(silu(c), sd["final_layer.adaln_modulation.1.weight"], sd.get("final_layer.adaln_modulation.1.bias"))
    shift, scale = m.chunk(2, dim=-1)
    h = layer_norm(x, cfg.layernorm_eps) * (1 + scale) + shift
    return linear(h, sd["final_layer.linear.weight"], sd.get("final_layer.linear.bias"))


def dit_forward(sd: dict, cfg: DiTConfig, x: torch.Tensor, t: torch.Tensor, condition: dict,
                selected_class: int = 0, taps: dict | None = None) -> torch.Tensor:
    """DiT.forward in eval mode, src/scldm/nnets.py:273-297."""
    dt = sd["pos_embed"].dtype
    x = x.to(dt)
    c = t_embedder(sd, cfg, t).unsqueeze(1)
    ce = condition_embedding(sd, cfg, condition, selected_class)
    if ce is not None:
        c = c + ce.unsqueeze(1)
    h = linear(x, sd["input_proj.weight"], sd.get("input_proj.bias")) + sd["pos_embed"]
    if taps is not None:
        taps["c"] = c
        taps["h0"] = h
    for i in range(cfg.n_layer):
        h = adaln_block(sd, cfg, i, h, c, taps if i == 0 else None)
    if taps is not None:
        taps["h_last"] = h
    return final_layer(sd, cfg, h, c)


def dit_forward_with_cfg(sd: dict, cfg: DiTConfig, x: torch.Tensor, t: torch.Tensor,
                         condition: dict | None, cfg_scale: dict | None) -> torch.Tensor:
    """DiT.forward_with_cfg, src/scldm/nnets.py:336-378.

    x is (2B, S, C): the full batch runs with all-null labels; the second half is
    re-run with labels and blended: joint -> one pass with mean(scale); mutually
    exclusive -> one pass per class in `cfg_scale` order, each added with its scale.
    """
    n = x.shape[0]
    half = n // 2
    uncond = {k: torch.full((n,), v, dtype=torch.long) for k, v in cfg.class_vocab_sizes.items()}
    u = dit_forward(sd, cfg, x, t, uncond)
    u1, u2 = u[:half], u[half:]
    g = u2.clone()
    if condition is not None and cfg_scale is not None:
        xh, th = x[half:], t[half:]
        if cfg.condition_strategy == "joint":
            full = {k: v[half:] for k, v in condition.items()}
            cp = dit_forward(sd, cfg, xh, th, full)
            avg = sum(cfg_scale.values()) / len(cfg_scale)
            g = g + avg * (cp - u2)
        else:
            for name, scale in cfg_scale.items():
                cp = dit_forward(sd, cfg, xh, th, {name: condition[name][half:]})
                g = g + scale * (cp - u2)
    return torch.cat([u1, g], dim=0)


def dit_flops_per_sample(cfg: DiTConfig) -> int:
    """Algorithmic FLOPs (2*MAC, GEMMs + attention contractions only) of one
    sample-forward; 210 763 776 at the base config."""
    D, S, H, Din = cfg.n_embed, cfg.seq_len, cfg.hidden_dim, cfg.n_embed_input
    per_block = 2 * D * 6 * D + S * (2 * D * 3 * D + 2 * D * D + 6 * D * H) + 2 * (2 * S * S * D)
    extra = 2 * cfg.frequency_embedding_size * D + 2 * D * D + S * 2 * Din * D + 2 * D * 2 * D + S * 2 * D * Din
    return cfg.n_layer * per_block + extra
