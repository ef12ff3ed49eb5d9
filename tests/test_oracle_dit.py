"""Pin the CPU oracle (oracle/dit.py) against golden vectors produced by the reference's own
DiT (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import golden_json, load_golden, max_abs_rel
from oracle.dit import DiTConfig, dit_flops_per_sample, dit_forward, dit_forward_with_cfg, layer_norm, mlp_hidden_dim
from oracle.weights import make_state_dict

CASES = ["dit_tiny", "dit_base", "dit_joint", "dit_me2", "dit_me2_256"]
TOL = 2e-5  # fp32 oracle vs fp32 reference, different summation order


def setup(name, dtype=torch.float32):
    g = load_golden(name)
    kw = golden_json(g, "kwargs_json")
    shapes = {k: tuple(v) for k, v in golden_json(g, "shapes_json").items()}
    sd = make_state_dict(shapes, int(g["seed"]), dtype=dtype)
    cfg = DiTConfig(n_embed=kw["n_embed"], n_embed_input=kw["n_embed_input"], n_layer=kw["n_layer"], n_head=kw["n_head"],
                    seq_len=kw["seq_len"], multiple_of=kw["multiple_of"], layernorm_eps=kw["layernorm_eps"],
                    class_vocab_sizes=kw["class_vocab_sizes"], condition_strategy=kw["condition_strategy"])
    return g, cfg, sd


@pytest.mark.parametrize("name", CASES)
def test_forward_matches_reference(name):
    g, cfg, sd = setup(name)
    cond = {k: torch.from_numpy(g[f"fwd_label_{k}"]) for k in golden_json(g, "fwd_classes")}
    taps = {}
    y = dit_forward(sd, cfg, torch.from_numpy(g["fwd_x"]), torch.from_numpy(g["fwd_t"]), cond, taps=taps)
    assert max_abs_rel(y, g["fwd_out"]) < TOL
    for ours, theirs in (("block0.mod1", "tap_block0.mod1"), ("block0.attn_out", "tap_block0.attn_out"),
                         ("block0.mod2", "tap_block0.mod2"), ("block0.mlp_out", "tap_block0.mlp_out")):
        assert max_abs_rel(taps[ours], g[theirs]) < TOL, ours


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("tag", ["s1", "s2"])
def test_forward_with_cfg_matches_reference(name, tag):
    g, cfg, sd = setup(name)
    cond = {k: torch.from_numpy(g[f"cfg_label_{k}"]) for k in cfg.class_vocab_sizes}
    scales = golden_json(g, f"cfg_scales_{tag}")
    y = dit_forward_with_cfg(sd, cfg, torch.from_numpy(g["cfg_x"]), torch.from_numpy(g["cfg_t"]), cond, scales)
    assert max_abs_rel(y, g[f"cfg_out_{tag}"]) < TOL


def test_fp64_oracle_is_closer_than_tolerance():
    g, cfg, sd = setup("dit_base", torch.float64)
    cond = {k: torch.from_numpy(g[f"fwd_label_{k}"]) for k in golden_json(g, "fwd_classes")}
    y = dit_forward(sd, cfg, torch.from_numpy(g["fwd_x"]), torch.from_numpy(g["fwd_t"]), cond)
    assert y.dtype == torch.float64 and max_abs_rel(y, g["fwd_out"]) < TOL


def test_modulate_argument_swap_kat():
    """chunk 0 acts as the SCALE in Block; 'fixing' the order must break parity."""
    g, cfg, sd = setup("dit_tiny")
    x = torch.from_numpy(g["fwd_x"])
    cond = {k: torch.from_numpy(g[f"fwd_label_{k}"]) for k in golden_json(g, "fwd_classes")}
    taps = {}
    dit_forward(sd, cfg, x, torch.from_numpy(g["fwd_t"]), cond, taps=taps)
    c = taps["c"]
    m = torch.nn.functional.silu(c) @ sd["blocks.0.adaln_modulation.1.weight"].T + sd["blocks.0.adaln_modulation.1.bias"]
    a0, a1 = m.chunk(6, dim=-1)[:2]
    ln = layer_norm(taps["h0"], cfg.layernorm_eps)
    as_reference = ln * (1 + a0) + a1
    as_named = ln * (1 + a1) + a0
    assert max_abs_rel(as_reference, g["tap_block0.mod1"]) < TOL
    assert max_abs_rel(as_named, g["tap_block0.mod1"]) > 1e-2


def test_shapes_and_flops():
    assert mlp_hidden_dim(256, 4) == 684 and mlp_hidden_dim(32, 4) == 88
    assert dit_flops_per_sample(DiTConfig(class_vocab_sizes={"clusters": 14})) == 210_763_776
    g = load_golden("dit_base")
    shapes = golden_json(g, "shapes_json")
    assert len(shapes) == 84 and shapes["blocks.0.mlp.w1.weight"] == [684, 256]
    assert sum(int(np.prod(v)) for v in shapes.values()) == 9_745_424
