"""GPU parity tests (run with -m gpu on the MI355X box): the HIP path, called through the C ABI via the
reference-shaped Python classes, against the golden vectors of the reference and against the CPU oracle.

Tolerances (scale-relative max error, max|a-b| / max|b|):
  fp32 path (exact-fp32 MFMA): 1e-4  - BASELINE.json north_star gate
  bf16 path (bf16 operands, fp32 accumulate/LN/softmax/residual): 3e-2 (bf16 has 8 mantissa bits; 8 layers deep)
"""
import numpy as np
import pytest
import torch

from conftest import golden_json, load_golden, max_abs_rel
from oracle.dit import DiTConfig, dit_forward, dit_forward_with_cfg
from oracle.transport import sample_ode_fixed
from oracle.weights import make_state_dict

pytestmark = pytest.mark.gpu
TOL_FP32 = 1e-4
TOL_BF16 = 3e-2


def build(name, precision="fp32"):
    from scldm_amd.nnets import DiT
    g = load_golden(name)
    kw = golden_json(g, "kwargs_json")
    shapes = {k: tuple(v) for k, v in golden_json(g, "shapes_json").items()}
    sd = make_state_dict(shapes, int(g["seed"]))
    m = DiT(**kw)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    m.precision = precision
    cfg = DiTConfig(n_embed=kw["n_embed"], n_embed_input=kw["n_embed_input"], n_layer=kw["n_layer"], n_head=kw["n_head"],
                    seq_len=kw["seq_len"], multiple_of=kw["multiple_of"], layernorm_eps=kw["layernorm_eps"],
                    class_vocab_sizes=kw["class_vocab_sizes"], condition_strategy=kw["condition_strategy"])
    return g, m, cfg, sd


def cu(a):
    return torch.from_numpy(np.asarray(a)).cuda()


@pytest.mark.parametrize("name", ["dit_base", "dit_joint"])
@pytest.mark.parametrize("precision,tol", [("fp32", TOL_FP32), ("bf16", TOL_BF16)])
def test_forward_matches_reference_golden(name, precision, tol):
    g, m, cfg, sd = build(name, precision)
    cond = {k: cu(g[f"fwd_label_{k}"]) for k in golden_json(g, "fwd_classes")}
    y = m(cu(g["fwd_x"]), cu(g["fwd_t"]), cond)
    assert y.shape == g["fwd_out"].shape and torch.isfinite(y).all()
    assert max_abs_rel(y.cpu(), g["fwd_out"]) < tol


@pytest.mark.parametrize("name", ["dit_base", "dit_joint"])
@pytest.mark.parametrize("tag", ["s1", "s2"])
def test_forward_with_cfg_matches_reference_golden(name, tag):
    g, m, cfg, sd = build(name)
    cond = {k: cu(g[f"cfg_label_{k}"]) for k in cfg.class_vocab_sizes}
    scales = golden_json(g, f"cfg_scales_{tag}")
    x, t = cu(g["cfg_x"]), cu(g["cfg_t"])
    y = m.forward_with_cfg(x, t, cond, scales)           # per-sample-t path
    assert max_abs_rel(y.cpu(), g[f"cfg_out_{tag}"]) < TOL_FP32
    t._scldm_uniform_t = True                              # scalar-t path with label de-duplication
    y2 = m.forward_with_cfg(x, t, cond, scales)
    assert max_abs_rel(y2.cpu(), g[f"cfg_out_{tag}"]) < TOL_FP32


@pytest.mark.parametrize("n", [1, 3, 8, 13, 37])
def test_ragged_batches_vs_oracle(n):
    """Batch sizes that do not fill a 64/128-token tile (tile padding, odd sample pairing)."""
    g, m, cfg, sd = build("dit_base")
    rng = np.random.default_rng(n)
    x = rng.standard_normal((n, 16, 16)).astype(np.float32)
    t = rng.uniform(0, 1, n).astype(np.float32)
    lab = rng.integers(0, 14, n).astype(np.int64)
    ref = dit_forward(sd, cfg, torch.from_numpy(x), torch.from_numpy(t), {"clusters": torch.from_numpy(lab)})
    y = m(cu(x), cu(t), {"clusters": cu(lab)})
    assert max_abs_rel(y.cpu(), ref) < TOL_FP32


def test_bf16_vs_oracle_medium_batch():
    g, m, cfg, sd = build("dit_base", "bf16")
    rng = np.random.default_rng(5)
    n = 96
    x = rng.standard_normal((n, 16, 16)).astype(np.float32)
    t = rng.uniform(0, 1, n).astype(np.float32)
    lab = rng.integers(0, 14, n).astype(np.int64)
    ref = dit_forward(sd, cfg, torch.from_numpy(x), torch.from_numpy(t), {"clusters": torch.from_numpy(lab)})
    y = m(cu(x), cu(t), {"clusters": cu(lab)})
    assert max_abs_rel(y.cpu(), ref) < TOL_BF16


@pytest.mark.parametrize("name,method,steps", [("dit_base", "euler", 5), ("dit_base", "heun", 4), ("dit_joint", "euler", 4)])
def test_fused_sampler_vs_oracle(name, method, steps):
    g, m, cfg, sd = build(name)
    rng = np.random.default_rng(11)
    B = 6
    z0 = rng.standard_normal((B, 16, 16)).astype(np.float32)
    labs = {k: rng.integers(0, v, B).astype(np.int64) for k, v in cfg.class_vocab_sizes.items()}
    scales = {k: 1.5 for k in cfg.class_vocab_sizes}
    z2 = torch.from_numpy(np.concatenate([z0, z0]))
    cond2 = {k: torch.from_numpy(np.concatenate([v, v])) for k, v in labs.items()}
    ref = sample_ode_fixed(z2, lambda x, t: dit_forward_with_cfg(sd, cfg, x, t, cond2, scales), steps, method)
    out = m.sample_ode_cfg(z2.cuda(), {k: v.cuda() for k, v in cond2.items()}, scales, steps, method)
    assert max_abs_rel(out.cpu(), ref) < TOL_FP32
    # the generic reference-style call chain (Sampler -> lambda -> forward_with_cfg) gives the same trajectory end
    from scldm_amd.transport import Sampler, create_transport
    fn = Sampler(create_transport("Linear", "velocity", "velocity", 1e-5, 1e-5)).sample_ode(sampling_method=method, num_steps=steps)
    condc = {k: v.cuda() for k, v in cond2.items()}
    model_fn = lambda x, t, **kw: m.forward_with_cfg(x, t, **kw, cfg_scale=scales)
    traj = fn(z2.cuda(), model_fn, **{"condition": condc})
    assert traj.shape[0] == steps and max_abs_rel(traj[-1].cpu(), ref) < TOL_FP32
    assert max_abs_rel(traj[-1].cpu(), out.cpu()) < 1e-5


def test_bitwise_repeatability_and_batch_permutation_at_full_size():
    """Size-independent properties at the benchmark batch (3 x 4096 sample-forwards): identical bytes on repeat,
    and permuting the cells permutes the outputs exactly (cells are independent; cross-sample MFMA blocks are masked)."""
    g, m, cfg, sd = build("dit_base", "bf16")
    gen = torch.Generator(device="cuda").manual_seed(0)
    n = 12288
    x = torch.randn(n, 16, 16, device="cuda", generator=gen)
    t = torch.rand(n, device="cuda", generator=gen)
    lab = torch.randint(0, 14, (n,), device="cuda", generator=gen)
    y1 = m(x, t, {"clusters": lab})
    y2 = m(x, t, {"clusters": lab})
    assert torch.equal(y1, y2) and torch.isfinite(y1).all()
    perm = torch.randperm(n, device="cuda", generator=gen)
    y3 = m(x[perm], t[perm], {"clusters": lab[perm]})
    assert torch.equal(y3, y1[perm])
    # spot-check 16 random cells of the big batch against the oracle
    idx = perm[:16].cpu()
    ref = dit_forward(sd, cfg, x.cpu()[idx], t.cpu()[idx], {"clusters": lab.cpu()[idx]})
    assert max_abs_rel(y1.cpu()[idx], ref) < TOL_BF16


def test_error_paths():
    from scldm_amd._lib import ScldmError
    g, m, cfg, sd = build("dit_base")
    with pytest.raises(ValueError):
        m(torch.zeros(2, 16, 8, device="cuda"), torch.zeros(2, device="cuda"), {"clusters": torch.zeros(2, dtype=torch.long, device="cuda")})
    with pytest.raises(ValueError):
        m(torch.zeros(2, 16, 16, device="cuda"), torch.zeros(2, device="cuda"), {"clusters": torch.zeros(3, dtype=torch.long, device="cuda")})
    y = m(torch.zeros(2, 16, 16, device="cuda"), torch.zeros(2, device="cuda"), {"clusters": torch.zeros(2, dtype=torch.long, device="cuda")})
    assert not y.requires_grad        # eval mode: fused inference kernel, output is not differentiable w.r.t. the parameters
    m.train()
    y = m(torch.zeros(2, 16, 16, device="cuda"), torch.zeros(2, device="cuda"), {"clusters": torch.zeros(2, dtype=torch.long, device="cuda")})
    assert y.requires_grad and y.grad_fn is not None   # training mode: autograd-bound HIP backward (tests/test_gpu_train.py)


def test_training_mode_label_dropout_matches_oracle_mixture():
    """Training-mode forward (force_drop_ids): every row equals the oracle output for its label OR for the null token,
    and with cfg_dropout_prob=0.8 most rows are dropped (nnets.py:401-402)."""
    g, m, cfg, sd = build("dit_base")
    rng = np.random.default_rng(3)
    n = 64
    x = rng.standard_normal((n, 16, 16)).astype(np.float32)
    t = rng.uniform(0, 1, n).astype(np.float32)
    lab = rng.integers(0, 14, n).astype(np.int64)
    ref_lab = dit_forward(sd, cfg, torch.from_numpy(x), torch.from_numpy(t), {"clusters": torch.from_numpy(lab)})
    ref_null = dit_forward(sd, cfg, torch.from_numpy(x), torch.from_numpy(t), {"clusters": torch.full((n,), 14, dtype=torch.long)})
    m.train()
    torch.manual_seed(0)
    with torch.no_grad():
        y = m(cu(x), cu(t), {"clusters": cu(lab)}).cpu()
    m.eval()
    scale = float(ref_lab.abs().max())
    e_lab = (y - ref_lab).abs().amax(dim=(1, 2)) / scale
    e_null = (y - ref_null).abs().amax(dim=(1, 2)) / scale
    assert bool((torch.minimum(e_lab, e_null) < TOL_FP32).all())
    dropped = int((e_null < e_lab).sum())
    assert 35 <= dropped <= 62  # Binomial(64, 0.8)


@pytest.mark.parametrize("vocab,strategy,B,method,steps,scale", [
    ({"cell_type": 50}, "mutually_exclusive", 2048, "heun", 3, 2.0),                 # BASELINE configs[2] shape (hlca)
    ({"cell_type": 18, "cytokine": 91}, "joint", 1024, "euler", 4, 1.0),              # configs[3] per-GPU shard (parse1m)
])
def test_fused_sampler_full_size_properties(vocab, strategy, B, method, steps, scale):
    """At BASELINE batch sizes: the fused bf16 sampler is bit-repeatable, sharding the batch gives the same cells
    (cells are independent), and a handful of cells agree with the oracle chain."""
    from scldm_amd.nnets import DiT
    from scldm_amd.sampling import sample_latents, shard_bounds
    kw = dict(n_embed=256, n_embed_input=16, n_layer=8, n_head=8, seq_len=16, dropout=0.0, bias=True, norm_layer="layernorm",
              multiple_of=4, layernorm_eps=1e-8, class_vocab_sizes=vocab, cfg_dropout_prob=0.8, condition_strategy=strategy)
    m = DiT(**kw)
    sd = make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 321)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    m.precision = "bf16"
    cfg = DiTConfig(class_vocab_sizes=vocab, condition_strategy=strategy)
    gen = torch.Generator(device="cuda").manual_seed(1)
    z0 = torch.randn(B, 16, 16, device="cuda", generator=gen)
    cond = {k: torch.randint(0, v, (B,), device="cuda", generator=gen) for k, v in vocab.items()}
    scales = {k: scale for k in vocab}
    out1 = sample_latents(m, z0, cond, scales, steps, method)
    out2 = sample_latents(m, z0, cond, scales, steps, method)
    assert out1.shape == (2 * B, 16, 16) and torch.equal(out1, out2) and torch.isfinite(out1).all()
    # shard-and-concatenate == whole batch (what the multi-GPU path relies on); tile pairing differs, values must not
    lo, hi = shard_bounds(B, 1, 3)
    part = sample_latents(m, z0[lo:hi], {k: v[lo:hi] for k, v in cond.items()}, scales, steps, method)
    b = hi - lo
    assert torch.equal(part[:b], out1[lo:hi]) and torch.equal(part[b:], out1[B + lo:B + hi])
    # oracle spot check on 4 cells
    idx = torch.tensor([0, 7, B // 2, B - 1])
    z2 = torch.cat([z0[idx.cuda()], z0[idx.cuda()]]).cpu()
    c2 = {k: torch.cat([v[idx.cuda()], v[idx.cuda()]]).cpu() for k, v in cond.items()}
    ref = sample_ode_fixed(z2, lambda x, t: dit_forward_with_cfg(sd, cfg, x, t, c2, scales), steps, method)
    got = torch.cat([out1[idx.cuda()], out1[(idx + B).cuda()]]).cpu()
    assert max_abs_rel(got, ref) < TOL_BF16


def test_adaptive_dopri5_sampling_agrees_with_fine_heun():
    """The reference's default sampler (sample_ode() -> dopri5, transport.py:324-332; parity unpinned, see oracle/transport.py):
    adaptive solve over the fused forward_with_cfg agrees with a 400-evaluation fused Heun solve, both through
    DiT.sample_ode_cfg and through the reference-style Sampler call (models.py:793-812)."""
    from scldm_amd.transport import Sampler, create_transport
    g, m, cfg, sd = build("dit_base")
    gen = torch.Generator(device="cuda").manual_seed(4)
    B = 6
    z0 = torch.randn(B, 16, 16, device="cuda", generator=gen)
    z2 = torch.cat([z0, z0])
    cond = {"clusters": torch.randint(0, 14, (B,), device="cuda", generator=gen).repeat(2)}
    scales = {"clusters": 2.0}
    ref = m.sample_ode_cfg(z2, cond, scales, 201, "heun")
    out = m.sample_ode_cfg(z2, cond, scales, 2, "dopri5", atol=1e-6, rtol=1e-6)
    assert max_abs_rel(out.cpu(), ref.cpu()) < 1e-3
    fn = Sampler(create_transport("Linear", "velocity", "velocity", 1e-5, 1e-5)).sample_ode()      # the reference's default call
    traj = fn(z2, m.forward_with_cfg, condition=cond, cfg_scale=scales)
    assert traj.shape == (50, 2 * B, 16, 16) and torch.equal(traj[0], z2)
    assert max_abs_rel(traj[-1].cpu(), ref.cpu()) < 1e-3
    assert 20 < fn.last_stats["evaluations"] < 2000


def test_forward_with_cfg_joint_mirror():
    """nnets.py:299-334: u + s * (c - u) on every row, from two plain forwards (joint strategy, hard-coded 'cell_line' scale key)."""
    from scldm_amd.nnets import DiT
    kw = dict(n_embed=256, n_embed_input=16, n_layer=2, n_head=8, seq_len=16, dropout=0.0, bias=True, norm_layer="layernorm", multiple_of=4,
              layernorm_eps=1e-8, class_vocab_sizes={"cell_line": 4, "gene": 30}, cfg_dropout_prob=0.8, condition_strategy="joint")
    m = DiT(**kw)
    sd = make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 55)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    cfg = DiTConfig(n_layer=2, class_vocab_sizes=kw["class_vocab_sizes"], condition_strategy="joint")
    gen = torch.Generator().manual_seed(2)
    x, t = torch.randn(5, 16, 16, generator=gen), torch.rand(5, generator=gen)
    cond = {"cell_line": torch.randint(0, 4, (5,), generator=gen), "gene": torch.randint(0, 30, (5,), generator=gen)}
    null = {"cell_line": torch.full((5,), 4), "gene": torch.full((5,), 30)}
    u, c = dit_forward(sd, cfg, x, t, null), dit_forward(sd, cfg, x, t, cond)
    out = m.forward_with_cfg_joint(x.cuda(), t.cuda(), {k: v.cuda() for k, v in cond.items()}, {"cell_line": 1.7})
    assert max_abs_rel(out.cpu(), u + 1.7 * (c - u)) < TOL_FP32


@pytest.mark.parametrize("n_layer", [1, 3])
@pytest.mark.parametrize("precision,tol", [("fp32", TOL_FP32), ("bf16", TOL_BF16)])
def test_odd_layer_counts(n_layer, precision, tol):
    """The fused kernel runs two layers per launch: odd depths end with a one-layer launch (and depth 1 is input projection,
    block and final layer in a single slot)."""
    from scldm_amd.nnets import DiT
    kw = dict(n_embed=256, n_embed_input=16, n_layer=n_layer, n_head=8, seq_len=16, dropout=0.0, bias=True, norm_layer="layernorm",
              multiple_of=4, layernorm_eps=1e-8, class_vocab_sizes={"clusters": 14}, cfg_dropout_prob=0.8)
    m = DiT(**kw)
    sd = make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 60 + n_layer)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    m.precision = precision
    cfg = DiTConfig(n_layer=n_layer, class_vocab_sizes={"clusters": 14})
    gen = torch.Generator().manual_seed(n_layer)
    x, t = torch.randn(9, 16, 16, generator=gen), torch.rand(9, generator=gen)
    lab = torch.randint(0, 14, (9,), generator=gen)
    with torch.no_grad():
        y = m(x.cuda(), t.cuda(), {"clusters": lab.cuda()})
    assert max_abs_rel(y.cpu(), dit_forward(sd, cfg, x, t, {"clusters": lab})) < tol


def test_tile_group_launches_are_bit_identical(monkeypatch):
    """SCLDM_GROUPS=2 (each fused launch split into two tile groups on two streams) and SCLDM_LPL=1 (one layer per launch)
    are scheduling choices only: same bits as the default."""
    g, m, cfg, sd = build("dit_base", "bf16")
    gen = torch.Generator(device="cuda").manual_seed(12)
    n = 4100                                             # 1 025 tiles of 64 tokens: enough for two groups, ragged tail
    x = torch.randn(n, 16, 16, device="cuda", generator=gen)
    t = torch.rand(n, device="cuda", generator=gen)
    lab = torch.randint(0, 14, (n,), device="cuda", generator=gen)
    with torch.no_grad():
        ref = m(x, t, {"clusters": lab})
        monkeypatch.setenv("SCLDM_GROUPS", "2")
        y = m(x, t, {"clusters": lab})
        torch.cuda.synchronize()
    assert torch.equal(y, ref)
