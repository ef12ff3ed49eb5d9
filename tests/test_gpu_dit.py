"""GPU parity tests (run with -m gpu on the MI355X box): the HIP path, called through the C ABI via the
reference-shaped Python classes, against the golden vectors of the reference and against the CPU oracle.

Tolerances (conftest.check_err: scale-relative max error max|a-b| / max|b| asserted against the tolerance, the elementwise
relative error floored at 1 % of max|b| printed beside it and asserted against 10 x the tolerance):
  fp32 path   (exact-fp32 MFMA): 1e-4  - BASELINE.json north_star gate
  bf16x3 path (split-bf16, three bf16 MFMAs per product sum - the reference's "high" matmul precision class): 1e-4, same gate
  bf16 path   (bf16 operands, fp32 accumulate/LN/softmax/residual): 3e-2 (bf16 has 8 mantissa bits; 8 layers deep)
"""
import numpy as np
import pytest
import torch

from conftest import check_err, golden_json, load_golden, max_abs_rel
from oracle.dit import DiTConfig, dit_forward, dit_forward_with_cfg
from oracle.transport import sample_ode_fixed
from oracle.weights import make_state_dict

pytestmark = pytest.mark.gpu
TOL_FP32 = 1e-4
TOL_BF16 = 3e-2
PARITY = [("fp32", TOL_FP32), ("bf16x3", TOL_FP32)]
# bound on the floored ELEMENTWISE relative error (an element at 1 % of the output scale): fp32 measures <= 1.3e-4, split-bf16
# <= 1.5e-3 (its scale-relative error is 1-2e-5; CFG with scale 2 doubles differences), bf16 is reported only
FLOOR_TOL = {"fp32": 1e-3, "bf16x3": 5e-3, "bf16": float("inf")}


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


@pytest.mark.parametrize("name", ["dit_base", "dit_joint", "dit_me2_256"])
@pytest.mark.parametrize("precision,tol", PARITY + [("bf16", TOL_BF16)])
def test_forward_matches_reference_golden(name, precision, tol):
    g, m, cfg, sd = build(name, precision)
    cond = {k: cu(g[f"fwd_label_{k}"]) for k in golden_json(g, "fwd_classes")}
    y = m(cu(g["fwd_x"]), cu(g["fwd_t"]), cond)
    assert y.shape == g["fwd_out"].shape and torch.isfinite(y).all()
    check_err(y.cpu(), g["fwd_out"], tol, f"forward {name} [{precision}] vs reference golden", FLOOR_TOL[precision])


@pytest.mark.parametrize("name", ["dit_base", "dit_joint", "dit_me2_256"])
@pytest.mark.parametrize("tag", ["s1", "s2"])
@pytest.mark.parametrize("precision,tol", PARITY)
def test_forward_with_cfg_matches_reference_golden(name, tag, precision, tol):
    """dit_me2_256: two mutually-exclusive classes = TWO conditional passes through the fused kernel with distinct scales
    (nnets.py:372-376).  The golden t is the scalar broadcast an ODE solver passes (integrators.py:103-104)."""
    g, m, cfg, sd = build(name, precision)
    cond = {k: cu(g[f"cfg_label_{k}"]) for k in cfg.class_vocab_sizes}
    scales = golden_json(g, f"cfg_scales_{tag}")
    x, t = cu(g["cfg_x"]), cu(g["cfg_t"])
    m.detect_uniform_t = False
    y = m.forward_with_cfg(x, t, cond, scales)           # per-sample-t path: one conditioning row per sample-forward
    check_err(y.cpu(), g[f"cfg_out_{tag}"], tol, f"forward_with_cfg {name}/{tag} [{precision}] per-sample t", FLOOR_TOL[precision])
    m.detect_uniform_t = True                              # dense uniform t: detected on device -> shared rows + label de-duplication
    y2 = m.forward_with_cfg(x, t, cond, scales)
    check_err(y2.cpu(), g[f"cfg_out_{tag}"], tol, f"forward_with_cfg {name}/{tag} [{precision}] dense uniform t", FLOOR_TOL[precision])
    y3 = m.forward_with_cfg(x, t[:1].expand(t.shape[0]), cond, scales)   # stride-0 view (what scldm_amd.transport passes): no sync
    assert torch.equal(y3, y2)


# ---- the reference-class fast precision: fp16 operands = TF32's 10 mantissa bits (VERDICT r2 missing #3) ------------------------
# Its tolerance is DERIVED from the reference's own arithmetic: oracle.dit.matmul_operand_bits(10) rounds every matmul operand to
# 10 mantissa bits (what set_float32_matmul_precision("high") does, inference.py:26); the fp16 path must be within 1.5 x that
# mode's error, both measured against the reference's exact-fp32 golden outputs on the same inputs.
TF32_FACTOR = 1.5


@pytest.mark.parametrize("name", ["dit_base", "dit_joint", "dit_me2_256"])
def test_fp16_forward_is_in_the_references_tf32_class(name):
    from oracle.dit import matmul_operand_bits
    g, m, cfg, sd = build(name, "fp16")
    cond_c = {k: torch.from_numpy(g[f"fwd_label_{k}"]) for k in golden_json(g, "fwd_classes")}
    x, t = torch.from_numpy(g["fwd_x"]), torch.from_numpy(g["fwd_t"])
    with matmul_operand_bits(10):
        y_tf32 = dit_forward(sd, cfg, x, t, cond_c)
    e_tf32 = max_abs_rel(y_tf32, g["fwd_out"])
    y = m(x.cuda(), t.cuda(), {k: v.cuda() for k, v in cond_c.items()})
    e = max_abs_rel(y.cpu(), g["fwd_out"])
    print(f"[parity] forward {name} [fp16] vs reference golden: {e:.3e}; TF32-operand oracle vs the same golden: {e_tf32:.3e} (ratio {e / e_tf32:.2f})")
    assert torch.isfinite(y).all() and e <= TF32_FACTOR * e_tf32 and e < 5e-3
    rep = m.fp16_weight_report()
    assert rep["overflow"] == 0 and rep["nonzero"] > 1_000_000
    # CFG (scale 2 doubles differences) on the scalar-t path
    for tag in ("s1", "s2"):
        condc = {k: torch.from_numpy(g[f"cfg_label_{k}"]) for k in cfg.class_vocab_sizes}
        scales = golden_json(g, f"cfg_scales_{tag}")
        xc, tc = torch.from_numpy(g["cfg_x"]), torch.from_numpy(g["cfg_t"])
        with matmul_operand_bits(10):
            r_tf32 = dit_forward_with_cfg(sd, cfg, xc, tc, condc, scales)
        e_tf32 = max_abs_rel(r_tf32, g[f"cfg_out_{tag}"])
        yc = m.forward_with_cfg(xc.cuda(), tc.cuda(), {k: v.cuda() for k, v in condc.items()}, scales)
        e = max_abs_rel(yc.cpu(), g[f"cfg_out_{tag}"])
        print(f"[parity] forward_with_cfg {name}/{tag} [fp16]: {e:.3e}; TF32-operand oracle: {e_tf32:.3e} (ratio {e / e_tf32:.2f})")
        assert e <= TF32_FACTOR * e_tf32


def test_fp16_sampler_is_in_the_references_tf32_class_and_bf16_is_not():
    """20 CFG Euler evaluations, 6 cells: fp16 stays within 1.5 x the TF32-operand oracle's distance from the exact-fp32 oracle; the
    bf16 throughput path (8 mantissa bits) is several times further out - which is why it is not the reference-class path."""
    from oracle.dit import matmul_operand_bits
    g, m, cfg, sd = build("dit_base", "fp16")
    gen = torch.Generator().manual_seed(9)
    B = 6
    z0 = torch.randn(B, 16, 16, generator=gen)
    z2 = torch.cat([z0, z0])
    cond = {"clusters": torch.randint(0, 14, (B,), generator=gen).repeat(2)}
    scales = {"clusters": 2.0}
    f = lambda xx, tt: dit_forward_with_cfg(sd, cfg, xx, tt, cond, scales)
    ref = sample_ode_fixed(z2, f, 21, "euler")
    with matmul_operand_bits(10):
        r_tf32 = sample_ode_fixed(z2, f, 21, "euler")
    e_tf32 = max_abs_rel(r_tf32, ref)
    errs = {}
    for prec in ("fp16", "bf16"):
        m.precision = prec
        out = m.sample_ode_cfg(z2.cuda(), {k: v.cuda() for k, v in cond.items()}, scales, 21, "euler")
        errs[prec] = max_abs_rel(out.cpu(), ref)
    print(f"[parity] sampler 20 evals: fp16 {errs['fp16']:.3e}, TF32-operand oracle {e_tf32:.3e}, bf16 {errs['bf16']:.3e}")
    assert errs["fp16"] <= TF32_FACTOR * e_tf32
    assert errs["bf16"] > 2.0 * errs["fp16"]


def test_dopri5_device_kernels_reproduce_the_host_composed_solver():
    """Round 5: on a CUDA fp32 state the Dormand-Prince stage points, error ratio and dense output run as four HIP kernels (scldm_rk_*)
    that perform the same fp32 operations in the same order as the torch expressions they replace.  A smooth analytic field (no DiT):
    the device path on the GPU against the host-composed path of the same driver on the CPU - same accepted / rejected step sequence,
    trajectories within 1e-6 (the error ratio's double sum is ordered differently, nothing else), and both at the analytic solution."""
    from scldm_amd.transport import Sampler, create_transport
    gen = torch.Generator().manual_seed(4)
    x = torch.randn(24, 16, 16, generator=gen)
    w = torch.randn(24, 1, 1, generator=gen) * 0.5

    def field(xx, tt, **kw):          # dx/dt = -(1 + t) w x  ->  x(t) = x0 exp(-w (t + t^2 / 2))
        return -(1.0 + tt.view(-1, 1, 1)) * kw["w"] * xx
    fn_c = Sampler(create_transport()).sample_ode()
    fn_g = Sampler(create_transport()).sample_ode()
    ref = fn_c(x, field, w=w)
    out = fn_g(x.cuda(), lambda xx, tt, **kw: field(xx, tt, **kw), w=w.cuda())
    sc, sg = fn_c.last_stats, fn_g.last_stats
    assert sg["evaluations"] == sc["evaluations"] and len(sg["accepted_steps"]) == len(sc["accepted_steps"]) and sg["rejected"] == sc["rejected"]
    assert max(abs(a[1] - b[1]) / b[1] for a, b in zip(sg["accepted_steps"], sc["accepted_steps"])) < 1e-6
    assert out.shape == ref.shape == (50, 24, 16, 16)
    assert float((out.cpu() - ref).abs().max()) <= 1e-6 * float(ref.abs().max())
    exact = x * torch.exp(-w * 1.5)
    assert float((out[-1].cpu() - exact).abs().max()) < 5e-5 * float(exact.abs().max())


LONG_RUNS = [
    # (tag, vocab, strategy, method, grid points, guidance): BASELINE.json configs[2] (hlca: 100 Heun steps = 200 evaluations, CFG 2.0),
    # the north-star row / configs[1] vocabulary (dentate, 100 Euler) and configs[3] (parse1m, joint conditioning, 100 Euler)
    ("hlca_heun100", {"cell_type": 50}, "mutually_exclusive", "heun", 101, 2.0),
    ("dentate_euler100", {"clusters": 14}, "mutually_exclusive", "euler", 101, 1.0),
    ("parse1m_joint_euler100", {"cell_type": 18, "cytokine": 91}, "joint", "euler", 101, 1.0),
]


@pytest.mark.parametrize("tag,vocab,strategy,method,steps,scale", LONG_RUNS)
def test_long_trajectories_at_every_precision(tag, vocab, strategy, method, steps, scale):
    """error growth over the FULL trajectories the configs name (100 Heun steps = 200 CFG evaluations at guidance
    2.0; 100 Euler evaluations), 4 cells (8 until round 4: the float64 CPU chains were 40 % of the GPU suite's wall time), every precision
    policy, against the float64 oracle chain.  fp32 and bf16x3 stay inside the
    1e-4 gate; fp16 within 1.5 x the distance of the TF32-operand oracle (the reference's own arithmetic) from the same float64
    chain; bf16 is reported and bounded by its own tolerance."""
    from oracle.dit import matmul_operand_bits
    from scldm_amd.nnets import DiT
    kw = dict(n_embed=256, n_embed_input=16, n_layer=8, n_head=8, seq_len=16, dropout=0.0, bias=True, norm_layer="layernorm",
              multiple_of=4, layernorm_eps=1e-8, class_vocab_sizes=vocab, cfg_dropout_prob=0.8, condition_strategy=strategy)
    m = DiT(**kw)
    sd = make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 2024)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    cfg = DiTConfig(class_vocab_sizes=vocab, condition_strategy=strategy)
    gen = torch.Generator().manual_seed(17)
    B = 4
    z0 = torch.randn(B, 16, 16, generator=gen)
    z2 = torch.cat([z0, z0])
    cond = {k: torch.randint(0, v, (B,), generator=gen).repeat(2) for k, v in vocab.items()}
    scales = {k: scale for k in vocab}
    sd64 = {k: v.double() for k, v in sd.items()}
    n_thr = torch.get_num_threads()
    torch.set_num_threads(min(16, n_thr))     # (an all-cores OpenMP team on these 384-row GEMMs is pathological on the 256-thread GPU hosts)
    try:
        ref64 = sample_ode_fixed(z2.double(), lambda xx, tt: dit_forward_with_cfg(sd64, cfg, xx, tt, cond, scales), steps, method)
        f32 = lambda xx, tt: dit_forward_with_cfg(sd, cfg, xx, tt, cond, scales)
        with matmul_operand_bits(10):
            e_tf32 = max_abs_rel(sample_ode_fixed(z2, f32, steps, method), ref64.float())
    finally:
        torch.set_num_threads(n_thr)
    condg = {k: v.cuda() for k, v in cond.items()}
    errs = {}
    for prec in ("fp32", "bf16x3", "fp16", "bf16"):
        m.precision = prec
        out = m.sample_ode_cfg(z2.cuda(), condg, scales, steps, method)
        assert torch.isfinite(out).all()
        errs[prec] = max_abs_rel(out.cpu(), ref64.float())
    n_eval = (steps - 1) * (2 if method == "heun" else 1)
    print(f"[parity] long trajectory {tag} ({n_eval} CFG evaluations, guidance {scale}) vs float64 oracle: "
          + ", ".join(f"{k} {v:.3e}" for k, v in errs.items()) + f"; TF32-operand oracle {e_tf32:.3e}")
    assert errs["fp32"] < TOL_FP32 and errs["bf16x3"] < TOL_FP32
    assert errs["fp16"] <= TF32_FACTOR * e_tf32
    assert errs["bf16"] < TOL_BF16


@pytest.mark.parametrize("strategy,vocab", [("mutually_exclusive", {"clusters": 14}), ("joint", {"cell_line": 4, "gene": 30})])
def test_forward_with_cfg_and_sampling_work_in_training_mode(strategy, vocab):
    """The reference's forward_with_cfg runs in either mode (it calls forward(..., force_drop_ids=False), nnets.py:353,367,375):
    mutually_exclusive gives the eval-mode result; joint keeps drawing its label-dropout mask in training mode (nnets.py:440-445),
    so every guided row equals the oracle's result for its labels OR for the null tokens.  Sampling without .eval() works too."""
    from scldm_amd.nnets import DiT
    kw = dict(n_embed=256, n_embed_input=16, n_layer=2, n_head=8, seq_len=16, dropout=0.0, bias=True, norm_layer="layernorm",
              multiple_of=4, layernorm_eps=1e-8, class_vocab_sizes=vocab, cfg_dropout_prob=0.5, condition_strategy=strategy)
    m = DiT(**kw)
    sd = make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 77)
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    cfg = DiTConfig(n_layer=2, class_vocab_sizes=vocab, condition_strategy=strategy)
    gen = torch.Generator().manual_seed(5)
    B = 24
    x = torch.randn(2 * B, 16, 16, generator=gen)
    t = torch.full((2 * B,), 0.3)
    cond = {k: torch.randint(0, v, (B,), generator=gen).repeat(2) for k, v in vocab.items()}
    null = {k: torch.full((2 * B,), v, dtype=torch.long) for k, v in vocab.items()}
    scales = {k: 1.7 for k in vocab}
    ref = dit_forward_with_cfg(sd, cfg, x, t, cond, scales)
    ref_null = dit_forward_with_cfg(sd, cfg, x, t, null, scales)
    m.eval()
    y_eval = m.forward_with_cfg(x.cuda(), t.cuda(), {k: v.cuda() for k, v in cond.items()}, scales).cpu()
    assert max_abs_rel(y_eval, ref) < TOL_FP32
    m.train()
    torch.manual_seed(1)
    with torch.no_grad():
        y = m.forward_with_cfg(x.cuda(), t.cuda(), {k: v.cuda() for k, v in cond.items()}, scales).cpu()
    scale_ = float(ref.abs().max())
    e_lab = (y - ref).abs().amax(dim=(1, 2)) / scale_
    e_null = (y - ref_null).abs().amax(dim=(1, 2)) / scale_
    assert bool((e_lab[:B] < TOL_FP32).all())                     # unconditional half: no labels involved
    if strategy == "joint":
        assert bool((torch.minimum(e_lab, e_null)[B:] < TOL_FP32).all())
        dropped = int((e_null[B:] < e_lab[B:]).sum())
        assert 3 <= dropped <= 21                                 # Binomial(24, 0.5)
    else:
        assert bool((e_lab < TOL_FP32).all())
    # gradients flow through a training-mode CFG evaluation (composed from differentiable forwards, as in the reference)
    xg = x.cuda().requires_grad_(True)
    out = m.forward_with_cfg(xg, t.cuda(), {k: v.cuda() for k, v in cond.items()}, scales)
    out.square().mean().backward()
    assert xg.grad is not None and torch.isfinite(xg.grad).all() and float(xg.grad.abs().max()) > 0
    m.zero_grad(set_to_none=True)
    z = m.sample_ode_cfg(x.cuda(), {k: v.cuda() for k, v in cond.items()}, scales, 4, "euler")       # still in training mode
    assert z.shape == x.shape and torch.isfinite(z).all()
    if strategy != "joint":
        m.eval()
        z_eval = m.sample_ode_cfg(x.cuda(), {k: v.cuda() for k, v in cond.items()}, scales, 4, "euler")
        assert max_abs_rel(z.cpu(), z_eval.cpu()) < TOL_FP32


@pytest.mark.parametrize("name", ["dit_base", "dit_joint"])
def test_guidance1_direct_option_matches_the_reference_arithmetic(name):
    """SCLDM_OPT_CFG1_DIRECT (DiT.guidance1_direct, off by default): with one conditional pass at guidance scale exactly 1.0 the
    guided half u2 + 1.0 * (c2 - u2) is the conditional output up to one fp32 rounding - the shortcut result stays inside the parity
    gate against the oracle (which evaluates the reference's expression term by term) and within 1e-6 of the default path; at any
    other scale the option changes nothing (bit-equal)."""
    g, m, cfg, sd = build(name, "fp32")
    rng = np.random.default_rng(23)
    B = 10
    z0 = rng.standard_normal((B, 16, 16)).astype(np.float32)
    labs = {k: rng.integers(0, v, B).astype(np.int64) for k, v in cfg.class_vocab_sizes.items()}
    z2 = torch.from_numpy(np.concatenate([z0, z0]))
    cond2 = {k: torch.from_numpy(np.concatenate([v, v])) for k, v in labs.items()}
    condg = {k: v.cuda() for k, v in cond2.items()}
    one = {k: 1.0 for k in cfg.class_vocab_sizes}
    ref = sample_ode_fixed(z2, lambda x, t: dit_forward_with_cfg(sd, cfg, x, t, cond2, one), 6, "euler")
    base = m.sample_ode_cfg(z2.cuda(), condg, one, 6, "euler")
    m.guidance1_direct = True
    fast = m.sample_ode_cfg(z2.cuda(), condg, one, 6, "euler")
    check_err(fast.cpu(), ref, TOL_FP32, f"guidance-1 direct sampler {name} vs oracle", FLOOR_TOL["fp32"])
    assert max_abs_rel(fast.cpu(), base.cpu()) < 1e-6
    assert torch.equal(fast[:B], base[:B])                                   # the unconditional half is the same computation
    t0 = torch.full((2 * B,), 0.25, device="cuda")[:1].expand(2 * B)         # scalar t (stride 0): forward_with_cfg takes the option too
    f_fast = m.forward_with_cfg(z2.cuda(), t0, condg, one)
    m.guidance1_direct = False
    f_base = m.forward_with_cfg(z2.cuda(), t0, condg, one)
    assert max_abs_rel(f_fast.cpu(), f_base.cpu()) < 1e-6
    other = {k: 1.5 for k in cfg.class_vocab_sizes}
    a = m.sample_ode_cfg(z2.cuda(), condg, other, 4, "heun")
    m.guidance1_direct = True
    b = m.sample_ode_cfg(z2.cuda(), condg, other, 4, "heun")
    assert torch.equal(a, b)


@pytest.mark.parametrize("precision", ["bf16", "bf16x3", "fp32"])
def test_adaln_kernels_give_a_cell_the_same_bits_in_any_batch(precision):
    """The adaLN projection runs on the matrix pipe with a kernel chosen by the number of unique conditioning rows (one row tile per wave
    up to 128 rows, rows split once + 128 x 128 blocks above; exact fp32 for the fp32 policy).  Every output element is one fixed
    sequence of MFMAs over k in all of them, so a cell's result must not depend on the batch it is evaluated in: 400 cells with ~390
    unique joint label tuples evaluated whole (block kernel) == the same cells in chunks of 40 (row-tile kernel), bit for bit."""
    from scldm_amd.nnets import DiT
    vocab = {"cell_type": 18, "cytokine": 91}
    kw = dict(n_embed=256, n_embed_input=16, n_layer=2, n_head=8, seq_len=16, dropout=0.0, bias=True, norm_layer="layernorm",
              multiple_of=4, layernorm_eps=1e-8, class_vocab_sizes=vocab, cfg_dropout_prob=0.8, condition_strategy="joint")
    m = DiT(**kw)
    sd = make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 99)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    m.precision = precision
    gen = torch.Generator(device="cuda").manual_seed(2)
    B = 400
    z0 = torch.randn(B, 16, 16, device="cuda", generator=gen)
    cond = {k: torch.randint(0, v, (B,), device="cuda", generator=gen) for k, v in vocab.items()}
    scales = {k: 1.3 for k in vocab}
    def run(lo, hi):
        z2 = torch.cat([z0[lo:hi], z0[lo:hi]])
        c2 = {k: torch.cat([v[lo:hi], v[lo:hi]]) for k, v in cond.items()}
        return m.sample_ode_cfg(z2, c2, scales, 3, "euler")
    whole = run(0, B)
    assert len(torch.unique(torch.stack([cond["cell_type"], cond["cytokine"]], 1), dim=0)) > 128
    for lo in range(0, B, 40):
        part = run(lo, lo + 40)
        assert torch.equal(part[:40], whole[lo:lo + 40]) and torch.equal(part[40:], whole[B + lo:B + lo + 40]), lo
    if precision != "fp32":
        cfg = DiTConfig(n_layer=2, class_vocab_sizes=vocab, condition_strategy="joint")
        idx = torch.arange(0, B, 57)
        z2 = torch.cat([z0[idx], z0[idx]]).cpu()
        c2 = {k: torch.cat([v[idx], v[idx]]).cpu() for k, v in cond.items()}
        ref = sample_ode_fixed(z2, lambda x, t: dit_forward_with_cfg(sd, cfg, x, t, c2, scales), 3, "euler")
        got = torch.cat([whole[idx], whole[idx + B]]).cpu()
        assert max_abs_rel(got, ref) < (TOL_FP32 if precision == "bf16x3" else TOL_BF16)


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_tail_split_gives_every_cell_the_same_bits(precision):
    """SCLDM_OPT_TAIL_SPLIT (an A/B knob, off by default: measured slower): the tiles of a launch's partial last round run as 32-token
    tiles beside the 64-token launch.  A cell's result must not depend on the tile shape that carried it: sampler outputs with the option on and off are
    equal bit for bit at batch sizes that hit each case of the split - all tiles halved (30 and 188 tiles, ragged), the slots of a
    0.6-round launch filled up (300 tiles: 212 halved), a full round plus a short tail (600 tiles: 512 + 88 halved)."""
    from scldm_amd.nnets import DiT
    vocab = {"cell_type": 18}
    kw = dict(n_embed=256, n_embed_input=16, n_layer=3, n_head=8, seq_len=16, dropout=0.0, bias=True, norm_layer="layernorm",
              multiple_of=4, layernorm_eps=1e-8, class_vocab_sizes=vocab, cfg_dropout_prob=0.8, condition_strategy="mutually_exclusive")
    m = DiT(**kw)
    sd = make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 41)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    m.precision = precision
    gen = torch.Generator(device="cuda").manual_seed(4)
    for B in (39, 250, 400, 799):       # 3 B sample-forwards of 16 tokens: 30 (29.25), 188 (187.5), 300, 600 (599.25) tiles of 64
        z0 = torch.randn(B, 16, 16, device="cuda", generator=gen)
        c = torch.randint(0, 18, (B,), device="cuda", generator=gen)
        z2, c2 = torch.cat([z0, z0]), {"cell_type": torch.cat([c, c])}
        m.tail_split = False
        a = m.sample_ode_cfg(z2, c2, {"cell_type": 1.7}, 2, "heun")
        m.tail_split = True
        b = m.sample_ode_cfg(z2, c2, {"cell_type": 1.7}, 2, "heun")
        assert torch.equal(a, b), B
        assert torch.isfinite(a).all()


def test_fp16_weight_range_check():
    """Weights beyond the fp16 range are refused when the fp16 stream is packed (pack-time range check)."""
    g, m, cfg, sd = build("dit_base", "fp16")
    x, t = torch.from_numpy(g["fwd_x"]).cuda(), torch.from_numpy(g["fwd_t"]).cuda()
    cond = {k: torch.from_numpy(g[f"fwd_label_{k}"]).cuda() for k in golden_json(g, "fwd_classes")}
    m(x, t, cond)
    with torch.no_grad():
        m.blocks[3].mlp.w1.weight[5, 7] = 7.0e4
    with pytest.raises(ValueError, match="fp16 range"):
        m(x, t, cond)
    m.precision = "bf16x3"                       # the other policies do not care
    assert torch.isfinite(m(x, t, cond)).all()
    with torch.no_grad():
        m.blocks[3].mlp.w1.weight[5, 7] = 0.01
        m.blocks[2].attn.c_proj.weight.mul_(1e-6)         # 65 536 subnormal weights (> 1 % would need more; just counted here)
    m.precision = "fp16"
    m(x, t, cond)
    rep = m.fp16_weight_report()
    assert rep["overflow"] == 0 and rep["subnormal"] >= 60_000


def test_forward_with_cfg_nonuniform_dense_t_takes_the_per_sample_path():
    g, m, cfg, sd = build("dit_base")
    cond = {"clusters": cu(g["cfg_label_clusters"])}
    scales = {"clusters": 2.0}
    x = cu(g["cfg_x"])
    n = x.shape[0]
    t = torch.linspace(0.1, 0.9, n, device="cuda")
    y = m.forward_with_cfg(x, t, cond, scales)
    ref = dit_forward_with_cfg(sd, cfg, x.cpu(), t.cpu(), {k: v.cpu() for k, v in cond.items()}, scales)
    check_err(y.cpu(), ref, TOL_FP32, "forward_with_cfg with a genuinely per-sample t")


@pytest.mark.parametrize("name", ["dit_base", "dit_me2_256"])
def test_forward_with_cfg_uniformity_is_decided_on_device(name):
    """A dense t reaches scldm_dit_forward_cfg with t_stride 2: the device decides between the shared-row plan and the per-sample
    plan (no `.item()` per evaluation).  Many duplicate labels (de-duplicated rows + cell_row) in both plans:
    non-uniform t == the explicit per-sample path bit for bit and matches the oracle; uniform t == the stride-0 path bit for bit;
    no host synchronisation happens inside the call."""
    g, m, cfg, sd = build(name)
    gen = torch.Generator().manual_seed(11)
    B = 24
    x = torch.randn(2 * B, 16, 16, generator=gen).cuda()
    cond = {k: torch.randint(0, min(v, 3), (B,), generator=gen).repeat(2).cuda() for k, v in cfg.class_vocab_sizes.items()}   # <= 3 values per class
    scales = {k: 1.5 + 0.5 * i for i, k in enumerate(cfg.class_vocab_sizes)}
    t = torch.rand(2 * B, generator=gen).cuda()
    m.detect_uniform_t = False
    y_ps = m.forward_with_cfg(x, t, cond, scales)
    m.detect_uniform_t = True
    y_dev = m.forward_with_cfg(x, t, cond, scales)
    assert torch.equal(y_dev, y_ps)
    ref = dit_forward_with_cfg(sd, cfg, x.cpu(), t.cpu(), {k: v.cpu() for k, v in cond.items()}, scales)
    check_err(y_dev.cpu(), ref, TOL_FP32, f"forward_with_cfg {name}: dense non-uniform t, plan chosen on device")
    tu = torch.full((2 * B,), 0.37, device="cuda")
    y_u = m.forward_with_cfg(x, tu, cond, scales)
    assert torch.equal(y_u, m.forward_with_cfg(x, tu[:1].expand(2 * B), cond, scales))
    ref_u = dit_forward_with_cfg(sd, cfg, x.cpu(), tu.cpu(), {k: v.cpu() for k, v in cond.items()}, scales)
    check_err(y_u.cpu(), ref_u, TOL_FP32, f"forward_with_cfg {name}: dense uniform t, plan chosen on device")
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")        # any host synchronisation inside the call now raises
    try:
        y2 = m.forward_with_cfg(x, tu, cond, scales)
        y3 = m.forward_with_cfg(x, t, cond, scales)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert torch.equal(y2, y_u) and torch.equal(y3, y_dev)


@pytest.mark.parametrize("n", [1, 3, 8, 13, 37])
@pytest.mark.parametrize("precision,tol", PARITY)
def test_ragged_batches_vs_oracle(n, precision, tol):
    """Batch sizes that do not fill a 64/128-token tile (tile padding, odd sample pairing)."""
    g, m, cfg, sd = build("dit_base", precision)
    rng = np.random.default_rng(n)
    x = rng.standard_normal((n, 16, 16)).astype(np.float32)
    t = rng.uniform(0, 1, n).astype(np.float32)
    lab = rng.integers(0, 14, n).astype(np.int64)
    ref = dit_forward(sd, cfg, torch.from_numpy(x), torch.from_numpy(t), {"clusters": torch.from_numpy(lab)})
    y = m(cu(x), cu(t), {"clusters": cu(lab)})
    check_err(y.cpu(), ref, tol, f"ragged n={n} [{precision}] vs oracle", FLOOR_TOL[precision])


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


@pytest.mark.parametrize("name,method,steps", [("dit_base", "euler", 5), ("dit_base", "heun", 4), ("dit_joint", "euler", 4),
                                               ("dit_me2_256", "euler", 5), ("dit_me2_256", "heun", 3)])
@pytest.mark.parametrize("precision,tol", PARITY)
def test_fused_sampler_vs_oracle(name, method, steps, precision, tol):
    g, m, cfg, sd = build(name, precision)
    rng = np.random.default_rng(11)
    B = 6
    z0 = rng.standard_normal((B, 16, 16)).astype(np.float32)
    labs = {k: rng.integers(0, v, B).astype(np.int64) for k, v in cfg.class_vocab_sizes.items()}
    scales = {k: 1.5 - 0.4 * i for i, k in enumerate(sorted(cfg.class_vocab_sizes))}   # distinct per class (two passes for dit_me2_256)
    z2 = torch.from_numpy(np.concatenate([z0, z0]))
    cond2 = {k: torch.from_numpy(np.concatenate([v, v])) for k, v in labs.items()}
    ref = sample_ode_fixed(z2, lambda x, t: dit_forward_with_cfg(sd, cfg, x, t, cond2, scales), steps, method)
    out = m.sample_ode_cfg(z2.cuda(), {k: v.cuda() for k, v in cond2.items()}, scales, steps, method)
    check_err(out.cpu(), ref, tol, f"fused sampler {name} {method} x{steps} [{precision}] vs oracle", FLOOR_TOL[precision])
    # the generic reference-style call chain (Sampler -> lambda -> forward_with_cfg) gives the same trajectory end
    from scldm_amd.transport import Sampler, create_transport
    fn = Sampler(create_transport("Linear", "velocity", "velocity", 1e-5, 1e-5)).sample_ode(sampling_method=method, num_steps=steps)
    condc = {k: v.cuda() for k, v in cond2.items()}
    model_fn = lambda x, t, **kw: m.forward_with_cfg(x, t, **kw, cfg_scale=scales)
    traj = fn(z2.cuda(), model_fn, **{"condition": condc})
    assert traj.shape[0] == steps and max_abs_rel(traj[-1].cpu(), ref) < tol
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


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
def test_headline_workload_vs_oracle(prec):
    """VERDICT r5 next #5a: the BENCH workload itself (`dentate_b4096_euler100`: bench.py's own make_model / make_inputs / seed, 4 096
    cells x 100 CFG Euler evaluations = 12 288 sample-forwards per evaluation) against the float32 oracle chain on 8 cells spread over
    the batch.  fp16 (the reference's arithmetic class) within 1.5 x the TF32-operand oracle on the same cells; bf16 (the dtype
    BASELINE.json names, narrower than the reference) <= 3e-2."""
    import bench
    from oracle.dit import matmul_operand_bits
    wl = bench.WORKLOADS["dentate_b4096_euler100"]
    B, steps = wl["B"], wl["evals"] + 1
    m = bench.make_model(wl, prec, torch.device("cuda:0"))
    z2, cond2, scales = bench.make_inputs(wl, B, torch.device("cuda:0"), seed=1234)     # rank 0's inputs in time_workload
    out = m.sample_ode_cfg(z2, cond2, scales, steps, wl["method"])
    assert out.shape == (2 * B, 16, 16) and torch.isfinite(out).all()
    idx = torch.tensor([0, 63, 64, 1000, 2047, 2048, 3333, B - 1])          # tile edges, both halves of the batch, the last cell
    rows = torch.cat([idx, idx + B]).cuda()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    cfg = DiTConfig(class_vocab_sizes=wl["vocab"], condition_strategy=wl["strategy"])
    zs, cs = z2[rows].cpu(), {k: v[rows].cpu() for k, v in cond2.items()}
    f32 = lambda xx, tt: dit_forward_with_cfg(sd, cfg, xx, tt, cs, scales)
    n_thr = torch.get_num_threads()
    torch.set_num_threads(min(16, n_thr))
    try:
        ref = sample_ode_fixed(zs, f32, steps, wl["method"])
        with matmul_operand_bits(10):
            e_tf32 = max_abs_rel(sample_ode_fixed(zs, f32, steps, wl["method"]), ref)
    finally:
        torch.set_num_threads(n_thr)
    err = max_abs_rel(out[rows].cpu(), ref)
    print(f"[parity] headline workload dentate_b4096_euler100 [{prec}] 8 cells of 4096 vs float32 oracle chain: {err:.3e}   "
          f"TF32-operand oracle {e_tf32:.3e}")
    if prec == "fp16":
        assert err <= TF32_FACTOR * e_tf32, (err, e_tf32)
    else:
        assert err < TOL_BF16, err


@pytest.mark.parametrize("vocab,strategy,B,method,steps,scale", [
    ({"cell_type": 50}, "mutually_exclusive", 2048, "heun", 101, 2.0),               # BASELINE configs[2] (hlca): 100 Heun steps
    ({"cell_type": 18, "cytokine": 91}, "joint", 1024, "euler", 101, 1.0),            # configs[3] per-GPU shard (parse1m): 100 Euler
])
def test_fused_sampler_full_size_properties(vocab, strategy, B, method, steps, scale):
    """At BASELINE batch sizes AND trajectory lengths (100 Heun steps = 200 CFG evaluations at guidance 2.0; 100 Euler evaluations):
    the fused bf16 sampler is bit-repeatable, sharding the batch gives the same cells (cells are independent), and a handful of
    cells agree with the oracle chain."""
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


_DOPRI5_ORACLE: dict = {}


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_dopri5_sampler_matches_float64_oracle(precision):
    """The reference's DEFAULT sampler (`sample_ode()` -> dopri5, 50 save points, atol = rtol = 1e-5: transport.py:324-331,
    models.py:793-812): the host-driven stepper over the fused forward_with_cfg against oracle/transport.py's float64
    restatement of torchdiffeq's driver running oracle.dit in float64, 4 cells (8 rows), whole 50-point trajectories.

    What can be asserted at which tolerance (measured on the CPU with bit-faithful fp32 arithmetic too):
      * at atol = rtol <= 1e-6 the solver's own global error is below the parity gate, and product == oracle at the SAME
        tolerance within 1e-4 (measured 3e-5 at 1e-6, 2e-6 at 1e-7);
      * at the default 1e-5 the float64 oracle itself is 2.0e-4 away from the converged solution (oracle at 1e-10) and so is
        any fp32 run of the same algorithm: the first step's error estimate (h0 ~ 0.045, true error ~1e-9) is below fp32
        rounding noise, the controller's growth factor after it is anywhere in [5, 10], and the 4-5 large steps that follow
        land 1-2e-4 apart.  There the assertion is 5e-4 against the oracle AND against the converged solution - the solver's
        tolerance, not an arithmetic difference."""
    from oracle.transport import sample_ode_dopri5
    from scldm_amd.transport import Sampler, create_transport
    g, m, cfg, sd = build("dit_base", precision)
    sd64 = {k: v.double() for k, v in sd.items()}
    gen = torch.Generator().manual_seed(4)
    B = 4
    z0 = torch.randn(B, 16, 16, generator=gen)
    z2 = torch.cat([z0, z0])
    cond = {"clusters": torch.randint(0, 14, (B,), generator=gen).repeat(2)}
    scales = {"clusters": 2.0}
    f64 = lambda x, t: dit_forward_with_cfg(sd64, cfg, x, t, cond, scales)
    condg = {k: v.cuda() for k, v in cond.items()}
    tr = create_transport("Linear", "velocity", "velocity", 1e-5, 1e-5)

    def oracle(tol):
        """float64 oracle solve at tolerance `tol` (None: the defaults) - identical for both precisions of this test: solved once per
        session (round 5: the four float64 CPU solves, twice, were 30 % of the GPU suite's wall time), on a 16-thread team."""
        if tol not in _DOPRI5_ORACLE:
            n_thr = torch.get_num_threads()
            torch.set_num_threads(min(16, n_thr))
            try:
                _DOPRI5_ORACLE[tol] = sample_ode_dopri5(z2, f64, return_stats=True) if tol is None else sample_ode_dopri5(z2, f64, 50, tol, tol, return_stats=True)
            finally:
                torch.set_num_threads(n_thr)
        return _DOPRI5_ORACLE[tol]
    for tol in (1e-6, 1e-7):
        ref, st = oracle(tol)
        fn = Sampler(tr).sample_ode(sampling_method="dopri5", num_steps=50, atol=tol, rtol=tol)
        traj = fn(z2.cuda(), m.forward_with_cfg, condition=condg, cfg_scale=scales)
        ls = fn.last_stats
        print(f"[dopri5 tol {tol:g}] oracle {len(st['accepted'])} accepted / {len(st['rejected'])} rejected, {st['evaluations']} evaluations; "
              f"product [{precision}] {len(ls['accepted_steps'])} / {len(ls['rejected_steps'])}, {ls['evaluations']}")
        assert traj.shape == (50, 2 * B, 16, 16) and torch.equal(traj[0].cpu(), z2)
        # (floored elementwise bound 5e-3: solver-tolerance-level differences land on elements at 1 % of the trajectory's scale)
        check_err(traj.cpu(), ref.float(), TOL_FP32, f"dopri5 trajectory, atol = rtol = {tol:g} [{precision}] vs float64 oracle", 5e-3)
        end = m.sample_ode_cfg(z2.cuda(), condg, scales, 2, "dopri5", atol=tol, rtol=tol)     # same steps; only the save times differ
        check_err(end.cpu(), ref[-1].float(), TOL_FP32, f"sample_ode_cfg dopri5, atol = rtol = {tol:g} [{precision}] vs float64 oracle", 5e-3)
    truth = oracle(1e-10)[0]
    ref, st = oracle(None)
    fn = Sampler(tr).sample_ode()                                                          # the reference's default call
    traj = fn(z2.cuda(), m.forward_with_cfg, condition=condg, cfg_scale=scales)
    ls = fn.last_stats
    print(f"[dopri5 default] oracle {len(st['accepted'])} / {len(st['rejected'])}, product [{precision}] {len(ls['accepted_steps'])} / "
          f"{len(ls['rejected_steps'])}; oracle vs converged {max_abs_rel(ref.float(), truth.float()):.2e}")
    assert traj.shape == (50, 2 * B, 16, 16) and 20 <= ls["evaluations"] <= 80
    (ta, ha), (tb, hb) = ls["accepted_steps"][0], st["accepted"][0]
    assert ta == tb == 0.0 and abs(ha - hb) < 1e-3 * hb                                  # same automatic initial step
    check_err(traj.cpu(), ref.float(), 5e-4, f"dopri5 default call [{precision}] vs float64 oracle (solver tolerance)", 5e-2)
    check_err(traj.cpu(), truth.float(), 5e-4, f"dopri5 default call [{precision}] vs converged solution (solver tolerance)", 5e-2)


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
        for knob, val in (("SCLDM_GROUPS", "2"), ("SCLDM_LPL", "1"), ("SCLDM_LPL", "3")):
            monkeypatch.setenv(knob, val)                  # run-time knobs are read once, when the native handle is created
            _, m2, _, _ = build("dit_base", "bf16")
            y = m2(x, t, {"clusters": lab})
            torch.cuda.synchronize()
            monkeypatch.delenv(knob)
            assert torch.equal(y, ref), (knob, val)


@pytest.mark.parametrize("case,precision", [("dit_base", "bf16"), ("dit_base", "fp32"), ("dit_joint", "bf16"), ("dit_me2_256", "fp16")])
def test_whole_solve_conditioning_is_bit_identical_to_per_evaluation_conditioning(case, precision, monkeypatch):
    """scldm_sample_ode prepares the adaLN vectors of EVERY evaluation of a fixed-grid solve in one batched pass ahead of the loop (they
    depend on t and the labels only) instead of two launches per evaluation: same kernels over more rows, so the trajectories must be the
    per-evaluation path's (SCLDM_COND_ALL=0) bit for bit - Euler and Heun, few rows (<= 128: row-tile projection) and many distinct label
    tuples (the 128 x 128-block projection), single evaluation (nothing to batch)."""
    g, m, cfg, sd = build(case, precision)
    monkeypatch.setenv("SCLDM_COND_ALL", "0")                # (read once, when the native handle is created)
    _, m0, _, _ = build(case, precision)
    monkeypatch.delenv("SCLDM_COND_ALL")
    vocab = cfg.class_vocab_sizes
    gen = torch.Generator(device="cuda").manual_seed(5)
    with torch.no_grad():
        for B, steps, method in ((40, 7, "euler"), (40, 5, "heun"), (300, 4, "euler"), (3, 2, "euler")):
            z0 = torch.randn(B, 16, 16, device="cuda", generator=gen)
            z2 = torch.cat([z0, z0])
            cond = {k: torch.randint(0, v, (B,), device="cuda", generator=gen).repeat(2) for k, v in vocab.items()}
            scales = {k: 1.5 for k in vocab}
            a = m.sample_ode_cfg(z2, cond, scales, steps, method)
            b = m0.sample_ode_cfg(z2, cond, scales, steps, method)
            assert torch.isfinite(a).all() and torch.equal(a, b), (B, steps, method)


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_small_launches_on_32_token_tiles_are_bit_identical(precision, monkeypatch):
    """Launches of at most 256 32-token tiles (512 samples) run the 32-token-tile instantiation of the fused kernel (a shorter walk
    while every tile still has a CU to itself).  Same weight stream, same arithmetic per token: the bytes must not depend on it -
    a cell's result may not change with the size of the batch it is sampled in."""
    g, m, cfg, sd = build("dit_base", precision)
    monkeypatch.setenv("SCLDM_SMALL_NTT", "0")             # (read once, when the native handle is created)
    _, m64, _, _ = build("dit_base", precision)
    monkeypatch.delenv("SCLDM_SMALL_NTT")
    gen = torch.Generator(device="cuda").manual_seed(31)
    with torch.no_grad():
        for n in (1, 5, 130, 512):
            x = torch.randn(n, 16, 16, device="cuda", generator=gen)
            t = torch.rand(n, device="cuda", generator=gen)
            lab = {"clusters": torch.randint(0, 14, (n,), device="cuda", generator=gen)}
            assert torch.equal(m(x, t, lab), m64(x, t, lab)), n
        n = 600                                              # above the threshold: rows of a big launch == the same rows launched alone
        x = torch.randn(n, 16, 16, device="cuda", generator=gen)
        t = torch.rand(n, device="cuda", generator=gen)
        lab = {"clusters": torch.randint(0, 14, (n,), device="cuda", generator=gen)}
        whole = m(x, t, lab)
        part = m(x[:100], t[:100], {"clusters": lab["clusters"][:100]})
        assert torch.equal(whole[:100], part)


# --------------------------------------------------------------------------------------------------------------------
# round-2 additions: configurations and boundary behaviour the round-1 review found untested
# --------------------------------------------------------------------------------------------------------------------
def test_dentate_b512_euler50_bf16_properties():
    """BASELINE.json configs[1] (dentate_gyrus, batch 512, 50 Euler evaluations, bf16) at full size: finite, bit-repeatable,
    split-batch == whole batch, the unconditional half does not depend on the labels, and 8 cells agree with the oracle chain."""
    from scldm_amd.sampling import sample_latents
    g, m, cfg, sd = build("dit_base", "bf16")
    gen = torch.Generator(device="cuda").manual_seed(2)
    B = 512
    z0 = torch.randn(B, 16, 16, device="cuda", generator=gen)
    cond = {"clusters": torch.randint(0, 14, (B,), device="cuda", generator=gen)}
    scales = {"clusters": 1.0}
    out = sample_latents(m, z0, cond, scales, 51, "euler")
    assert out.shape == (2 * B, 16, 16) and torch.isfinite(out).all()
    assert torch.equal(out, sample_latents(m, z0, cond, scales, 51, "euler"))
    half = sample_latents(m, z0[:200], {"clusters": cond["clusters"][:200]}, scales, 51, "euler")
    assert torch.equal(half[:200], out[:200]) and torch.equal(half[200:], out[B:B + 200])
    other = sample_latents(m, z0, {"clusters": (cond["clusters"] + 3) % 14}, scales, 51, "euler")
    assert torch.equal(other[:B], out[:B]) and not torch.equal(other[B:], out[B:])
    idx = torch.arange(0, B, 64, device="cuda")
    z2 = torch.cat([z0[idx], z0[idx]]).cpu()
    c2 = {"clusters": torch.cat([cond["clusters"][idx]] * 2).cpu()}
    ref = sample_ode_fixed(z2, lambda x, t: dit_forward_with_cfg(sd, cfg, x, t, c2, scales), 51, "euler")
    got = torch.cat([out[idx], out[idx + B]]).cpu()
    check_err(got, ref, TOL_BF16, "dentate b512 x 50 Euler [bf16] vs oracle chain (8 cells)")


def test_dit_l_depth_parity_fp32():
    """A DiT-L (1024 wide, 24 layers, 16 heads - BASELINE.json configs[4] names it; not a reference config) on the
    generic HIP path: fp32 parity of forward and forward_with_cfg on 3 cells at full depth."""
    from scldm_amd.nnets import DiT
    vocab = {"cell_line": 4, "gene": 2024}
    kw = dict(n_embed=1024, n_embed_input=16, n_layer=24, n_head=16, seq_len=16, dropout=0.0, bias=True, norm_layer="layernorm",
              multiple_of=4, layernorm_eps=1e-8, class_vocab_sizes=vocab, cfg_dropout_prob=0.8, condition_strategy="joint")
    m = DiT(**kw)
    sd = make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 77)
    sd = {k: (v * 0.4 if k.endswith("weight") and v.dim() == 2 and v.shape[1] >= 1024 else v) for k, v in sd.items()}   # keep 24 layers O(1)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    cfg = DiTConfig(n_embed=1024, n_layer=24, n_head=16, class_vocab_sizes=vocab, condition_strategy="joint")
    gen = torch.Generator().manual_seed(9)
    x, t = torch.randn(3, 16, 16, generator=gen), torch.rand(3, generator=gen)
    cond = {"cell_line": torch.randint(0, 4, (3,), generator=gen), "gene": torch.randint(0, 2024, (3,), generator=gen)}
    with torch.no_grad():
        y = m(x.cuda(), t.cuda(), {k: v.cuda() for k, v in cond.items()})
    check_err(y.cpu(), dit_forward(sd, cfg, x, t, cond), TOL_FP32, "DiT-L (24 layers) forward fp32 vs oracle")
    x2, t2 = torch.cat([x, x]), torch.full((6,), 0.3)
    c2 = {k: torch.cat([v, v]) for k, v in cond.items()}
    scales = {"cell_line": 1.5, "gene": 2.5}
    with torch.no_grad():
        y2 = m.forward_with_cfg(x2.cuda(), t2.cuda(), {k: v.cuda() for k, v in c2.items()}, scales)
    check_err(y2.cpu(), dit_forward_with_cfg(sd, cfg, x2, t2, c2, scales), TOL_FP32, "DiT-L forward_with_cfg fp32 vs oracle")


def test_model_without_null_rows():
    """cfg_dropout_prob == 0: the class tables have `vocab` rows (nnets.py:241-243).  forward with full labels works and is
    exact; everything that needs a null token raises (the reference's nn.Embedding raises IndexError there) instead of
    reading past the table."""
    from scldm_amd.nnets import DiT
    kw = dict(n_embed=256, n_embed_input=16, n_layer=2, n_head=8, seq_len=16, dropout=0.0, bias=True, norm_layer="layernorm",
              multiple_of=4, layernorm_eps=1e-8, class_vocab_sizes={"clusters": 14}, cfg_dropout_prob=0.0)
    m = DiT(**kw)
    assert m.class_embeddings["clusters"].weight.shape == (14, 256)
    sd = make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 88)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    cfg = DiTConfig(n_layer=2, class_vocab_sizes={"clusters": 14})
    gen = torch.Generator().manual_seed(3)
    x, t = torch.randn(5, 16, 16, generator=gen), torch.rand(5, generator=gen)
    lab = torch.tensor([0, 13, 5, 13, 7])
    y = m(x.cuda(), t.cuda(), {"clusters": lab.cuda()})
    check_err(y.cpu(), dit_forward(sd, cfg, x, t, {"clusters": lab}), TOL_FP32, "forward without null rows")
    with pytest.raises(IndexError):
        m.forward_with_cfg(torch.cat([x, x]).cuda(), torch.full((10,), 0.5).cuda(), {"clusters": torch.cat([lab, lab]).cuda()}, {"clusters": 1.0})
    with pytest.raises(IndexError):
        m.sample_ode_cfg(torch.cat([x, x]).cuda(), {"clusters": torch.cat([lab, lab]).cuda()}, {"clusters": 1.0}, 3, "euler")
    # training path: gradients of the 14-row table land inside it (the round-1 kernel wrote a 15th row)
    m.train()
    xg = x.cuda()
    pred = m(xg, t.cuda(), {"clusters": lab.cuda()}, force_drop_ids=False)
    guard = torch.full((4096,), 7.0, device="cuda")   # likely neighbour of the gradient allocation
    pred.square().mean().backward()
    gtab = m.class_embeddings["clusters"].weight.grad
    assert gtab.shape == (14, 256) and torch.isfinite(gtab).all() and bool((guard == 7.0).all())
    from oracle.train import training_grads  # noqa: F401  (gradient parity itself: tests/test_gpu_train.py)
    with pytest.raises(IndexError):
        m(xg, t.cuda(), {"clusters": lab.cuda()}, force_drop_ids=True)   # label dropout needs the null row


def test_out_of_range_labels_are_reported_not_read():
    g, m, cfg, sd = build("dit_base")
    x, t = cu(g["cfg_x"]), cu(g["cfg_t"])
    n = x.shape[0]
    bad = torch.full((n,), 99, dtype=torch.long, device="cuda")
    with pytest.raises(IndexError):
        m.sample_ode_cfg(x, {"clusters": bad}, {"clusters": 1.0}, 3, "euler")          # caught host-side with the de-duplicated rows
    y = m(x, t, {"clusters": bad})                                                       # plain forward: clamped on device, counted
    assert torch.isfinite(y).all()
    with pytest.raises(IndexError):
        m.check_labels()
    assert m.check_labels() == 0                                                         # the counter was reset


def test_deepcopy_and_pickle_after_forward():
    """ema_pytorch deep-copies the model (reference models.py:446); after a first forward the module holds a native handle,
    which must not travel (VERDICT r1 robustness #12)."""
    import copy
    import pickle
    g, m, cfg, sd = build("dit_base")
    cond = {k: cu(g[f"fwd_label_{k}"]) for k in golden_json(g, "fwd_classes")}
    x, t = cu(g["fwd_x"]), cu(g["fwd_t"])
    y = m(x, t, cond)
    m2 = copy.deepcopy(m)
    m3 = pickle.loads(pickle.dumps(m))
    assert m2._handle is None and m3._handle is None and m._handle is not None
    assert torch.equal(m2(x, t, cond), y) and torch.equal(m3.cuda()(x, t, cond), y) and torch.equal(m(x, t, cond), y)
    with torch.no_grad():                      # the copies are independent models
        for p in m2.parameters():
            p.mul_(1.01)
    assert not torch.equal(m2(x, t, cond), y) and torch.equal(m(x, t, cond), y)


@pytest.mark.parametrize("precision", ["fp32", "bf16", "bf16x3"])
def test_inplace_data_updates_are_picked_up(precision):
    """EMA-style updates through `.data` change neither the storage nor torch's version counter: the fused path
    must still see them (device-side fingerprint in scldm_dit_refresh_weights), for every packed precision."""
    g, m, cfg, sd = build("dit_base", precision)
    cond = {k: cu(g[f"fwd_label_{k}"]) for k in golden_json(g, "fwd_classes")}
    x, t = cu(g["fwd_x"]), cu(g["fwd_t"])
    y0 = m(x, t, cond)
    versions = [p._version for p in m.parameters()]
    other = make_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 999)
    for k, p in m.state_dict().items():
        p.data.lerp_(other[k].cuda(), 0.5)       # what ema_pytorch does
    assert [p._version for p in m.parameters()] == versions
    y1 = m(x, t, cond)
    assert not torch.equal(y1, y0)
    sd2 = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    tol = TOL_BF16 if precision == "bf16" else TOL_FP32
    check_err(y1.cpu(), dit_forward(sd2, cfg, x.cpu(), t.cpu(), {k: v.cpu() for k, v in cond.items()}), tol, f"after .data update [{precision}]", FLOOR_TOL[precision])
    m.blocks[3].mlp.w1.weight.data[5, 7] += 1.0     # a single element of a large tensor: the fingerprint covers EVERY element
    y2 = m(x, t, cond)
    assert not torch.equal(y2, y1)
    m.class_embeddings["clusters"].weight.data[int(cond["clusters"][0])] *= 1.5     # a partial `.data` write to a class-table row
    y3 = m(x, t, cond)
    assert not torch.equal(y3, y2)
    m.invalidate_weights()                            # forcing a re-pack changes nothing once the copies are current
    assert torch.equal(m(x, t, cond), y3)


@pytest.mark.parametrize("precision", ["bf16", "fp16", "bf16x3", "fp32"])
@pytest.mark.parametrize("knobs", [{}, {"SCLDM_LPL": "1"}, {"SCLDM_LPL": "2"}, {"SCLDM_LPL": "3"}, {"SCLDM_FT": "1", "SCLDM_X3_FT": "1"},
                                   {"SCLDM_X3_NTT": "1"}])
def test_bit_repeatability_across_shapes_and_launch_groupings(precision, knobs, monkeypatch):
    """run-to-run differences were once bisected to codegen (packed f32 math in the LayerNorm sweep).  Every precision,
    every kernel shape and every layers-per-launch grouping must give the same bytes on repeated runs with two workgroups per CU
    busy (n = 2049: ragged last tile), and the grouping must not change the bytes at all (floating-point contraction is off in the
    fused kernel, so a layer computes the same values whichever slot of a launch it runs in)."""
    g, m0, cfg, sd = build("dit_base", precision)
    gen = torch.Generator(device="cuda").manual_seed(77)
    n = 2049
    x = torch.randn(n, 16, 16, device="cuda", generator=gen)
    t = torch.rand(n, device="cuda", generator=gen)
    lab = {"clusters": torch.randint(0, 14, (n,), device="cuda", generator=gen)}
    with torch.no_grad():
        ref = m0(x, t, lab)
        for k, v in knobs.items():
            monkeypatch.setenv(k, v)
        _, m, _, _ = build("dit_base", precision)
        ys = [m(x, t, lab) for _ in range(4)]
        torch.cuda.synchronize()
    assert all(torch.equal(y, ys[0]) for y in ys[1:])
    if not any(k in knobs for k in ("SCLDM_FT", "SCLDM_X3_FT", "SCLDM_X3_NTT")) or (precision != "bf16x3" and "SCLDM_X3_NTT" in knobs):
        assert torch.equal(ys[0], ref)            # same kernel shape, other grouping: identical
    else:
        tol = TOL_BF16 if precision in ("bf16", "fp16") else TOL_FP32
        assert max_abs_rel(ys[0].cpu(), ref.cpu()) < tol   # another reduction tree: close, not identical
