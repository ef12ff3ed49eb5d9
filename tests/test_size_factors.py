"""Size-factor draw: oracle and device sampler against outputs of the reference method itself
(tests/golden/size_factors.npz).  The table logic runs on any torch device, so the CPU suite covers it; the GPU test adds
the statistical check of the draw at a BASELINE batch size."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import golden_json, load_golden
from oracle.size_factors import sample_log_size_factors


def _encoder(attrs):
    a = dict(attrs)
    for k in ("mu_size_factor", "sd_size_factor"):
        if k in a:   # JSON turned the integer class indices of the independent maps into strings
            a[k] = {key: {(int(c) if c.lstrip("-").isdigit() else c): v for c, v in m.items()} for key, m in a[k].items()}
    return SimpleNamespace(**a)


def _cases():
    g = load_golden("size_factors")
    return g, golden_json(g, "cases_json")


@pytest.mark.parametrize("name", ["independent_key", "inferred_key", "joint", "no_stats", "no_matching_key"])
def test_oracle_and_table_sampler_match_reference(name):
    from scldm_amd.sampling import SizeFactorSampler
    g, cases = _cases()
    strategy, attrs, cond = cases[name]
    enc = _encoder(attrs)
    B = len(next(iter(cond.values())))
    ref = g[f"out_{name}"]
    assert np.array_equal(sample_log_size_factors(enc, strategy, {k: np.asarray(v) for k, v in cond.items()}, B), ref)
    smp = SizeFactorSampler(enc, strategy, "cpu")
    out = smp.sample({k: torch.tensor(v) for k, v in cond.items()}, B, eps=torch.zeros(B))
    assert np.array_equal(out.numpy(), ref)
    assert np.array_equal(smp.sample(None, 4).numpy(), g["out_condition_none"])


def test_draw_uses_mean_and_std_per_cell():
    from scldm_amd.sampling import SizeFactorSampler
    enc = SimpleNamespace(size_factor_condition_key="ct", mu_size_factor={"ct": {0: 7.0, 1: 9.0}}, sd_size_factor={"ct": {0: 0.5, 1: 0.25}})
    lab = torch.tensor([0, 1, 1, 5, 0])
    eps = torch.tensor([1.0, -2.0, 0.0, 3.0, -1.0])
    out = SizeFactorSampler(enc, "mutually_exclusive", "cpu").sample({"ct": lab}, 5, eps=eps)
    assert torch.equal(out, torch.tensor([7.5, 8.5, 9.0, 0.0, 6.5]))
    ora = sample_log_size_factors(enc, "mutually_exclusive", {"ct": lab.numpy()}, 5, eps.numpy())
    assert np.array_equal(out.numpy(), ora)
    with pytest.raises(ValueError, match="must match batch size"):
        SizeFactorSampler(enc, "mutually_exclusive", "cpu").sample({"ct": lab}, 4)


@pytest.mark.gpu
def test_device_draw_statistics_at_batch_8192():
    from scldm_amd.sampling import SizeFactorSampler
    toks = {f"{i}_{j}": f"c{i}_p{j}" for i in range(18) for j in range(91) if (i + j) % 7}      # parse1m-sized joint space, some missing
    rng = np.random.default_rng(0)
    mu = {t: float(rng.uniform(6, 10)) for t in toks.values()}
    sd = {t: float(rng.uniform(0.1, 0.6)) for t in toks.values()}
    enc = SimpleNamespace(joint_key="ct_cyt", joint_components=["cell_type", "cytokine"], joint_idx_2_classes=toks,
                          mu_size_factor={"ct_cyt": mu}, sd_size_factor={"ct_cyt": sd})
    smp = SizeFactorSampler(enc, "joint", "cuda")
    B = 8192
    gen = torch.Generator(device="cuda").manual_seed(1)
    cond = {"cell_type": torch.randint(0, 18, (B,), device="cuda", generator=gen), "cytokine": torch.randint(0, 91, (B,), device="cuda", generator=gen)}
    eps = torch.randn(B, device="cuda", generator=gen)
    out = smp.sample(cond, B, eps=eps)
    ref = sample_log_size_factors(enc, "joint", {k: v.cpu().numpy() for k, v in cond.items()}, B, eps.cpu().numpy())
    assert np.allclose(out.cpu().numpy(), ref, rtol=0, atol=1e-6)
    # the draw itself: one well-populated class, 200k cells
    one = {"cell_type": torch.full((200_000,), 3, device="cuda"), "cytokine": torch.full((200_000,), 5, device="cuda")}
    d = smp.sample(one, 200_000, generator=gen)
    m, s = mu[toks["3_5"]], sd[toks["3_5"]]
    assert abs(float(d.mean()) - m) < 5 * s / np.sqrt(200_000) and abs(float(d.std()) - s) < 0.01 * s
