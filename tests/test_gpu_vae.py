"""GPU parity of the MCAB encode / decode kernels (fp32) against the reference's golden vectors and the oracle.
Tolerance: 1e-4 scale-relative max error (BASELINE.json north_star gate)."""
import numpy as np
import pytest
import torch

from conftest import check_err, golden_json, load_golden, max_abs_rel
from oracle.vae import VAEConfig, decode, encode
from oracle.weights import make_state_dict
from test_abi_cpu import _build_vae

pytestmark = pytest.mark.gpu
TOL = 1e-4


def build(name):
    g = load_golden(name)
    shapes = {k: tuple(v) for k, v in golden_json(g, "shapes_json").items()}
    sd = make_state_dict(shapes, int(g["seed"]))
    vae = _build_vae(int(g["n_genes"]), shared_theta="decoder_head.theta.weight" in shapes)
    vae.load_state_dict(sd, strict=True)
    return g, vae.cuda().eval(), sd, VAEConfig(n_genes=int(g["n_genes"]))


def cu(a):
    return torch.from_numpy(np.asarray(a)).cuda()


@pytest.mark.parametrize("name", ["vae_small", "vae_2000", "vae_unshared"])
def test_encode_decode_match_reference_golden(name):
    g, vae, sd, cfg = build(name)
    z = vae.encode(cu(g["counts"]), cu(g["genes"]), cu(g["counts_subset"]), cu(g["genes_subset"]))
    assert z.shape == g["z"].shape and max_abs_rel(z.cpu(), g["z"]) < TOL
    nb = vae.decode(cu(g["z"]), cu(g["genes"]), cu(g["library_size"]))
    assert max_abs_rel(nb.mu.cpu(), g["mu"]) < TOL and max_abs_rel(nb.theta.cpu(), g["theta"]) < 1e-5
    nb2 = vae.decode(cu(g["zrand"]), cu(g["genes"]), cu(g["library_size"]))
    assert max_abs_rel(nb2.mu.cpu(), g["mu_rand"]) < TOL
    assert torch.allclose(nb.mu.sum(1).cpu(), torch.from_numpy(g["library_size"][:, 0]), rtol=1e-4)
    import contextlib
    # (with gradients enabled forward() is the differentiable route, which is built for the shared-theta head only)
    with torch.no_grad() if name == "vae_unshared" else contextlib.nullcontext():
        params, z2 = vae(cu(g["counts"]), cu(g["genes"]), cu(g["library_size"]), cu(g["counts_subset"]), cu(g["genes_subset"]))
    assert torch.equal(z2, z) and set(params) == {"mu", "theta"}


@pytest.mark.parametrize("B,S,G", [(1, 1, 1), (3, 31, 33), (2, 64, 1025), (5, 100, 4099)])
def test_ragged_sizes_vs_oracle(B, S, G):
    """Sizes that do not fill 32-gene tiles / 1024-gene chunks; S=1 and G=1 edge cases."""
    g, vae, sd, cfg = build("vae_2000")
    rng = np.random.default_rng(B * 1000 + S + G)
    genes_s = rng.integers(0, 2000, (B, S)).astype(np.int64)
    counts_s = rng.poisson(1.5, (B, S)).astype(np.float32)
    genes = np.stack([rng.permutation(2001)[:G] if G <= 2001 else rng.integers(0, 2001, G) for _ in range(B)]).astype(np.int64)
    lib = rng.uniform(100, 2000, (B, 1)).astype(np.float32)
    zr = rng.standard_normal((B, 16, 16)).astype(np.float32)
    z_ref = encode(sd, cfg, torch.from_numpy(counts_s), torch.from_numpy(genes_s))
    z = vae.encode(cu(counts_s), cu(genes_s))
    assert max_abs_rel(z.cpu(), z_ref) < TOL
    mu_ref, th_ref = decode(sd, cfg, torch.from_numpy(zr), torch.from_numpy(genes), torch.from_numpy(lib))
    nb = vae.decode(cu(zr), cu(genes), cu(lib))
    assert max_abs_rel(nb.mu.cpu(), mu_ref) < TOL and max_abs_rel(nb.theta.cpu(), th_ref) < 1e-5


def test_full_size_properties_dentate():
    """dentate_gyrus shape (G=17002, S=6147): mu rows sum to the library size, results are repeatable bit for bit,
    permuting a cell's gene list permutes its mu, and a few cells agree with the oracle."""
    G, S, B = 17002, 6147, 8
    g, vae, sd, cfg = build("vae_2000")
    from test_abi_cpu import _build_vae as bv
    shapes = {k: tuple(v.shape) for k, v in bv(G).state_dict().items()}
    sd = make_state_dict(shapes, 77)
    vae = bv(G)
    vae.load_state_dict(sd, strict=True)
    vae = vae.cuda().eval()
    cfg = VAEConfig(n_genes=G)
    rng = np.random.default_rng(3)
    genes = np.tile(np.arange(G, dtype=np.int64), (B, 1))
    counts = rng.poisson(0.7, (B, G)).astype(np.float32)
    sub = np.stack([np.sort(rng.permutation(G)[:S]) for _ in range(B)])
    lib = counts.sum(1, keepdims=True) + 1
    z = vae.encode(cu(np.take_along_axis(counts, sub, 1)), cu(np.take_along_axis(genes, sub, 1)))
    nb = vae.decode(z, cu(genes), cu(lib))
    nb_again = vae.decode(z, cu(genes), cu(lib))
    assert torch.equal(nb.mu, nb_again.mu)
    assert torch.allclose(nb.mu.sum(1), cu(lib[:, 0]), rtol=2e-4)
    perm = rng.permutation(G)
    nb_p = vae.decode(z, cu(genes[:, perm]), cu(lib))
    assert max_abs_rel(nb_p.mu.cpu(), nb.mu.cpu()[:, perm]) < 1e-5   # same values up to the softmax-sum order
    z_ref = encode(sd, cfg, torch.from_numpy(np.take_along_axis(counts, sub, 1))[:2], torch.from_numpy(np.take_along_axis(genes, sub, 1))[:2])
    assert max_abs_rel(z.cpu()[:2], z_ref) < TOL
    mu_ref, _ = decode(sd, cfg, z.cpu()[:2], torch.from_numpy(genes[:2]), torch.from_numpy(lib[:2]))
    assert max_abs_rel(nb.mu.cpu()[:2], mu_ref) < TOL
    x = nb.sample()
    assert x.shape == (B, G) and (x >= 0).all()


def test_sample_cells_harness_matches_oracle_chain():
    """S1: noise -> fused CFG Euler sampling -> decode, against oracle DiT + oracle transport + oracle decode."""
    from oracle.dit import DiTConfig, dit_forward_with_cfg
    from oracle.transport import sample_ode_fixed
    from scldm_amd.nnets import DiT
    from scldm_amd.sampling import sample_cells
    g, vae, sd_v, cfg_v = build("vae_2000")
    gd = load_golden("dit_base")
    kw = golden_json(gd, "kwargs_json")
    shapes = {k: tuple(v) for k, v in golden_json(gd, "shapes_json").items()}
    sd_d = make_state_dict(shapes, int(gd["seed"]))
    dit = DiT(**kw)
    dit.load_state_dict(sd_d, strict=True)
    dit = dit.cuda().eval()
    cfg_d = DiTConfig(class_vocab_sizes=kw["class_vocab_sizes"], condition_strategy=kw["condition_strategy"])
    rng = np.random.default_rng(9)
    B, G = 3, 300
    z0 = rng.standard_normal((B, 16, 16)).astype(np.float32)
    lab = rng.integers(0, 14, B).astype(np.int64)
    genes = np.stack([rng.permutation(2000)[:G] for _ in range(B)]).astype(np.int64)
    logsf = rng.normal(7.0, 0.3, B).astype(np.float32)
    scales = {"clusters": 2.0}
    nb, z = sample_cells(dit, vae, {"clusters": cu(lab)}, scales, B, cu(genes), cu(logsf), num_steps=4, sampling_method="euler",
                         z0=cu(z0), draw_counts=False)
    z2 = torch.from_numpy(np.concatenate([z0, z0]))
    cond2 = {"clusters": torch.from_numpy(np.concatenate([lab, lab]))}
    z_ref = sample_ode_fixed(z2, lambda x, t: dit_forward_with_cfg(sd_d, cfg_d, x, t, cond2, scales), 4, "euler")
    lib = torch.exp(torch.from_numpy(np.concatenate([logsf, logsf]))).view(-1, 1)
    mu_ref, th_ref = decode(sd_v, cfg_v, z_ref, torch.from_numpy(np.concatenate([genes, genes])), lib)
    assert z.shape == (2 * B, 16, 16) and max_abs_rel(z.cpu(), z_ref) < TOL
    assert max_abs_rel(nb.mu.cpu(), mu_ref) < 2e-4 and max_abs_rel(nb.theta.cpu(), th_ref) < 1e-5
    with pytest.raises(ValueError):
        sample_cells(dit, vae, {"clusters": cu(lab)}, scales, B + 1, cu(genes), cu(logsf))
    with pytest.raises(AssertionError):
        sample_cells(dit, vae, {"clusters": cu(lab)}, {"other": 1.0}, B, cu(genes), cu(logsf))


def test_generation_chain_with_device_size_factors_matches_oracle_chain():
    """The end-to-end chain bench.py times as `generation_end_to_end` (reference: LatentDiffusion.predict_step -> sample,
    src/scldm/models.py:707-819): SizeFactorSampler.sample -> sample_cells (fused CFG Euler + MCAB decode) at 8 cells, (mu, theta)
    and latents against the oracle chain (oracle size factors with the same eps, oracle DiT + transport, oracle decode); then the
    counts drawn by the fused decoder go through dense_to_csr and come back as the arrays scipy builds from the same matrix."""
    from types import SimpleNamespace
    import scipy.sparse as sp
    from oracle.dit import DiTConfig, dit_forward_with_cfg
    from oracle.size_factors import sample_log_size_factors
    from oracle.transport import sample_ode_fixed
    from scldm_amd.datamodule import dense_to_csr
    from scldm_amd.nnets import DiT
    from scldm_amd.sampling import SizeFactorSampler, sample_cells
    g, vae, sd_v, cfg_v = build("vae_2000")
    gd = load_golden("dit_base")
    kw = golden_json(gd, "kwargs_json")
    shapes = {k: tuple(v) for k, v in golden_json(gd, "shapes_json").items()}
    sd_d = make_state_dict(shapes, int(gd["seed"]))
    dit = DiT(**kw)
    dit.load_state_dict(sd_d, strict=True)
    dit = dit.cuda().eval()
    cfg_d = DiTConfig(class_vocab_sizes=kw["class_vocab_sizes"], condition_strategy=kw["condition_strategy"])
    rng = np.random.default_rng(17)
    B, G = 8, 257
    enc = SimpleNamespace(size_factor_condition_key="clusters", mu_size_factor={"clusters": {i: 6.5 + 0.2 * i for i in range(13)}},
                          sd_size_factor={"clusters": {i: 0.1 + 0.02 * i for i in range(13)}})        # label 13 has no statistics -> 0
    z0 = rng.standard_normal((B, 16, 16)).astype(np.float32)
    lab = np.array([0, 3, 13, 7, 12, 1, 13, 5], dtype=np.int64)
    eps = rng.standard_normal(B).astype(np.float32)
    genes = np.stack([rng.permutation(2000)[:G] for _ in range(B)]).astype(np.int64)
    scales = {"clusters": 1.0}
    smp = SizeFactorSampler(enc, "mutually_exclusive", "cuda")
    sf = smp.sample({"clusters": cu(lab)}, B, eps=cu(eps))
    sf_ref = sample_log_size_factors(enc, "mutually_exclusive", {"clusters": lab}, B, eps)
    assert np.array_equal(sf.cpu().numpy(), sf_ref)
    nb, z = sample_cells(dit, vae, {"clusters": cu(lab)}, scales, B, cu(genes), None, num_steps=6, sampling_method="euler", z0=cu(z0),
                         draw_counts=False, size_factor_sampler=SimpleNamespace(sample=lambda c, b: sf))
    z2 = torch.from_numpy(np.concatenate([z0, z0]))
    cond2 = {"clusters": torch.from_numpy(np.concatenate([lab, lab]))}
    z_ref = sample_ode_fixed(z2, lambda x, t: dit_forward_with_cfg(sd_d, cfg_d, x, t, cond2, scales), 6, "euler")
    lib = torch.exp(torch.from_numpy(np.concatenate([sf_ref, sf_ref]))).view(-1, 1)
    mu_ref, th_ref = decode(sd_v, cfg_v, z_ref, torch.from_numpy(np.concatenate([genes, genes])), lib)
    assert z.shape == (2 * B, 16, 16) and max_abs_rel(z.cpu(), z_ref) < TOL
    assert max_abs_rel(nb.mu.cpu(), mu_ref) < 2e-4 and max_abs_rel(nb.theta.cpu(), th_ref) < 1e-5
    assert torch.allclose(nb.mu.sum(1).cpu(), lib.view(-1), rtol=1e-4)          # rows sum to the drawn library size (exp(0) = 1 where no statistics)
    counts, _ = sample_cells(dit, vae, {"clusters": cu(lab)}, scales, B, cu(genes), sf, num_steps=6, sampling_method="euler", z0=cu(z0))
    indptr, indices, data = dense_to_csr(counts)
    ref = sp.csr_matrix(counts.cpu().numpy())
    assert np.array_equal(indptr.cpu().numpy(), ref.indptr) and np.array_equal(indices.cpu().numpy(), ref.indices)
    assert np.array_equal(data.cpu().numpy(), ref.data) and counts.shape == (2 * B, G) and (counts >= 0).all()


def test_generate_cells_stream_equals_the_serial_chain_batch_by_batch():
    """The two-stream prediction loop (sampling.generate_cells_stream: batch i's decode / draw / CSR / host copies beside batch i + 1's
    ODE) returns, batch by batch and in order, exactly the host arrays of the serial chain sample_cells -> dense_to_csr -> to_host with
    the same noise, size factors and draw seeds; the consumer's grad mode is untouched between yields."""
    from scldm_amd.datamodule import dense_to_csr, to_host
    from scldm_amd.nnets import DiT
    from scldm_amd.sampling import generate_cells_stream, sample_cells
    g, vae, sd_v, cfg_v = build("vae_2000")
    vae.precision = "fp16"
    gd = load_golden("dit_base")
    kw = golden_json(gd, "kwargs_json")
    shapes = {k: tuple(v) for k, v in golden_json(gd, "shapes_json").items()}
    dit = DiT(**kw)
    dit.load_state_dict(make_state_dict(shapes, int(gd["seed"])), strict=True)
    dit = dit.cuda().eval()
    dit.precision = "bf16"
    rng = np.random.default_rng(23)
    B, G, K = 24, 300, 4
    genes = cu(np.stack([rng.permutation(2000)[:G] for _ in range(B)]).astype(np.int64))
    scales = {"clusters": 1.5}
    items = [({"clusters": cu(rng.integers(0, 13, B))}, cu((6.0 + rng.standard_normal(B) * 0.3).astype(np.float32)),
              cu(rng.standard_normal((B, 16, 16)).astype(np.float32))) for _ in range(K)]
    seeds = [101, 202, 303, 404]
    serial = []
    for (cond, sf, z0), seed in zip(items, seeds):
        counts, z = sample_cells(dit, vae, cond, scales, B, genes, sf, num_steps=5, sampling_method="heun", z0=z0, seed=seed)
        serial.append(to_host(*dense_to_csr(counts), z))
    got = []
    for out in generate_cells_stream(dit, vae, items, scales, genes, num_steps=5, sampling_method="heun", seeds=seeds):
        assert torch.is_grad_enabled()
        got.append(out)
    assert len(got) == K
    for a, b in zip(serial, got):
        assert all(not t.is_cuda for t in b) and len(b) == 4
        for x, y in zip(a, b):
            assert x.dtype == y.dtype and torch.equal(x, y)
    assert not torch.equal(got[0][2], got[1][2])
    # k consecutive batches sharing ONE solve (a cell's trajectory does not depend on its batch): the same arrays batch by batch, also when
    # the last group is short
    for k in (2, 3, 8):
        merged = list(generate_cells_stream(dit, vae, items, scales, genes, num_steps=5, sampling_method="heun", seeds=seeds, merge_batches=k))
        assert len(merged) == K
        for a, b in zip(serial, merged):
            for x, y in zip(a, b):
                assert x.dtype == y.dtype and torch.equal(x, y), k
    # size factors drawn by the sampler when a batch carries none; a generator (lazy) source works
    from types import SimpleNamespace
    smp = SimpleNamespace(sample=lambda c, b: torch.full((b,), 6.0, device="cuda"))
    outs = list(generate_cells_stream(dit, vae, (it[0] for it in items[:2]), scales, genes, num_steps=3, size_factor_sampler=smp))
    assert len(outs) == 2 and outs[0][0].shape == (2 * B + 1,) and int(outs[0][0][-1]) == outs[0][2].numel()
    with pytest.raises(ValueError, match="size_factor_sampler"):
        next(generate_cells_stream(dit, vae, [items[0][0]], scales, genes, num_steps=3))
    # inside the loop the CFG plan is sync-free (dense label-tuple rows): out-of-range labels are clamped on the device and raised after the loop
    assert dit.deferred_label_check is False
    bad = ({"clusters": torch.full((B,), 99, dtype=torch.long, device="cuda")}, items[0][1], items[0][2])
    with pytest.raises(IndexError):
        list(generate_cells_stream(dit, vae, [items[0], bad], scales, genes, num_steps=3, seeds=[1, 2]))
    assert dit.deferred_label_check is False and dit.check_labels() == 0


def test_bf16_decode_close_to_fp32():
    """decode with bf16 operands in the per-gene contractions (vae.precision = "bf16"): mu within bf16 noise of the fp32 path,
    rows still sum to the library size, theta identical."""
    g, vae, sd, cfg = build("vae_2000")
    z = cu(g["z"])
    genes, lib = cu(g["genes"]), cu(g["library_size"])
    ref = vae.decode(z, genes, lib)
    vae.precision = "bf16"
    nb = vae.decode(z, genes, lib)
    assert torch.equal(nb.theta, ref.theta)
    assert max_abs_rel(nb.mu.cpu(), ref.mu.cpu()) < 3e-2
    assert float(((nb.mu - ref.mu).norm() / ref.mu.norm())) < 1e-2
    assert torch.allclose(nb.mu.sum(1, keepdim=True), lib.view(-1, 1), rtol=1e-4)
    assert max_abs_rel(nb.mu.cpu(), g["mu"]) < 3e-2


def test_bf16_encode_close_to_fp32():
    g, vae, sd, cfg = build("vae_2000")
    counts, genes = cu(g["counts_subset"]), cu(g["genes_subset"])
    ref = vae.encode(cu(g["counts"]), cu(g["genes"]), counts, genes)
    vae.precision = "bf16"
    z = vae.encode(cu(g["counts"]), cu(g["genes"]), counts, genes)
    assert max_abs_rel(z.cpu(), ref.cpu()) < 3e-2 and max_abs_rel(z.cpu(), g["z"]) < 3e-2


def _fresh_vae(G, seed):
    from test_abi_cpu import _build_vae as bv
    vae = bv(G)
    sd = make_state_dict({k: tuple(v.shape) for k, v in vae.state_dict().items()}, seed)
    vae.load_state_dict(sd, strict=True)
    return vae.cuda().eval(), sd, VAEConfig(n_genes=G)


def test_hlca_size_encode_decode_vs_oracle():
    """hlca shape (G = 27 997 genes out, S = 10 186 tokens in; experiments/configs/datamodule/default.yaml:60-64), the largest
    MCAB configuration of the reference: 3 cells (an odd batch: the last trunk tile is half padding) against the oracle,
    mu rows sum to the library size, bit-repeatable."""
    G, S, B = 27997, 10186, 3
    vae, sd, cfg = _fresh_vae(G, 91)
    rng = np.random.default_rng(17)
    genes = np.tile(np.arange(G, dtype=np.int64), (B, 1))
    counts = rng.poisson(0.5, (B, G)).astype(np.float32)
    sub = np.stack([np.sort(rng.permutation(G)[:S]) for _ in range(B)])
    lib = counts.sum(1, keepdims=True) + 1
    cs, gs = np.take_along_axis(counts, sub, 1), np.take_along_axis(genes, sub, 1)
    z = vae.encode(cu(cs), cu(gs))
    z_ref = encode(sd, cfg, torch.from_numpy(cs), torch.from_numpy(gs))
    check_err(z.cpu(), z_ref, TOL, "hlca encode (S=10186) vs oracle")
    nb = vae.decode(z, cu(genes), cu(lib))
    mu_ref, th_ref = decode(sd, cfg, z.cpu(), torch.from_numpy(genes), torch.from_numpy(lib))
    check_err(nb.mu.cpu(), mu_ref, TOL, "hlca decode mu (G=27997) vs oracle")
    assert max_abs_rel(nb.theta.cpu(), th_ref) < 1e-5
    assert torch.allclose(nb.mu.sum(1), cu(lib[:, 0]), rtol=2e-4)
    assert torch.equal(vae.decode(z, cu(genes), cu(lib)).mu, nb.mu) and torch.equal(vae.encode(cu(cs), cu(gs)), z)


@pytest.mark.parametrize("B", [1, 2, 7, 9, 64])
def test_trunk_tiles_any_batch(B):
    """The per-cell trunks run two cells per 32-token MFMA tile, eight per workgroup: batches around those boundaries."""
    g, vae, sd, cfg = build("vae_2000")
    rng = np.random.default_rng(B)
    S, G = 40, 70
    genes_s = rng.integers(0, 2000, (B, S)).astype(np.int64)
    counts_s = rng.poisson(1.5, (B, S)).astype(np.float32)
    genes = np.stack([rng.permutation(2001)[:G] for _ in range(B)]).astype(np.int64)
    lib = rng.uniform(100, 2000, (B, 1)).astype(np.float32)
    zr = rng.standard_normal((B, 16, 16)).astype(np.float32)
    check_err(vae.encode(cu(counts_s), cu(genes_s)).cpu(), encode(sd, cfg, torch.from_numpy(counts_s), torch.from_numpy(genes_s)), TOL, f"encode B={B}")
    mu_ref, _ = decode(sd, cfg, torch.from_numpy(zr), torch.from_numpy(genes), torch.from_numpy(lib))
    check_err(vae.decode(cu(zr), cu(genes), cu(lib)).mu.cpu(), mu_ref, TOL, f"decode B={B}")


def test_vae_deepcopy_and_inplace_update():
    import copy
    g, vae, sd, cfg = build("vae_2000")
    z, genes, lib = cu(g["z"]), cu(g["genes"]), cu(g["library_size"])
    mu0 = vae.decode(z, genes, lib).mu
    v2 = copy.deepcopy(vae)
    assert v2._handle is None and torch.equal(v2.decode(z, genes, lib).mu, mu0)
    versions = [p._version for p in vae.parameters()]
    for p in vae.decoder.parameters():
        p.data.mul_(1.02)                      # EMA-style update: no version bump
    assert [p._version for p in vae.parameters()] == versions
    mu1 = vae.decode(z, genes, lib).mu
    assert not torch.equal(mu1, mu0)
    sd2 = {k: v.detach().cpu() for k, v in vae.state_dict().items()}
    mu_ref, _ = decode(sd2, cfg, z.cpu(), genes.cpu(), lib.cpu())
    check_err(mu1.cpu(), mu_ref, TOL, "decode after an in-place .data update")


# --------------------------------------------------------------------------------------------------------------------
# negative-binomial draw on device (scldm_nb_sample / scldm_vae_decode_sample): RNG-dependent, so tested statistically
# --------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mu,theta", [(5.0, 2.0), (0.05, 10.0), (200.0, 0.5), (30.0, 50.0), (0.7, 0.3)])
def test_nb_draw_matches_the_analytic_distribution(mu, theta):
    """counts ~ Poisson(Gamma(theta, rate theta/mu)) is NB(n = theta, p = theta / (theta + mu)): chi-square of 400 000 draws against
    the analytic pmf (covers Gamma shape < 1 and >= 1, Poisson by inversion and by PTRS), plus mean / variance."""
    from scipy import stats
    from scldm_amd.stochastic_layers import NegativeBinomial
    n = 400_000
    nb = NegativeBinomial(mu=torch.full((n,), mu, device="cuda"), theta=torch.full((n,), theta, device="cuda"))
    x = nb.sample(seed=1234).cpu().numpy()
    assert (x >= 0).all() and (x == np.round(x)).all()
    var = mu + mu * mu / theta
    assert abs(x.mean() - mu) < 6 * np.sqrt(var / n)
    assert abs(x.var() - var) < 0.05 * var + 6 * var * np.sqrt(2.0 / n) * 3
    dist = stats.nbinom(theta, theta / (theta + mu))
    hi = int(dist.ppf(1 - 1e-4)) + 1
    edges = np.unique(np.round(dist.ppf(np.linspace(0, 1 - 1e-4, 30))).astype(int))
    edges = np.concatenate([[0], edges[edges > 0], [hi + 1]])
    obs, exp = [], []
    for lo_, hi_ in zip(edges[:-1], edges[1:]):
        p = dist.cdf(hi_ - 1) - (dist.cdf(lo_ - 1) if lo_ > 0 else 0.0)
        if p * n < 20:
            continue
        obs.append(((x >= lo_) & (x < hi_)).sum())
        exp.append(p * n)
    obs, exp = np.array(obs, float), np.array(exp, float)
    chi2 = ((obs - exp) ** 2 / exp).sum()
    assert chi2 < stats.chi2(len(obs)).ppf(1 - 1e-6), (chi2, len(obs))


def test_nb_draw_is_reproducible_and_seeded():
    from scldm_amd.stochastic_layers import NegativeBinomial
    gen = torch.Generator(device="cuda").manual_seed(0)
    mu = torch.rand(1000, 37, device="cuda", generator=gen) * 20
    theta = torch.rand(1000, 37, device="cuda", generator=gen) * 5 + 0.1
    nb = NegativeBinomial(mu=mu, theta=theta)
    a, b, c = nb.sample(seed=7), nb.sample(seed=7), nb.sample(seed=8)
    assert torch.equal(a, b) and not torch.equal(a, c)
    torch.manual_seed(3); d = nb.sample()
    torch.manual_seed(3); e = nb.sample()
    assert torch.equal(d, e)                        # default seed comes from torch's global generator
    assert float((nb.sample(seed=9)[mu == 0]).sum()) == 0.0
    # the first 1000 elements do not depend on how many more are drawn (counter = element index)
    flat = NegativeBinomial(mu=mu.reshape(-1)[:1000], theta=theta.reshape(-1)[:1000]).sample(seed=7)
    assert torch.equal(flat, a.reshape(-1)[:1000])


def test_decode_sample_fuses_decode_and_draw():
    """decode_sample(z, genes, lib, seed) == NegativeBinomial(decode(z, genes, lib)).sample(seed): same parameters (recomputed in
    registers), same counter-based stream - equal except where the two passes' mu differ by an ulp and flip an accept/reject."""
    g, vae, sd, cfg = build("vae_2000")
    z, genes, lib = cu(g["z"]), cu(g["genes"]), cu(g["library_size"])
    counts = vae.decode_sample(z, genes, lib, seed=42)
    nb = vae.decode(z, genes, lib)
    ref = nb.sample(seed=42)
    assert counts.shape == nb.mu.shape and (counts >= 0).all() and torch.equal(counts, counts.round())
    assert float((counts != ref).float().mean()) < 1e-3
    assert torch.equal(vae.decode_sample(z, genes, lib, seed=42), counts)
    # many draws average to mu
    acc = torch.zeros_like(counts)
    for s in range(64):
        acc += vae.decode_sample(z, genes, lib, seed=100 + s)
    rel = (acc.sum(1) / 64 - lib.view(-1)).abs() / lib.view(-1)
    assert float(rel.max()) < 0.05


# ---------------------------------------------------------------------------------------------------------------------------------
# precision = "fp16": the reference's own arithmetic class for MCAB (TF32 operands under set_float32_matmul_precision("high"):
# experiments/scripts/inference.py:26, train.py:18; src/scldm/layers.py:248-264,305-330).  Gate = the DiT's: the error against the
# exact-fp32 reference output must not exceed 1.5 x the error of the ORACLE run with every matmul operand rounded to 10 mantissa
# bits, on the same inputs.
# ---------------------------------------------------------------------------------------------------------------------------------
def _tf32_gate(err_fp16, err_tf32, what):
    print(f"[parity] {what}: fp16-operand kernels {err_fp16:.3e}   TF32-operand oracle {err_tf32:.3e}   ratio {err_fp16 / err_tf32:.2f} (gate 1.5)")
    assert err_fp16 <= 1.5 * err_tf32, (what, err_fp16, err_tf32)


@pytest.mark.parametrize("name", ["vae_small", "vae_2000"])
def test_fp16_policy_within_the_tf32_operand_oracle_on_reference_goldens(name):
    from oracle.dit import matmul_operand_bits
    g, vae, sd, cfg = build(name)
    t = torch.from_numpy
    with matmul_operand_bits(10):
        z_tf = encode(sd, cfg, t(g["counts_subset"]), t(g["genes_subset"]))
        mu_tf, _ = decode(sd, cfg, t(g["z"]), t(g["genes"]), t(g["library_size"]))
        mu_tf_r, _ = decode(sd, cfg, t(g["zrand"]), t(g["genes"]), t(g["library_size"]))
    vae.precision = "fp16"
    z = vae.encode(cu(g["counts"]), cu(g["genes"]), cu(g["counts_subset"]), cu(g["genes_subset"]))
    _tf32_gate(max_abs_rel(z.cpu(), g["z"]), max_abs_rel(z_tf, g["z"]), f"{name} encode")
    nb = vae.decode(cu(g["z"]), cu(g["genes"]), cu(g["library_size"]))
    _tf32_gate(max_abs_rel(nb.mu.cpu(), g["mu"]), max_abs_rel(mu_tf, g["mu"]), f"{name} decode mu")
    assert max_abs_rel(nb.theta.cpu(), g["theta"]) < 1e-5      # theta = exp(table[gene]): no contraction, exact in every policy
    nb_r = vae.decode(cu(g["zrand"]), cu(g["genes"]), cu(g["library_size"]))
    _tf32_gate(max_abs_rel(nb_r.mu.cpu(), g["mu_rand"]), max_abs_rel(mu_tf_r, g["mu_rand"]), f"{name} decode mu (random latents)")
    assert torch.allclose(nb.mu.sum(1).cpu(), t(g["library_size"][:, 0]), rtol=1e-4)
    # the fused draw takes the same policy, is reproducible, and has the decode's mean
    c1 = vae.decode_sample(cu(g["z"]), cu(g["genes"]), cu(g["library_size"]), seed=5)
    assert torch.equal(c1, vae.decode_sample(cu(g["z"]), cu(g["genes"]), cu(g["library_size"]), seed=5)) and (c1 >= 0).all()
    # fp16 is strictly closer to the reference than bf16 on the same inputs
    vae.precision = "bf16"
    e_bf = max_abs_rel(vae.decode(cu(g["z"]), cu(g["genes"]), cu(g["library_size"])).mu.cpu(), g["mu"])
    assert max_abs_rel(nb.mu.cpu(), g["mu"]) < e_bf


def test_fp16_policy_hlca_size_within_the_tf32_operand_oracle():
    """The largest MCAB configuration of the reference (G = 27 997, S = 10 186), 3 cells, fp16 operands against the exact oracle,
    gated by the TF32-operand oracle; rows sum to the library size; bit-repeatable."""
    from oracle.dit import matmul_operand_bits
    G, S, B = 27997, 10186, 3
    vae, sd, cfg = _fresh_vae(G, 91)
    rng = np.random.default_rng(17)
    genes = np.tile(np.arange(G, dtype=np.int64), (B, 1))
    counts = rng.poisson(0.5, (B, G)).astype(np.float32)
    sub = np.stack([np.sort(rng.permutation(G)[:S]) for _ in range(B)])
    lib = counts.sum(1, keepdims=True) + 1
    cs, gs = np.take_along_axis(counts, sub, 1), np.take_along_axis(genes, sub, 1)
    t = torch.from_numpy
    z_ref = encode(sd, cfg, t(cs), t(gs))
    mu_ref, _ = decode(sd, cfg, z_ref, t(genes), t(lib))
    with matmul_operand_bits(10):
        z_tf = encode(sd, cfg, t(cs), t(gs))
        mu_tf, _ = decode(sd, cfg, z_ref, t(genes), t(lib))
    vae.precision = "fp16"
    z = vae.encode(cu(cs), cu(gs))
    _tf32_gate(max_abs_rel(z.cpu(), z_ref), max_abs_rel(z_tf, z_ref), "hlca encode (S=10186)")
    nb = vae.decode(z_ref.cuda(), cu(genes), cu(lib))
    _tf32_gate(max_abs_rel(nb.mu.cpu(), mu_ref), max_abs_rel(mu_tf, mu_ref), "hlca decode mu (G=27997)")
    assert torch.allclose(nb.mu.sum(1), cu(lib[:, 0]), rtol=2e-4)
    assert torch.equal(vae.decode(z_ref.cuda(), cu(genes), cu(lib)).mu, nb.mu) and torch.equal(vae.encode(cu(cs), cu(gs)), z)


def test_unknown_vae_precision_raises():
    g, vae, sd, cfg = build("vae_small")
    vae.precision = "bf16x3"       # a DiT-only policy
    with pytest.raises(RuntimeError, match="precision"):
        vae.decode(cu(g["z"]), cu(g["genes"]), cu(g["library_size"]))


def test_unshared_theta_head_draw_and_training_guard():
    """decoder_name negative_binomial_unshared_theta (src/scldm/stochastic_layers.py:94-96,109-112; golden `vae_unshared` from the
    reference): theta (B, G) = exp of the head's second output in every precision policy; the fused draw uses those per-element
    dispersions (same counts as NegativeBinomial(mu, theta).sample with the same seed: both are nb_draw(seed, element, mu, theta));
    training raises (the HIP backward is the shared-theta head's)."""
    g, vae, sd, cfg = build("vae_unshared")
    z, genes, lib = cu(g["z"]), cu(g["genes"]), cu(g["library_size"])
    nb = vae.decode(z, genes, lib)
    assert max_abs_rel(nb.mu.cpu(), g["mu"]) < TOL and max_abs_rel(nb.theta.cpu(), g["theta"]) < TOL
    assert float(nb.theta.std()) > 0 and not torch.equal(nb.theta[0], nb.theta[1])      # per cell AND gene
    for prec, tol in (("fp16", 2e-3), ("bf16", 3e-2)):
        vae.precision = prec
        nbp = vae.decode(z, genes, lib)
        assert max_abs_rel(nbp.theta.cpu(), g["theta"]) < tol and max_abs_rel(nbp.mu.cpu(), g["mu"]) < tol
    vae.precision = "fp32"
    c1 = vae.decode_sample(z, genes, lib, seed=11)
    assert torch.equal(c1, vae.decode_sample(z, genes, lib, seed=11)) and (c1 >= 0).all()
    assert torch.equal(c1, nb.sample(seed=11))
    vae.train()
    with pytest.raises(NotImplementedError, match="shared-theta"):
        vae(cu(g["counts"]), genes, lib, cu(g["counts_subset"]), cu(g["genes_subset"]))


@pytest.mark.parametrize("B,S,G", [(1, 1, 1), (3, 31, 33), (2, 64, 1025), (5, 100, 4099)])
def test_fp16_policy_ragged_sizes_vs_oracle(B, S, G):
    """The 16-bit operand policy at sizes that do not fill 32-gene tiles / 1 024-gene chunks (S = 1, G = 1 included): finite, inside the
    TF32-operand oracle's class (<= 2e-3 of the exact oracle: the oracle with 10-bit operands reads 1.5-6e-4 on the goldens), rows
    sum to the library size."""
    g, vae, sd, cfg = build("vae_2000")
    vae.precision = "fp16"
    rng = np.random.default_rng(B * 1000 + S + G)
    genes_s = rng.integers(0, 2000, (B, S)).astype(np.int64)
    counts_s = rng.poisson(1.5, (B, S)).astype(np.float32)
    genes = np.stack([rng.permutation(2001)[:G] if G <= 2001 else rng.integers(0, 2001, G) for _ in range(B)]).astype(np.int64)
    lib = rng.uniform(100, 2000, (B, 1)).astype(np.float32)
    zr = rng.standard_normal((B, 16, 16)).astype(np.float32)
    z = vae.encode(cu(counts_s), cu(genes_s))
    assert torch.isfinite(z).all() and max_abs_rel(z.cpu(), encode(sd, cfg, torch.from_numpy(counts_s), torch.from_numpy(genes_s))) < 2e-3
    mu_ref, th_ref = decode(sd, cfg, torch.from_numpy(zr), torch.from_numpy(genes), torch.from_numpy(lib))
    nb = vae.decode(cu(zr), cu(genes), cu(lib))
    assert max_abs_rel(nb.mu.cpu(), mu_ref) < 2e-3 and max_abs_rel(nb.theta.cpu(), th_ref) < 1e-5
    assert torch.allclose(nb.mu.sum(1, keepdim=True).cpu(), torch.from_numpy(lib), rtol=1e-4)


def test_to_host_matches_cpu_copies():
    from scldm_amd.datamodule import dense_to_csr, to_host
    g = torch.Generator(device="cuda").manual_seed(0)
    dense = torch.poisson(torch.full((33, 517), 0.2, device="cuda"), generator=g)
    indptr, indices, data = dense_to_csr(dense)
    h = to_host(indptr, indices, data, dense)
    assert all(not t.is_cuda and t.is_pinned() for t in h)
    for a, b in zip(h, (indptr, indices, data, dense)):
        assert torch.equal(a, b.cpu()) and a.dtype == b.dtype
