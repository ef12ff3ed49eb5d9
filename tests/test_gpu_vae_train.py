"""GPU parity of the TransformerVAE TRAINING step (BASELINE configs[0]): scldm_vae_train_forward / _backward
through the C ABI and the autograd binding (`TransformerVAE.forward` with gradients enabled), against
  (a) digests of the REFERENCE's own autograd gradients of every parameter (tests/golden/vae_train_*.npz), and
  (b) autograd over the CPU oracle (oracle/vae_train.py) on seeded ragged sizes, including a gradient through the returned z.
fp32; tolerance 1e-4 of each tensor's scale (max |entry|, or l2 / sqrt(numel) for the digests)."""
import numpy as np
import pytest
import torch

from conftest import golden_json, load_golden, max_abs_rel
from oracle.train import grad_digest
from oracle.vae import VAEConfig
from oracle.vae_train import FROZEN, log_nb_positive as log_nb_oracle, vae_training_grads
from oracle.weights import make_state_dict
from test_abi_cpu import _build_vae

pytestmark = pytest.mark.gpu
TOL = 1e-4
BIAS = "decoder_head.params.bias"   # mathematically zero gradient (softmax over genes is shift-invariant): noise on both sides


def build(n_genes, seed):
    vae = _build_vae(n_genes)
    shapes = {k: tuple(v.shape) for k, v in vae.state_dict().items()}
    sd = make_state_dict(shapes, seed)
    vae.load_state_dict(sd, strict=True)
    return vae.cuda().train(), sd, VAEConfig(n_genes=n_genes)


def cu(a):
    return torch.from_numpy(np.asarray(a)).cuda()


def hip_step(vae, counts, genes, lib, counts_s, genes_s, z_weight=None, fused_loss=False):
    for p in vae.parameters():
        p.grad = None
    params, z = vae(cu(counts), cu(genes), cu(lib), cu(counts_s), cu(genes_s))
    if fused_loss:
        from scldm_amd.distributions import log_nb_positive
        recon = -log_nb_positive(cu(counts), params["mu"], params["theta"])
    else:   # the reference's own eager formula on top of the differentiable outputs (models.py:243)
        recon = -log_nb_oracle(cu(counts), params["mu"], params["theta"])
    loss = recon.sum(dim=1).mean()
    if z_weight is not None:
        loss = loss + (z * cu(z_weight)).sum()
    loss.backward()
    return loss.detach(), params, z.detach()


@pytest.mark.parametrize("name", ["vae_train_small", "vae_train_2000"])
@pytest.mark.parametrize("fused_loss", [False, True])
def test_gradients_match_reference_digests(name, fused_loss):
    g = load_golden(name)
    vae, sd, cfg = build(int(g["n_genes"]), int(g["seed"]))
    loss, params, z = hip_step(vae, g["counts"], g["genes"], g["library_size"], g["counts_subset"], g["genes_subset"], fused_loss=fused_loss)
    assert abs(float(loss) - float(g["loss"])) <= TOL * abs(float(g["loss"]))
    assert max_abs_rel(params["mu"].detach().cpu(), g["mu"]) < TOL and max_abs_rel(z.cpu(), g["z"]) < TOL
    assert golden_json(g, "frozen_json") == list(FROZEN)
    wn = g["grad_decoder_head.params.weight"][1]
    bad = {}
    for name_, p in vae.named_parameters():
        if name_ in FROZEN:
            assert p.grad is None
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), name_
        ref, ours = g[f"grad_{name_}"], grad_digest(p.grad)
        if name_ == BIAS:
            if not abs(ours[0]) <= 1e-3 * wn:
                bad[name_] = ours[0]
            continue
        scale = max(np.abs(ref[2:]).max(), ref[1] / np.sqrt(p.numel()))
        e = max(np.abs(ours[2:] - ref[2:]).max() / scale, abs(ours[1] - ref[1]) / (ref[1] + 1e-30))
        if not e <= TOL:
            bad[name_] = e
    print(f"[parity] VAE training {name} (fused loss {fused_loss}): {len(bad)} of 175 gradients outside {TOL:g}")
    assert not bad, bad


@pytest.mark.parametrize("B,S,G,n_genes", [(5, 70, 130, 200), (3, 300, 1000, 2000), (9, 64, 65, 300), (1, 1, 1, 50)])
def test_gradients_match_oracle_on_ragged_sizes_with_a_gradient_through_z(B, S, G, n_genes):
    """Batches that do not fill a four-cell wave, gene axes that do not fill 64-token tiles, repeated genes in a cell (scatter-add
    into the same embedding row), padding tokens with zero counts, and a loss term on the returned latent (d loss / d z path)."""
    vae, sd, cfg = build(n_genes, 300 + B)
    rng = np.random.default_rng(B * 100 + G)
    genes = rng.integers(0, n_genes + 1, (B, G)).astype(np.int64)
    counts = rng.poisson(0.9, (B, G)).astype(np.float32)
    genes_s = rng.integers(0, n_genes + 1, (B, S)).astype(np.int64)
    counts_s = rng.poisson(0.9, (B, S)).astype(np.float32)
    counts_s[:, -max(1, S // 5):] = 0.0                 # padding-style tokens (mask token rows with zero counts)
    lib = (counts.sum(1, keepdims=True) + 1.0).astype(np.float32)
    zw = (0.3 * rng.standard_normal((B, 16, 16))).astype(np.float32)
    loss, params, z = hip_step(vae, counts, genes, lib, counts_s, genes_s, z_weight=zw)
    t = torch.from_numpy
    loss_o, (mu_o, th_o, z_o), grads = vae_training_grads(sd, cfg, t(counts), t(genes), t(lib), t(counts_s), t(genes_s), z_weight=t(zw))
    assert abs(float(loss) - float(loss_o)) <= TOL * abs(float(loss_o))
    assert max_abs_rel(z.cpu(), z_o) < TOL and max_abs_rel(params["mu"].detach().cpu(), mu_o) < TOL
    wn = float(grads["decoder_head.params.weight"].norm())
    bad = {}
    for name_, p in vae.named_parameters():
        if name_ in FROZEN:
            continue
        ref = grads[name_]
        if name_ == BIAS:
            if not abs(float(p.grad)) <= 1e-3 * wn:
                bad[name_] = float(p.grad)
            continue
        e = max_abs_rel(p.grad.cpu(), ref) if float(ref.abs().max()) > 0 else float(p.grad.abs().max())
        if not e < TOL:
            bad[name_] = e
    print(f"[parity] VAE training B={B} S={S} G={G}: worst gradient error "
          f"{max([0.0] + [max_abs_rel(p.grad.cpu(), grads[k]) for k, p in vae.named_parameters() if k not in FROZEN and k != BIAS and float(grads[k].abs().max()) > 0]):.2e}")
    assert not bad, bad


def test_fused_log_nb_positive_matches_the_reference_formula_and_its_autograd():
    from scldm_amd.distributions import log_nb_positive
    gen = torch.Generator().manual_seed(3)
    x = torch.poisson(torch.full((7, 1300), 1.2), generator=gen)
    x[0, :5] = torch.tensor([0.0, 1.0, 50.0, 400.0, 3000.0])
    mu = torch.rand(7, 1300, generator=gen) * 5 + 1e-3
    mu[1, :4] = torch.tensor([1e-6, 1e-3, 80.0, 2500.0])
    theta = torch.exp(torch.randn(7, 1300, generator=gen))
    theta[2, :4] = torch.tensor([1e-3, 0.05, 30.0, 900.0])
    w = torch.randn(7, 1300, generator=gen)
    mo, to = mu.double().requires_grad_(True), theta.double().requires_grad_(True)
    ref = log_nb_oracle(x.double(), mo, to)
    (ref * w.double()).sum().backward()
    mg, tg = mu.cuda().requires_grad_(True), theta.cuda().requires_grad_(True)
    out = log_nb_positive(x.cuda(), mg, tg)
    (out * w.cuda()).sum().backward()
    # fp32 arithmetic on differences of lgamma / log values of size ~1e4 (x = 3000): the yardstick is the error of the reference's
    # own fp32 eager formula against float64 on the same inputs - the fused kernel must not be worse than twice that
    rel = lambda a, b: float(((a.double().cpu() - b).abs() / (b.abs() + 1e-3 * b.abs().max())).max())
    m32, t32 = mu.clone().requires_grad_(True), theta.clone().requires_grad_(True)
    ref32 = log_nb_oracle(x, m32, t32)
    (ref32 * w).sum().backward()
    e_out, e_dmu, e_dth = rel(out.detach(), ref.detach()), rel(mg.grad, mo.grad), rel(tg.grad, to.grad)
    r_out, r_dmu, r_dth = rel(ref32.detach(), ref.detach()), rel(m32.grad, mo.grad), rel(t32.grad, to.grad)
    print(f"[parity] fused log_nb_positive vs float64: value {e_out:.2e} (torch fp32 eager {r_out:.2e}), d mu {e_dmu:.2e} ({r_dmu:.2e}), "
          f"d theta {e_dth:.2e} ({r_dth:.2e})")
    assert e_out <= 2 * r_out + 1e-5 and e_dmu <= 2 * r_dmu + 1e-5 and e_dth <= 2 * r_dth + 1e-4
    assert e_out < 5e-4 and e_dmu < 1e-4 and e_dth < 1e-3


def test_training_loop_at_the_dentate_shape_reduces_the_loss():
    """BASELINE configs[0] shape: batch 32, G = 17 002 decoded genes, S = 6 147 encoder tokens; 20 AdamW steps on a fixed batch."""
    G, S, B, n_genes = 17002, 6147, 32, 17002
    vae, sd, cfg = build(n_genes, 401)
    with torch.no_grad():   # a realistic starting point: embeddings O(1), theta table at its reference init (ones)
        vae.input_layer.gene_embedding.weight.normal_(0, 1.0)
        vae.encoder.ca_layer.inducing_points.normal_(0, 1.0)
        vae.decoder_head.theta.weight.fill_(1.0)
    rng = np.random.default_rng(5)
    rate = rng.gamma(0.3, 2.0, (1, G)).astype(np.float32)
    counts = rng.poisson(rate * rng.uniform(0.5, 1.5, (B, 1))).astype(np.float32)
    genes = np.tile(np.arange(G, dtype=np.int64), (B, 1))
    from scldm_amd.datamodule import tokenize_cells_expressed
    tok = tokenize_cells_expressed(cu(counts), cu(genes[0]), S, n_genes)
    lib = cu(counts).sum(1, keepdim=True)
    opt = torch.optim.AdamW(vae.parameters(), lr=2e-3)
    from scldm_amd.distributions import log_nb_positive
    losses = []
    for _ in range(20):
        opt.zero_grad(set_to_none=True)
        params, z = vae(cu(counts), cu(genes), lib, tok["counts_subset"], tok["genes_subset"])
        loss = (-log_nb_positive(cu(counts), params["mu"], params["theta"])).sum(dim=1).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    print(f"[parity] VAE training loop (B=32, G=17002, S=6147): loss {losses[0]:.1f} -> {losses[-1]:.1f}")
    assert all(np.isfinite(losses)) and losses[-1] < 0.97 * losses[0] and min(losses[10:]) < min(losses[:5])


def test_eval_and_no_grad_forward_stay_on_the_inference_path():
    g = load_golden("vae_train_small")
    vae, sd, cfg = build(int(g["n_genes"]), int(g["seed"]))
    args = (cu(g["counts"]), cu(g["genes"]), cu(g["library_size"]), cu(g["counts_subset"]), cu(g["genes_subset"]))
    with torch.no_grad():
        p0, z0 = vae(*args)
    assert not p0["mu"].requires_grad and not z0.requires_grad
    p1, z1 = vae(*args)
    assert p1["mu"].requires_grad and z1.requires_grad and torch.equal(p1["mu"], p0["mu"]) and torch.equal(z1, z0)
    for p in vae.parameters():
        p.requires_grad_(False)
    p2, _ = vae(*args)                          # frozen VAE (the LDM training case, models.py:432-435): plain inference
    assert not p2["mu"].requires_grad


@pytest.mark.parametrize("env", [{"SCLDM_VAE_GENE_MFMA": "1"}, {"SCLDM_VAE_GENE_MFMA": "0"}, {"SCLDM_VAE_CELL_WIDE": "0", "SCLDM_VAE_GENE_WIDE": "0"}],
                         ids=["per-gene backward with only the MLP on the matrix pipe", "per-gene backward on the VALU", "first-version kernels"])
def test_the_selectable_earlier_kernel_versions_pass_the_same_gates(env):
    """The backward kernels exist in up to four generations (vae_train.hpp: one token per lane; vae_train_wide.hpp: 16 lanes per
    token, with the per-gene contractions on the VALU, the MLP's on fp32 MFMA tiles, or - the default - all of them).  The library reads the selecting environment once per
    process, so the earlier generations run the digest and ragged-size gates in a child process - they stay honest A/B baselines."""
    import os, subprocess, sys
    cmd = [sys.executable, "-m", "pytest", __file__, "-q", "-x", "-m", "gpu", "-k", "digests or ragged", "-p", "no:cacheprovider"]
    r = subprocess.run(cmd, env={**os.environ, **env}, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "8 passed" in r.stdout, r.stdout[-500:]


@pytest.mark.parametrize("n_layer,n_lat,pos,multiple_of", [(3, 24, False, 32), (1, 7, True, 8), (10, 32, True, 4)])
def test_gradients_match_oracle_on_other_model_shapes(n_layer, n_lat, pos, multiple_of):
    """The kernels take up to 16 trunk layers, latent widths up to 32 and SwiGLU hidden widths up to 96; the reference's config is one
    point of that family (8 layers, 16 wide, hidden 88, positional encoding).  Others: a latent width above 16 (the second feature of a
    lane is inside the latent LayerNorm), an odd width, no positional encoding, hidden 96 (no zero-padded units), 1 and 10 layers."""
    from scldm_amd.layers import InputTransformerVAE
    from scldm_amd.nnets import Decoder, Encoder
    from scldm_amd.stochastic_layers import NegativeBinomialTransformerLayer
    from scldm_amd.vae import TransformerVAE
    n_genes, B, S, G = 300, 6, 90, 150
    enc = Encoder(n_layer=n_layer, n_inducing_points=16, n_embed=32, n_embed_latent=n_lat, n_head=8, n_head_cross=4, dropout=0.0, bias=False,
                  multiple_of=multiple_of, layernorm_eps=1e-8, norm_layer="layernorm", positional_encoding=pos)
    dec = Decoder(n_genes=n_genes, n_embed=32, n_embed_latent=n_lat, n_head=8, n_head_cross=4, n_layer=n_layer, n_inducing_points=16,
                  dropout=0.0, bias=False, multiple_of=multiple_of, layernorm_eps=1e-8, norm_layer="layernorm", shared_embedding=True,
                  use_adaln=False)
    head = NegativeBinomialTransformerLayer(n_genes=n_genes, shared_theta=True, n_embed=32, norm_layer="layernorm", layernorm_eps=1e-8)
    vae = TransformerVAE(encoder=enc, decoder=dec, decoder_head=head, input_layer=InputTransformerVAE(n_genes=n_genes, n_embed=32, agg_func="log1p"))
    sd = make_state_dict({k: tuple(v.shape) for k, v in vae.state_dict().items()}, 900 + n_layer)
    vae.load_state_dict(sd, strict=True)
    vae = vae.cuda().train()
    cfg = VAEConfig(n_genes=n_genes, n_embed_latent=n_lat, n_layer=n_layer, multiple_of=multiple_of, positional_encoding=pos)
    rng = np.random.default_rng(n_layer * 10 + n_lat)
    genes = rng.integers(0, n_genes + 1, (B, G)).astype(np.int64)
    counts = rng.poisson(0.9, (B, G)).astype(np.float32)
    genes_s = rng.integers(0, n_genes + 1, (B, S)).astype(np.int64)
    counts_s = rng.poisson(0.9, (B, S)).astype(np.float32)
    lib = (counts.sum(1, keepdims=True) + 1.0).astype(np.float32)
    zw = (0.3 * rng.standard_normal((B, 16, n_lat))).astype(np.float32)
    loss, params, z = hip_step(vae, counts, genes, lib, counts_s, genes_s, z_weight=zw)
    t = torch.from_numpy
    loss_o, (mu_o, th_o, z_o), grads = vae_training_grads(sd, cfg, t(counts), t(genes), t(lib), t(counts_s), t(genes_s), z_weight=t(zw))
    assert abs(float(loss) - float(loss_o)) <= TOL * abs(float(loss_o))
    assert max_abs_rel(z.cpu(), z_o) < TOL and max_abs_rel(params["mu"].detach().cpu(), mu_o) < TOL
    wn = float(grads["decoder_head.params.weight"].norm())
    bad, worst = {}, 0.0
    for name_, p in vae.named_parameters():
        if name_ in FROZEN:
            continue
        ref = grads[name_]
        if name_ == BIAS:
            if not abs(float(p.grad)) <= 1e-3 * wn:
                bad[name_] = float(p.grad)
            continue
        e = max_abs_rel(p.grad.cpu(), ref) if float(ref.abs().max()) > 0 else float(p.grad.abs().max())
        worst = max(worst, e)
        if not e < TOL:
            bad[name_] = e
    print(f"[parity] VAE training, {n_layer} layers, latent width {n_lat}, pos {pos}, hidden multiple {multiple_of}: worst gradient error {worst:.2e}")
    assert not bad, bad
