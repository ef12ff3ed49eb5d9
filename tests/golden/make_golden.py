#!/usr/bin/env python3
"""Generate golden input/output vectors by IMPORTING THE REFERENCE (build container only).

Run from the repo root:   python tests/golden/make_golden.py
Needs /root/reference (read-only mount); nothing else in the repo reads it.  The
reference modules are imported under a stub parent package:
`scldm.layers`, `scldm.nnets`, `scldm.stochastic_layers`, `scldm.vae`,
`scldm.transport` depend only on torch/numpy once `scvi.distributions.NegativeBinomial`
and `torchdiffeq.odeint` exist as import-time stubs.

Only DATA is written (inputs, expected outputs, state_dict key/shape lists) to
tests/golden/*.npz.  Weights are NOT stored: both sides rebuild them from
`oracle.weights.make_state_dict(shapes, seed)`.
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("SCLDM_REFERENCE", "/root/reference/src/scldm")


def _install_stubs():
    pkg = types.ModuleType("scldm")
    pkg.__path__ = [REF]
    sys.modules["scldm"] = pkg
    scvi = types.ModuleType("scvi")
    scvid = types.ModuleType("scvi.distributions")

    class NegativeBinomial:  # plain holder; the reference only constructs it (vae.py:87)
        def __init__(self, mu, theta):
            self.mu, self.theta = mu, theta

    scvid.NegativeBinomial = NegativeBinomial
    scvi.distributions = scvid
    sys.modules["scvi"] = scvi
    sys.modules["scvi.distributions"] = scvid
    td = types.ModuleType("torchdiffeq")
    td.odeint = None  # never called: fixed-step parity is unpinned (see oracle/transport.py)
    sys.modules["torchdiffeq"] = td


_install_stubs()
from scldm.layers import InputTransformerVAE  # noqa: E402
from scldm.nnets import Decoder, DiT, Encoder  # noqa: E402
from scldm.stochastic_layers import NegativeBinomialTransformerLayer  # noqa: E402
from scldm.transport import create_transport  # noqa: E402
from scldm.vae import TransformerVAE  # noqa: E402

from oracle.weights import make_state_dict, shapes_of  # noqa: E402

DIT_CASES = {
    # name: (kwargs, batch, seed)
    "dit_tiny": (dict(n_embed=64, n_embed_input=16, n_layer=2, n_head=4, seq_len=16, class_vocab_sizes={"a": 5},
                      condition_strategy="mutually_exclusive"), 4, 101),
    "dit_base": (dict(n_embed=256, n_embed_input=16, n_layer=8, n_head=8, seq_len=16,
                      class_vocab_sizes={"clusters": 14}, condition_strategy="mutually_exclusive"), 2, 102),
    "dit_joint": (dict(n_embed=256, n_embed_input=16, n_layer=8, n_head=8, seq_len=16,
                       class_vocab_sizes={"cell_type": 18, "cytokine": 91}, condition_strategy="joint"), 2, 103),
    "dit_me2": (dict(n_embed=64, n_embed_input=16, n_layer=2, n_head=4, seq_len=16,
                     class_vocab_sizes={"a": 5, "b": 7}, condition_strategy="mutually_exclusive"), 3, 104),
}
# generated AFTER every other fixture (constructing a module consumes torch's global RNG, which gen_init depends on)
LATE_DIT_CASES = {
    # 256-wide two-class mutually-exclusive model: the fused forward_with_cfg / sample_ode path with TWO conditional passes
    # (nnets.py:372-376), which the 64-wide dit_me2 cannot reach (the fused kernels are specialised to n_embed 256)
    "dit_me2_256": (dict(n_embed=256, n_embed_input=16, n_layer=2, n_head=8, seq_len=16,
                         class_vocab_sizes={"a": 5, "b": 7}, condition_strategy="mutually_exclusive"), 3, 105),
}
COMMON = dict(dropout=0.0, bias=True, norm_layer="layernorm", multiple_of=4, layernorm_eps=1e-8, cfg_dropout_prob=0.8)


def build_dit(kwargs, seed):
    m = DiT(**kwargs, **COMMON)
    shapes = shapes_of(m)
    m.load_state_dict(make_state_dict(shapes, seed), strict=True)
    m.eval()
    return m, shapes


def gen_dit(name, kwargs, B, seed):
    m, shapes = build_dit(kwargs, seed)
    rng = np.random.default_rng(seed + 1000)
    S, C = kwargs["seq_len"], kwargs["n_embed_input"]
    out = {"shapes_json": np.array(json.dumps({k: list(v) for k, v in shapes.items()})),
           "kwargs_json": np.array(json.dumps({**kwargs, **COMMON})), "seed": np.array(seed)}
    # ---- plain forward with per-sample t and labels (single-class dicts for mutually_exclusive) ----
    x = rng.standard_normal((B, S, C)).astype(np.float32)
    t = rng.uniform(0, 1, (B,)).astype(np.float32)
    labels = {k: rng.integers(0, v, (B,)).astype(np.int64) for k, v in kwargs["class_vocab_sizes"].items()}
    out["fwd_x"], out["fwd_t"] = x, t
    for k, v in labels.items():
        out[f"fwd_label_{k}"] = v
    taps = {}
    blk = m.blocks[0]
    hooks = [
        blk.attn.register_forward_hook(lambda mod, i, o: taps.__setitem__("block0.attn_out", o.detach().numpy().copy())),
        blk.attn.register_forward_pre_hook(lambda mod, i: taps.__setitem__("block0.mod1", i[0].detach().numpy().copy())),
        blk.mlp.register_forward_pre_hook(lambda mod, i: taps.__setitem__("block0.mod2", i[0].detach().numpy().copy())),
        blk.mlp.register_forward_hook(lambda mod, i, o: taps.__setitem__("block0.mlp_out", o.detach().numpy().copy())),
        blk.register_forward_hook(lambda mod, i, o: taps.__setitem__("block0.out", o.detach().numpy().copy())),
    ]
    names = sorted(kwargs["class_vocab_sizes"])
    with torch.no_grad():
        if kwargs["condition_strategy"] == "joint":
            cond = {k: torch.from_numpy(v) for k, v in labels.items()}
        else:
            cond = {names[0]: torch.from_numpy(labels[names[0]])}  # one available class -> deterministic (nnets.py:395)
        y = m(torch.from_numpy(x), torch.from_numpy(t), cond, force_drop_ids=False)
    for h in hooks:
        h.remove()
    out["fwd_out"] = y.numpy()
    out["fwd_classes"] = np.array(json.dumps(sorted(cond.keys())))
    for k, v in taps.items():
        out[f"tap_{k}"] = v
    # ---- forward_with_cfg on a doubled batch, scalar-broadcast t (integrators.py:103-104) ----
    z = rng.standard_normal((B, S, C)).astype(np.float32)
    x2 = np.concatenate([z, z], 0)
    t2 = np.full((2 * B,), 0.37, np.float32)
    lab2 = {k: np.concatenate([v, v]) for k, v in labels.items()}
    out["cfg_x"], out["cfg_t"] = x2, t2
    for k, v in lab2.items():
        out[f"cfg_label_{k}"] = v
    for tag, scales in (("s1", {k: 1.0 for k in names}), ("s2", {k: 2.0 - 0.5 * i for i, k in enumerate(names)})):
        with torch.no_grad():
            y = m.forward_with_cfg(torch.from_numpy(x2), torch.from_numpy(t2),
                                   condition={k: torch.from_numpy(v) for k, v in lab2.items()}, cfg_scale=scales)
        out[f"cfg_out_{tag}"] = y.numpy()
        out[f"cfg_scales_{tag}"] = np.array(json.dumps(scales))
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    print(name, "fwd |y|max", float(np.abs(out["fwd_out"]).max()), "cfg |y|max", float(np.abs(out["cfg_out_s2"]).max()))
    return m


def gen_transport(m, kwargs, seed):
    """Transport.training_losses (Linear/velocity) with x0, t injected."""
    tr = create_transport(path_type="Linear", prediction="velocity", loss_weight="velocity", train_eps=1e-5, sample_eps=1e-5)
    assert tr.train_eps == 0 and tr.sample_eps == 0
    rng = np.random.default_rng(seed + 2000)
    B, S, C = 3, kwargs["seq_len"], kwargs["n_embed_input"]
    x1 = rng.standard_normal((B, S, C)).astype(np.float32)
    x0 = rng.standard_normal((B, S, C)).astype(np.float32)
    t = rng.uniform(0, 1, (B,)).astype(np.float32)
    lab = rng.integers(0, 5, (B,)).astype(np.int64)
    tr.sample = lambda x1_: (torch.from_numpy(t), torch.from_numpy(x0), x1_)  # replaces the RNG draws (transport.py:97-108)
    with torch.no_grad():
        terms = tr.training_losses(lambda xt, tt, **kw: m(xt, tt, kw["condition"], force_drop_ids=False),
                                   torch.from_numpy(x1), {"condition": {"a": torch.from_numpy(lab)}})
    np.savez_compressed(os.path.join(HERE, "transport_tiny.npz"), x1=x1, x0=x0, t=t, label_a=lab,
                        pred=terms["pred"].numpy(), loss=terms["loss"].numpy())
    print("transport loss", terms["loss"].numpy())


TRAIN_CASES = {
    # name: (kwargs, batch, seed) -- gradients of mean(training_losses) w.r.t. every parameter
    "train_tiny": (dict(n_embed=64, n_embed_input=16, n_layer=2, n_head=4, seq_len=16, class_vocab_sizes={"a": 5},
                        condition_strategy="mutually_exclusive"), 3, 301),
    "train_base2": (dict(n_embed=256, n_embed_input=16, n_layer=2, n_head=8, seq_len=16,
                         class_vocab_sizes={"cell_type": 18, "cytokine": 91}, condition_strategy="joint"), 5, 302),
}
GRAD_SAMPLE = 64  # strided entries kept per gradient tensor (plus its sum and L2 norm)


def grad_digest(g: np.ndarray) -> np.ndarray:
    """[sum, l2, GRAD_SAMPLE strided entries] -- small enough to commit, sensitive to any misplaced entry."""
    f = g.reshape(-1).astype(np.float64)
    stride = max(1, f.size // GRAD_SAMPLE)
    smp = f[::stride][:GRAD_SAMPLE]
    smp = np.pad(smp, (0, GRAD_SAMPLE - smp.size))
    return np.concatenate([[f.sum(), np.sqrt((f * f).sum())], smp]).astype(np.float64)


def gen_train(name, kwargs, B, seed):
    """Transport.training_losses forward + autograd backward through the reference DiT (transport.py:110-150)."""
    m, shapes = build_dit(kwargs, seed)
    for p in m.parameters():
        if p.dtype.is_floating_point:
            p.grad = None
    tr = create_transport(path_type="Linear", prediction="velocity", loss_weight="velocity", train_eps=1e-5, sample_eps=1e-5)
    rng = np.random.default_rng(seed + 3000)
    S, C = kwargs["seq_len"], kwargs["n_embed_input"]
    x1 = rng.standard_normal((B, S, C)).astype(np.float32)
    x0 = rng.standard_normal((B, S, C)).astype(np.float32)
    t = rng.uniform(0, 1, (B,)).astype(np.float32)
    labels = {k: rng.integers(0, v, (B,)).astype(np.int64) for k, v in kwargs["class_vocab_sizes"].items()}
    # one class per cell replaced by its null token, as training-mode label dropout would do (nnets.py:300-334, index == vocab size)
    k0 = sorted(labels)[0]
    labels[k0][B // 2] = kwargs["class_vocab_sizes"][k0]
    tr.sample = lambda x1_: (torch.from_numpy(t), torch.from_numpy(x0), x1_)
    cond = {k: torch.from_numpy(v) for k, v in labels.items()}
    terms = tr.training_losses(lambda xt, tt, **kw: m(xt, tt, kw["condition"], force_drop_ids=False),
                               torch.from_numpy(x1), {"condition": cond})
    terms["loss"].mean().backward()
    out = {"kwargs_json": np.array(json.dumps({**kwargs, **COMMON})), "seed": np.array(seed), "x1": x1, "x0": x0, "t": t,
           "shapes_json": np.array(json.dumps({k: list(v) for k, v in shapes.items()})),
           "pred": terms["pred"].detach().numpy(), "loss": terms["loss"].detach().numpy()}
    for k, v in labels.items():
        out[f"label_{k}"] = v
    frozen = []
    for k, p in m.named_parameters():
        if p.grad is None:
            frozen.append(k)
            continue
        out[f"grad_{k}"] = grad_digest(p.grad.numpy())
    out["frozen_json"] = np.array(json.dumps(frozen))
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    print(name, "loss", terms["loss"].detach().numpy(), "frozen", frozen, "n grads", sum(k.startswith("grad_") for k in out))


def _import_reference_tokenizer():
    """scldm.datamodule.tokenize_cells is pure NumPy, but its module imports the data stack (anndata, lightning,
    cellarium-ml) at import time; give those import-time-only stand-ins (nothing of them is called by tokenize_cells) and
    a StrEnum shim for Python 3.10 so scldm.constants imports for real."""
    import enum
    if not hasattr(enum, "StrEnum"):
        class StrEnum(str, enum.Enum):
            def __str__(self):
                return str(self.value)
        enum.StrEnum = StrEnum

    class _Unused:
        def __init__(self, *a, **k):
            raise RuntimeError("import-time stand-in: must not be used")

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules.setdefault(name, m)
    mod("anndata", AnnData=_Unused)
    mod("pytorch_lightning", LightningDataModule=object)
    mod("cellarium")
    mod("cellarium.ml")
    mod("cellarium.ml.data", DistributedAnnDataCollection=_Unused, IterableDistributedAnnDataCollectionDataset=_Unused)
    mod("cellarium.ml.utilities")
    mod("cellarium.ml.utilities.data", AnnDataField=_Unused, convert_to_tensor=lambda x: x)
    mod("scldm._utils", get_tissue_adata_files=None, sort_h5ad_files=None)
    mod("scldm.encoder", VocabularyEncoderSimplified=object)
    from scldm.datamodule import tokenize_cells
    return tokenize_cells


TOKENIZE_CASES = {"tok_small": (5, 40, 12, 401), "tok_dentate": (3, 17002, 6147, 402)}   # name: (N, G, genes_seq_len, seed)


def gen_tokenize(name, N, G, S, seed):
    """tokenize_cells(sample_genes="expressed") (src/scldm/datamodule.py:660-731): expressed genes compacted to the front of
    a genes_seq_len window, mask token / zero padding; plus library_size."""
    tokenize_cells = _import_reference_tokenizer()
    rng = np.random.default_rng(seed)
    rate = min(0.25, 0.8 * S / G)
    counts = (rng.poisson(0.9, (N, G)) * (rng.random((N, G)) < rate)).astype(np.float32)
    counts[0, :] = 0                                   # a cell with nothing expressed
    if name == "tok_small":
        counts[1, :] = 0
        counts[1, :S] = 3                              # exactly genes_seq_len expressed genes
    gene_ids = rng.permutation(G + 1)[:G].astype(np.int64) + 1   # token ids 1..G+1 in arbitrary order; 0 = mask token

    class Enc:                                         # the two members tokenize_cells uses (encoder.py:20,141-148)
        mask_token_idx = 0

        @staticmethod
        def encode_genes(tokens):
            return gene_ids

    out = tokenize_cells(counts, [str(i) for i in range(G)], Enc, S, "expressed")
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), counts=counts, gene_ids=gene_ids, genes_seq_len=np.array(S),
                        mask_idx=np.array(0), genes_subset=out["genes_subset"], counts_subset=out["counts_subset"],
                        library_size=out["library_size"])
    print(name, "expressed per cell", (counts > 0).sum(1), "lib", out["library_size"][:, 0])


SIZE_FACTOR_CASES = {
    # name -> (condition_strategy, vocabulary-encoder attributes, condition labels).  Standard deviations are 1e-12 so the
    # reference's Normal(loc, scale).sample() returns loc to fp32 precision: the fixture pins the LOOKUP logic (key choice, joint
    # keys, missing statistics -> 0); the draw itself is checked statistically on the GPU.
    "independent_key": ("mutually_exclusive",
                        dict(size_factor_condition_key="cell_type",
                             mu_size_factor={"cell_type": {0: 7.5, 1: 8.25, 3: 6.125}, "donor": {0: 1.0}},
                             sd_size_factor={"cell_type": {0: 1e-12, 1: 1e-12, 3: 1e-12}, "donor": {0: 1e-12}}),
                        {"cell_type": [0, 1, 2, 3, 1, 0], "donor": [0, 0, 0, 0, 0, 0]}),
    "inferred_key": ("mutually_exclusive",
                     dict(mu_size_factor={"zeta": {0: 3.0, 1: 4.0}, "beta": {0: 5.0, 1: 5.5, 2: 9.0}},
                          sd_size_factor={"zeta": {0: 1e-12, 1: 1e-12}, "beta": {0: 1e-12, 1: 1e-12, 2: 1e-12}}),
                     {"zeta": [0, 1, 1, 0], "beta": [2, 0, 1, 1]}),
    "joint": ("joint",
              dict(joint_key="ct_cyt", joint_components=["cell_type", "cytokine"],
                   joint_idx_2_classes={"0_0": "A_x", "0_1": "A_y", "1_0": "B_x", "2_1": "C_y"},
                   mu_size_factor={"ct_cyt": {"A_x": 6.5, "A_y": 7.0, "B_x": 8.0}},
                   sd_size_factor={"ct_cyt": {"A_x": 1e-12, "A_y": 1e-12, "B_x": 1e-12, "C_y": 1e-12}}),
              {"cell_type": [0, 0, 1, 1, 2, 0], "cytokine": [0, 1, 0, 1, 1, 0]}),
    "no_stats": ("joint", dict(), {"cell_type": [0, 1], "cytokine": [1, 0]}),
    "no_matching_key": ("mutually_exclusive", dict(mu_size_factor={"other": {0: 1.0}}, sd_size_factor={"other": {0: 1e-12}}),
                        {"cell_type": [0, 1, 0]}),
}


def gen_size_factors():
    """LatentDiffusion._sample_log_size_factors (src/scldm/models.py:473-597).  scldm.models cannot be imported here (lightning,
    scvi, ema_pytorch, ...), so the METHOD is taken from the reference file with ast and executed as it stands against a duck-typed
    `self` (trainer.datamodule.vocabulary_encoder, diffusion_model.condition_strategy, device) - nothing of it is copied."""
    import ast
    import logging
    import textwrap
    from types import SimpleNamespace
    from typing import cast
    from torch.distributions import Normal
    path = os.path.join(REF, "models.py")
    src = open(path).read()
    fn_src = None
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.FunctionDef) and node.name == "_sample_log_size_factors":
            fn_src = textwrap.dedent(ast.get_source_segment(src, node))
    ns = {"torch": torch, "Normal": Normal, "cast": cast, "logger": logging.getLogger("golden")}
    exec(compile(fn_src, path, "exec"), ns)
    fn = ns["_sample_log_size_factors"]
    out = {"cases_json": np.array(json.dumps({k: [v[0], v[1], v[2]] for k, v in SIZE_FACTOR_CASES.items()}))}
    for name, (strategy, attrs, cond) in SIZE_FACTOR_CASES.items():
        enc = SimpleNamespace(**attrs)
        me = SimpleNamespace(trainer=SimpleNamespace(datamodule=SimpleNamespace(vocabulary_encoder=enc)),
                             diffusion_model=SimpleNamespace(condition_strategy=strategy), device=torch.device("cpu"))
        condition = {k: torch.tensor(v) for k, v in cond.items()}
        B = len(next(iter(cond.values())))
        res = fn(me, condition, B)
        out[f"out_{name}"] = res.numpy()
        print("size factors", name, res.numpy())
    res = fn(SimpleNamespace(trainer=SimpleNamespace(datamodule=SimpleNamespace(vocabulary_encoder=SimpleNamespace())),
                             diffusion_model=SimpleNamespace(condition_strategy="joint"), device=torch.device("cpu")), None, 4)
    out["out_condition_none"] = res.numpy()
    np.savez_compressed(os.path.join(HERE, "size_factors.npz"), **out)


MMD_CASES = {"mmd_small": (5, 7, 33, 501), "mmd_counts": (48, 40, 700, 502)}   # name: (Bx, By, D, seed)


def gen_mmd(name, Bx, By, D, seed):
    """Kernel matrices and MMDLoss values of src/scldm/evaluations.py:10-82 (its `import ot` gets an import-time stand-in;
    `wasserstein`, the only user of POT, is not exercised)."""
    sys.modules.setdefault("ot", types.ModuleType("ot"))
    from scldm.evaluations import BrayCurtisKernel, MMDLoss, RBFKernel, RuzickaKernel, TanimotoKernel
    rng = np.random.default_rng(seed)
    cx = (rng.poisson(1.5, (Bx, D)) * (rng.random((Bx, D)) < 0.3)).astype(np.float32)
    cy = (rng.poisson(1.2, (By, D)) * (rng.random((By, D)) < 0.3)).astype(np.float32)
    cx[0] = 0                                                        # an empty cell: denominators hit the 1e-8 guard
    lx = np.log1p(cx / np.maximum(cx.sum(1, keepdims=True), 1) * 1e4).astype(np.float32)   # models.py:899-900 scaling
    ly = np.log1p(cy / np.maximum(cy.sum(1, keepdims=True), 1) * 1e4).astype(np.float32)
    zx = rng.standard_normal((Bx, D)).astype(np.float32) * 0.05      # signed data at a scale where the RBF kernel is not all ~0
    zy = rng.standard_normal((By, D)).astype(np.float32) * 0.05 + 0.01
    out = {"cx": cx, "cy": cy, "lx": lx, "ly": ly, "zx": zx, "zy": zy}
    t = torch.from_numpy
    for tag, kern, (a, b) in (("rbf", RBFKernel(), (zx, zy)), ("rbf_s", RBFKernel(scale=0.37), (zx, zy)),
                              ("braycurtis", BrayCurtisKernel(), (lx, ly)), ("braycurtis_signed", BrayCurtisKernel(), (zx, zy)),
                              ("tanimoto", TanimotoKernel(), (cx, cy)), ("ruzicka", RuzickaKernel(), (lx, ly)),
                              ("ruzicka_signed", RuzickaKernel(), (zx, zy))):
        out[f"k_{tag}"] = kern(t(a), t(b)).numpy()
        out[f"mmd_{tag}"] = MMDLoss(kern)(t(a), t(b)).numpy()
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    print(name, {k: float(v) for k, v in out.items() if k.startswith("mmd_")})


def gen_init():
    """DiT.initialize_weights (src/scldm/nnets.py:458-492): the frozen sin|cos pos_embed exactly, and per-parameter statistics of a
    fresh reference module (which tensors are exactly zero, std / abs-max of the random ones)."""
    torch.manual_seed(1234)
    kw = dict(n_embed=256, n_embed_input=16, n_layer=2, n_head=8, seq_len=16, class_vocab_sizes={"clusters": 14},
              condition_strategy="mutually_exclusive")
    m = DiT(**kw, **COMMON)
    stats = {k: [float(v.float().mean()), float(v.float().std()), float(v.abs().max())] for k, v in m.state_dict().items()}
    np.savez_compressed(os.path.join(HERE, "dit_init.npz"), pos_embed=m.pos_embed.detach().numpy(),
                        kwargs_json=np.array(json.dumps({**kw, **COMMON})), stats_json=np.array(json.dumps(stats)),
                        requires_grad_json=np.array(json.dumps({k: bool(p.requires_grad) for k, p in m.named_parameters()})))
    print("init: zero tensors", [k for k, v in stats.items() if v[2] == 0.0])


VAE_CASES = {"vae_small": (dict(n_genes=60), 50, 20, 2, 201), "vae_2000": (dict(n_genes=2000), 2000, 2000, 2, 202)}
# round 6: decoder_name negative_binomial_unshared_theta (stochastic_layers.py:94-96): params Linear(32, 2), no theta table
LATE_VAE_CASES = {"vae_unshared": (dict(n_genes=300, shared_theta=False), 300, 120, 3, 203)}


def gen_vae(name, n_genes, G, S, B, seed, shared_theta=True):
    enc = Encoder(n_layer=8, n_inducing_points=16, n_embed=32, n_embed_latent=16, n_head=8, n_head_cross=4, dropout=0.0,
                  bias=False, multiple_of=4, layernorm_eps=1e-8, norm_layer="layernorm", positional_encoding=True)
    dec = Decoder(n_genes=n_genes, n_embed=32, n_embed_latent=16, n_head=8, n_head_cross=4, n_layer=8, n_inducing_points=16,
                  dropout=0.0, bias=False, multiple_of=4, layernorm_eps=1e-8, norm_layer="layernorm", shared_embedding=True,
                  use_adaln=False)
    head = NegativeBinomialTransformerLayer(n_genes=n_genes, shared_theta=shared_theta, n_embed=32, norm_layer="layernorm",
                                            layernorm_eps=1e-8)
    inp = InputTransformerVAE(n_genes=n_genes, n_embed=32, agg_func="log1p")
    vae = TransformerVAE(encoder=enc, decoder=dec, decoder_head=head, input_layer=inp)
    shapes = shapes_of(vae)
    vae.load_state_dict(make_state_dict(shapes, seed), strict=True)
    vae.eval()
    rng = np.random.default_rng(seed + 1000)
    genes = np.stack([rng.permutation(n_genes)[:G] for _ in range(B)]).astype(np.int64)
    counts = rng.poisson(0.7, (B, G)).astype(np.float32)
    sub = np.stack([np.sort(rng.permutation(G)[:S]) for _ in range(B)])
    genes_subset = np.take_along_axis(genes, sub, 1)
    counts_subset = np.take_along_axis(counts, sub, 1)
    lib = counts.sum(1, keepdims=True).astype(np.float32) + 1.0
    with torch.no_grad():
        z = vae.encode(torch.from_numpy(counts), torch.from_numpy(genes), torch.from_numpy(counts_subset),
                       torch.from_numpy(genes_subset))
        nb = vae.decode(z, torch.from_numpy(genes), torch.from_numpy(lib))
        zrand = torch.from_numpy(rng.standard_normal((B, 16, 16)).astype(np.float32))
        nb2 = vae.decode(zrand, torch.from_numpy(genes), torch.from_numpy(lib))
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"),
                        shapes_json=np.array(json.dumps({k: list(v) for k, v in shapes.items()})), seed=np.array(seed),
                        n_genes=np.array(n_genes), genes=genes, counts=counts, genes_subset=genes_subset,
                        counts_subset=counts_subset, library_size=lib, z=z.numpy(), mu=nb.mu.numpy(), theta=nb.theta.numpy(),
                        zrand=zrand.numpy(), mu_rand=nb2.mu.numpy())
    print(name, "z absmax", float(z.abs().max()), "mu rowsum", nb.mu.sum(1).numpy(), "lib", lib[:, 0])


def gen_vae_train(name, n_genes, G, S, B, seed):
    """TransformerVAE.forward + VAE.loss (-log_nb_positive(counts, mu, theta).sum(1).mean(): src/scldm/vae.py:29-56,
    src/scldm/models.py:231-249, src/scldm/distributions.py:6-42) and the reference's own autograd gradient of every parameter
    (digests), on the inputs of gen_vae's case of the same size.  BASELINE configs[0] / VERDICT r2 row V1."""
    from scldm.distributions import log_nb_positive
    enc = Encoder(n_layer=8, n_inducing_points=16, n_embed=32, n_embed_latent=16, n_head=8, n_head_cross=4, dropout=0.0,
                  bias=False, multiple_of=4, layernorm_eps=1e-8, norm_layer="layernorm", positional_encoding=True)
    dec = Decoder(n_genes=n_genes, n_embed=32, n_embed_latent=16, n_head=8, n_head_cross=4, n_layer=8, n_inducing_points=16,
                  dropout=0.0, bias=False, multiple_of=4, layernorm_eps=1e-8, norm_layer="layernorm", shared_embedding=True,
                  use_adaln=False)
    head = NegativeBinomialTransformerLayer(n_genes=n_genes, shared_theta=True, n_embed=32, norm_layer="layernorm",
                                            layernorm_eps=1e-8)
    inp = InputTransformerVAE(n_genes=n_genes, n_embed=32, agg_func="log1p")
    vae = TransformerVAE(encoder=enc, decoder=dec, decoder_head=head, input_layer=inp)
    shapes = shapes_of(vae)
    vae.load_state_dict(make_state_dict(shapes, seed), strict=True)
    vae.train()
    rng = np.random.default_rng(seed + 1000)
    genes = np.stack([rng.permutation(n_genes)[:G] for _ in range(B)]).astype(np.int64)
    counts = rng.poisson(0.7, (B, G)).astype(np.float32)
    sub = np.stack([np.sort(rng.permutation(G)[:S]) for _ in range(B)])
    genes_subset = np.take_along_axis(genes, sub, 1)
    counts_subset = np.take_along_axis(counts, sub, 1)
    lib = counts.sum(1, keepdims=True).astype(np.float32) + 1.0
    params, z = vae(torch.from_numpy(counts), torch.from_numpy(genes), torch.from_numpy(lib), torch.from_numpy(counts_subset),
                    torch.from_numpy(genes_subset))
    recon = -log_nb_positive(torch.from_numpy(counts), params["mu"], params["theta"])
    loss = recon.sum(dim=1).mean()
    loss.backward()
    out = {"shapes_json": np.array(json.dumps({k: list(v) for k, v in shapes.items()})), "seed": np.array(seed), "n_genes": np.array(n_genes),
           "genes": genes, "counts": counts, "genes_subset": genes_subset, "counts_subset": counts_subset, "library_size": lib,
           "loss": np.array(float(loss.detach())), "recon_row": recon.sum(dim=1).detach().numpy(), "z": z.detach().numpy(),
           "mu": params["mu"].detach().numpy(), "theta": params["theta"].detach().numpy()}
    frozen = []
    for k, p in vae.named_parameters():
        if p.grad is None:
            frozen.append(k)
            continue
        out[f"grad_{k}"] = grad_digest(p.grad.numpy())
    out["frozen_json"] = np.array(json.dumps(frozen))
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    print(name, "loss", float(loss), "frozen", frozen, "n grads", sum(k.startswith("grad_") for k in out))


VAE_TRAIN_CASES = {"vae_train_small": (dict(n_genes=60), 50, 20, 3, 211), "vae_train_2000": (dict(n_genes=2000), 2000, 700, 2, 212)}


if __name__ == "__main__":
    torch.manual_seed(0)
    models = {}
    for name, (kw, B, seed) in DIT_CASES.items():
        models[name] = gen_dit(name, kw, B, seed)
    gen_transport(models["dit_tiny"], DIT_CASES["dit_tiny"][0], 101)
    for name, (kw, B, seed) in TRAIN_CASES.items():
        gen_train(name, kw, B, seed)
    for name, (N, G, S, seed) in TOKENIZE_CASES.items():
        gen_tokenize(name, N, G, S, seed)
    gen_size_factors()
    gen_init()
    for name, (Bx, By, D, seed) in MMD_CASES.items():
        gen_mmd(name, Bx, By, D, seed)
    for name, (kw, G, S, B, seed) in VAE_CASES.items():
        gen_vae(name, kw["n_genes"], G, S, B, seed, kw.get("shared_theta", True))
    for name, (kw, B, seed) in LATE_DIT_CASES.items():
        gen_dit(name, kw, B, seed)
    for name, (kw, G, S, B, seed) in VAE_TRAIN_CASES.items():     # (last: everything above stays bit-identical to earlier rounds)
        gen_vae_train(name, kw["n_genes"], G, S, B, seed)
    for name, (kw, G, S, B, seed) in LATE_VAE_CASES.items():
        gen_vae(name, kw["n_genes"], G, S, B, seed, kw.get("shared_theta", True))
