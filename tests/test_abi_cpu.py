"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol the header
declares, argument validation works without a GPU, and the Python mirror keeps the reference's
constructor kwargs / state_dict keys (pinned by the key+shape lists stored in the golden fixtures)."""
import ctypes as C
import os
import re

import pytest
import torch

from conftest import ROOT, golden_json, load_golden
from oracle.weights import make_state_dict


def test_library_exports_every_declared_symbol():
    from scldm_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "scldm_hip.h")).read()
    declared = set(re.findall(r"\b(scldm_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no prototypes parsed"
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), f"libscldm_hip.so does not export {name}"
    assert declared == set(_lib.EXPORTS)
    assert L.scldm_version() == 5


def test_create_rejects_unsupported_shapes_without_gpu():
    from scldm_amd import _lib
    L = _lib.lib()
    cfg = _lib.DitConfig(n_embed=64, n_embed_input=16, n_layer=2, n_head=4, seq_len=16, hidden_dim=172, layernorm_eps=1e-8, n_classes=0)
    h = C.c_void_p()
    rc = L.scldm_dit_create(C.byref(cfg), C.byref(h))
    assert rc == -1 and b"n_embed=256" in L.scldm_last_error()
    with pytest.raises(_lib.ScldmError):
        _lib.check(rc, "scldm_dit_create")
    # a wider shape gets a handle for the generic (training-kernel) path only: the fused entry points refuse it
    cfg = _lib.DitConfig(n_embed=1024, n_embed_input=16, n_layer=24, n_head=16, seq_len=16, hidden_dim=2732, layernorm_eps=1e-8, n_classes=0)
    assert L.scldm_dit_create(C.byref(cfg), C.byref(h)) == 0
    assert L.scldm_dit_train_saved_bytes(h, 2) > 0 and L.scldm_dit_workspace_bytes(h, 2, 2, 0) >= 0
    w = _lib.DitWeights()
    assert L.scldm_dit_load_weights(h, C.byref(w), None) == -1 and b"fused DiT layer" in L.scldm_last_error()
    L.scldm_dit_destroy(h)


@pytest.mark.parametrize("name", ["dit_base", "dit_joint"])
def test_dit_state_dict_is_checkpoint_compatible(name):
    from scldm_amd.nnets import DiT
    g = load_golden(name)
    kw = golden_json(g, "kwargs_json")
    shapes = {k: tuple(v) for k, v in golden_json(g, "shapes_json").items()}
    m = DiT(**kw)
    ours = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert ours == shapes  # same keys, same shapes as the reference module
    m.load_state_dict(make_state_dict(shapes, int(g["seed"])), strict=True)
    assert m.seq_len == 16 and m.condition_strategy == kw["condition_strategy"] and m.n_embed == 256
    assert m.class_vocab_sizes == kw["class_vocab_sizes"] and m.cfg_dropout_prob == 0.8


def test_fresh_dit_matches_reference_init_structure():
    from scldm_amd.nnets import DiT
    from scldm_amd.layers import sincos_pos_embed
    torch.manual_seed(0)
    m = DiT(n_embed=256, n_embed_input=16, n_layer=2, n_head=8, seq_len=16, dropout=0.0, bias=True, norm_layer="layernorm",
            multiple_of=4, layernorm_eps=1e-8, class_vocab_sizes={"clusters": 14}, cfg_dropout_prob=0.8)
    # adaLN-Zero: zero modulation + zero output layer (nnets.py:480-492); sin|cos pos table (layers.py:383)
    assert float(m.blocks[0].adaln_modulation[1].weight.abs().sum()) == 0.0
    assert float(m.final_layer.linear.weight.abs().sum()) == 0.0
    pe = sincos_pos_embed(256, 16)
    assert pe.shape == (16, 256) and abs(pe[0, 0]) < 1e-12 and abs(pe[0, 128] - 1.0) < 1e-12
    assert torch.allclose(m.pos_embed[0], torch.from_numpy(pe).float())
    assert m.class_embeddings["clusters"].weight.shape == (15, 256)
    assert not m.pos_embed.requires_grad


def test_no_cpu_fallback():
    from scldm_amd.nnets import DiT
    m = DiT(n_embed=256, n_embed_input=16, n_layer=1, n_head=8, seq_len=16, dropout=0.0, bias=True, norm_layer="layernorm",
            multiple_of=4, layernorm_eps=1e-8, class_vocab_sizes={"a": 3}).eval()
    with pytest.raises(RuntimeError, match="CUDA"):
        m(torch.zeros(1, 16, 16), torch.zeros(1), {"a": torch.zeros(1, dtype=torch.long)})
    with pytest.raises(RuntimeError, match="parameter container"):
        m.blocks[0](torch.zeros(1, 16, 256))


def test_transport_face_and_training_losses_cpu():
    """Transport.training_losses is plain tensor math around the model call: check it against the reference fixture
    with the oracle DiT standing in as `model` (checker role only)."""
    from scldm_amd.transport import Sampler, create_transport
    from oracle.dit import dit_forward
    from test_oracle_dit import setup
    from conftest import max_abs_rel
    g = load_golden("transport_tiny")
    _, cfg, sd = setup("dit_tiny")
    tr = create_transport(path_type="Linear", prediction="velocity", loss_weight="velocity", train_eps=1e-5, sample_eps=1e-5)
    assert tr.train_eps == 0 and tr.sample_eps == 0
    tr.sample = lambda x1: (torch.from_numpy(g["t"]), torch.from_numpy(g["x0"]), x1)
    lab = torch.from_numpy(g["label_a"])
    out = tr.training_losses(lambda xt, t, condition: dit_forward(sd, cfg, xt, t, condition), torch.from_numpy(g["x1"]),
                             {"condition": {"a": lab}})
    assert max_abs_rel(out["loss"], g["loss"]) < 2e-5 and max_abs_rel(out["pred"], g["pred"]) < 2e-5
    with pytest.raises(NotImplementedError):
        create_transport(path_type="VP")
    with pytest.raises(NotImplementedError):
        Sampler(tr).sample_ode(sampling_method="rk4")          # only euler / heun / dopri5 (the reference's own choices)
    with pytest.raises(NotImplementedError):
        Sampler(tr).sample_ode(reverse=True)
    # generic python ODE loop: KAT on f = -x, and grid semantics (N points -> N-1 evaluations)
    seen = []
    fn = Sampler(tr).sample_ode(sampling_method="euler", num_steps=5)
    traj = fn(torch.ones(2, 3), lambda x, t: (seen.append(float(t[0])), -x)[1])
    assert seen == [0.0, 0.25, 0.5, 0.75] and traj.shape == (5, 2, 3)
    assert torch.equal(traj[-1], torch.full((2, 3), 0.31640625))


def _build_vae(n_genes, shared_theta=True):
    from scldm_amd.layers import InputTransformerVAE
    from scldm_amd.nnets import Decoder, Encoder
    from scldm_amd.stochastic_layers import NegativeBinomialTransformerLayer
    from scldm_amd.vae import TransformerVAE
    enc = Encoder(n_layer=8, n_inducing_points=16, n_embed=32, n_embed_latent=16, n_head=8, n_head_cross=4, dropout=0.0, bias=False,
                  multiple_of=4, layernorm_eps=1e-8, norm_layer="layernorm", positional_encoding=True)
    dec = Decoder(n_genes=n_genes, n_embed=32, n_embed_latent=16, n_head=8, n_head_cross=4, n_layer=8, n_inducing_points=16,
                  dropout=0.0, bias=False, multiple_of=4, layernorm_eps=1e-8, norm_layer="layernorm", shared_embedding=True,
                  use_adaln=False)
    head = NegativeBinomialTransformerLayer(n_genes=n_genes, shared_theta=shared_theta, n_embed=32, norm_layer="layernorm", layernorm_eps=1e-8)
    inp = InputTransformerVAE(n_genes=n_genes, n_embed=32, agg_func="log1p")
    return TransformerVAE(encoder=enc, decoder=dec, decoder_head=head, input_layer=inp)


@pytest.mark.parametrize("name", ["vae_small", "vae_unshared"])
def test_vae_state_dict_is_checkpoint_compatible(name):
    g = load_golden(name)
    shapes = {k: tuple(v) for k, v in golden_json(g, "shapes_json").items()}
    vae = _build_vae(int(g["n_genes"]), shared_theta="decoder_head.theta.weight" in shapes)
    ours = {k: tuple(v.shape) for k, v in vae.state_dict().items()}
    assert ours == shapes
    vae.load_state_dict(make_state_dict(shapes, int(g["seed"])), strict=True)
    # attributes the reference's LatentDiffusion reads (models.py:789; vae.py:43,46,79,82)
    assert vae.encoder.latent_embedding == 16 and isinstance(vae.decoder.gene_embedding, torch.nn.Identity)
    assert vae.decoder_head.__class__.__name__ == "NegativeBinomialTransformerLayer"
    assert float(_build_vae(5).decoder_head.theta.weight.min()) == 1.0  # ones init (stochastic_layers.py:93)
    with pytest.raises(RuntimeError, match="CUDA"):
        vae.encode(torch.zeros(1, 4), torch.zeros(1, 4, dtype=torch.long))


def test_negative_binomial_holder_has_no_cpu_draw():
    """The draw is a HIP kernel (scldm_nb_sample); statistical tests live in tests/test_gpu_vae.py."""
    from scldm_amd.stochastic_layers import NegativeBinomial
    nb = NegativeBinomial(mu=torch.full((4, 3), 5.0), theta=torch.full((4, 3), 2.0))
    assert nb.mean.shape == (4, 3)
    with pytest.raises(RuntimeError, match="CUDA"):
        nb.sample()


def test_fresh_dit_matches_reference_initialisation():
    """A9 (nnets.py:458-492) against a fresh reference module (tests/golden/dit_init.npz): identical frozen pos_embed, the same
    tensors exactly zero (adaLN-Zero, output layer, Linear biases), the same trainable flags, and matching spread of the random ones."""
    from scldm_amd.nnets import DiT
    g = load_golden("dit_init")
    torch.manual_seed(99)
    m = DiT(**golden_json(g, "kwargs_json"))
    assert torch.allclose(m.pos_embed.detach(), torch.from_numpy(g["pos_embed"]), atol=1e-6)
    stats = golden_json(g, "stats_json")
    assert set(stats) == set(m.state_dict())
    assert golden_json(g, "requires_grad_json") == {k: bool(p.requires_grad) for k, p in m.named_parameters()}
    for k, v in m.state_dict().items():
        mean, std, amax = stats[k]
        if amax == 0.0:
            assert float(v.abs().max()) == 0.0, k
        elif k != "pos_embed":
            assert abs(float(v.float().std()) - std) < 0.08 * std, k          # same distribution family and scale
            assert float(v.abs().max()) < 1.6 * amax, k                       # xavier-uniform stays bounded, normal(0.02) tails


def test_adamw_launch_table_layout_is_built_on_the_host():
    """scldm_adamw_table_build / _update are host-side (no HIP call): [count x {p, g, m, v, ema, n} (48 B)] then one {tensor, chunk} per
    4 096-element workgroup; empty tensors own no workgroup; _update rewrites the records only."""
    import ctypes as C
    import struct
    from scldm_amd import _lib
    L = _lib.lib()
    sizes = [5, 4096, 0, 4097, 12288]
    ent = (_lib.AdamwEntry * len(sizes))()
    for i, n in enumerate(sizes):
        ent[i].p, ent[i].g, ent[i].m, ent[i].v, ent[i].n = 0x1000 * (i + 1), 0x2000 * (i + 1), 0x3000 * (i + 1), 0x4000 * (i + 1), n
    ema = (C.c_void_p * len(sizes))(*[0x5000 * (i + 1) for i in range(len(sizes))])
    nb = L.scldm_adamw_table_bytes(ent, len(sizes))
    blocks = [1, 1, 0, 2, 3]
    assert nb == 48 * len(sizes) + 8 * sum(blocks) and L.scldm_adamw_table_records_bytes(len(sizes)) == 48 * len(sizes)
    # gradient-norm clipping (version 5): [norm, coefficient, -, - | one partial per workgroup]; the launch struct grew by two fields
    assert L.scldm_adamw_clip_workspace_bytes(sum(blocks)) == 4 * (4 + sum(blocks)) and L.scldm_adamw_clip_workspace_bytes(0) == 0
    assert C.sizeof(_lib.AdamwLaunch) == 80 and _lib.AdamwLaunch.max_grad_norm.offset == 64 and _lib.AdamwLaunch.clip_ws.offset == 72
    buf = (C.c_char * nb)()
    n_blocks = C.c_int(0)
    _lib.check(L.scldm_adamw_table_build(ent, C.cast(ema, C.POINTER(C.c_void_p)), len(sizes), buf, nb, C.byref(n_blocks)), "build")
    assert n_blocks.value == sum(blocks)
    raw = bytes(buf)
    for i, n in enumerate(sizes):
        assert struct.unpack_from("<6q", raw, 48 * i) == (0x1000 * (i + 1), 0x2000 * (i + 1), 0x3000 * (i + 1), 0x4000 * (i + 1), 0x5000 * (i + 1), n)
    got = [struct.unpack_from("<2i", raw, 48 * len(sizes) + 8 * b) for b in range(n_blocks.value)]
    assert got == [(i, c) for i, k in enumerate(blocks) for c in range(k)]
    ent[3].g = 0x7777
    rec = (C.c_char * (48 * len(sizes)))()
    _lib.check(L.scldm_adamw_table_update(ent, None, len(sizes), rec, 48 * len(sizes)), "update")
    assert struct.unpack_from("<6q", bytes(rec), 48 * 3) == (0x4000, 0x7777, 0xC000, 0x10000, 0, 4097)
    with pytest.raises(_lib.ScldmError):
        _lib.check(L.scldm_adamw_table_build(ent, None, len(sizes), buf, 8, C.byref(n_blocks)), "too small")
    ent[1].v = None
    with pytest.raises(_lib.ScldmError, match="NULL"):
        _lib.check(L.scldm_adamw_table_build(ent, None, len(sizes), buf, nb, C.byref(n_blocks)), "null")


def test_weights_unchanged_context_nests_and_is_per_thread():
    """scldm_amd.nnets.weights_unchanged (the samplers' evaluation loops skip the device-side weight fingerprint pass inside it): a depth
    counter, restored on exit and on exceptions, private to the thread that entered it."""
    import threading
    from scldm_amd.nnets import _ASSUME_WEIGHTS_UNCHANGED as A, weights_unchanged
    assert A.depth == 0
    with weights_unchanged():
        seen = []
        t = threading.Thread(target=lambda: seen.append(A.depth)); t.start(); t.join()
        assert A.depth == 1 and seen == [0]
        with pytest.raises(RuntimeError):
            with weights_unchanged():
                assert A.depth == 2
                raise RuntimeError("x")
        assert A.depth == 1
    assert A.depth == 0


def test_round6_host_faces_fail_loudly_without_gpu():
    """The round-6 host entry points on CPU modules: the prediction loop and the fused training step refuse to run (no CPU path), the
    optimizer's extension of torch's signature validates its argument, and an optimizer on CPU parameters raises at step()."""
    from scldm_amd.nnets import DiT
    from scldm_amd.optim import AdamW
    from scldm_amd.sampling import generate_cells_stream
    from scldm_amd.training import FusedTrainStep
    from scldm_amd.transport import create_transport
    dit = DiT(n_embed=256, n_embed_input=16, n_layer=1, n_head=8, seq_len=16, dropout=0.0, bias=True, norm_layer="layernorm", multiple_of=4,
              layernorm_eps=1e-8, class_vocab_sizes={"clusters": 14}, cfg_dropout_prob=0.8)
    assert dit.deferred_label_check is False
    with pytest.raises(RuntimeError, match="CUDA"):
        next(generate_cells_stream(dit, None, [{"clusters": torch.zeros(4, dtype=torch.long)}], {"clusters": 1.0}, torch.zeros(4, 8, dtype=torch.long)))
    with pytest.raises(ValueError):
        AdamW(dit.parameters(), max_grad_norm=-1.0)
    opt = AdamW(dit.parameters(), lr=1e-3, max_grad_norm=10.0)
    assert opt.max_grad_norm == 10.0 and opt.last_grad_norm is None
    with pytest.raises(RuntimeError, match="CUDA"):
        FusedTrainStep(dit, create_transport("Linear", "velocity", "velocity", 1e-5, 1e-5), opt, 8, ["clusters"])
    for p in dit.parameters():
        p.grad = torch.zeros_like(p)
    with pytest.raises(RuntimeError, match="no CPU path"):
        opt.step()
