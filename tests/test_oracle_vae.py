"""Pin oracle/vae.py (MCAB encode/decode + NB head) against the reference fixtures."""
import pytest
import torch

from conftest import golden_json, load_golden, max_abs_rel
from oracle.vae import VAEConfig, decode, encode
from oracle.weights import make_state_dict


def setup(name, dtype=torch.float32):
    g = load_golden(name)
    shapes = {k: tuple(v) for k, v in golden_json(g, "shapes_json").items()}
    sd = make_state_dict(shapes, int(g["seed"]), dtype=dtype)
    return g, VAEConfig(n_genes=int(g["n_genes"])), sd


@pytest.mark.parametrize("name", ["vae_small", "vae_2000"])
def test_encode_decode_match_reference(name):
    g, cfg, sd = setup(name)
    z = encode(sd, cfg, torch.from_numpy(g["counts_subset"]), torch.from_numpy(g["genes_subset"]))
    assert max_abs_rel(z, g["z"]) < 5e-5
    mu, theta = decode(sd, cfg, torch.from_numpy(g["z"]), torch.from_numpy(g["genes"]), torch.from_numpy(g["library_size"]))
    assert max_abs_rel(mu, g["mu"]) < 5e-5 and max_abs_rel(theta, g["theta"]) < 1e-6
    mu2, _ = decode(sd, cfg, torch.from_numpy(g["zrand"]), torch.from_numpy(g["genes"]), torch.from_numpy(g["library_size"]))
    assert max_abs_rel(mu2, g["mu_rand"]) < 5e-5
    assert torch.allclose(mu.sum(1, keepdim=True), torch.from_numpy(g["library_size"]), rtol=1e-5)


def test_state_dict_keys_pin():
    g = load_golden("vae_small")
    shapes = golden_json(g, "shapes_json")
    for k in ("encoder.ca_layer.inducing_points", "encoder.pos_embed", "encoder.encoder_latent_input.0.weight",
              "decoder.decoder_latent_input.1.weight", "decoder.decoder_cross_attention.attn.c_attn_q.weight",
              "input_layer.gene_embedding.weight", "decoder_head.theta.weight", "decoder_head.params.bias"):
        assert k in shapes, k
    assert shapes["encoder.ca_layer.mlp.w1.weight"] == [88, 32]
