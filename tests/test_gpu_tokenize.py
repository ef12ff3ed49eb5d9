"""GPU parity of the encoder input tokenizer (scldm_tokenize_expressed through the C ABI): bit exact against the reference's
golden outputs and the oracle; full-size properties at the BASELINE gene counts."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle.tokenize import tokenize_expressed

pytestmark = pytest.mark.gpu


def run(counts, gene_ids, S, mask_idx=0, **kw):
    from scldm_amd.datamodule import tokenize_cells_expressed
    return tokenize_cells_expressed(torch.from_numpy(counts).cuda(), torch.from_numpy(gene_ids).cuda(), S, mask_idx, **kw)


@pytest.mark.parametrize("name", ["tok_small", "tok_dentate"])
def test_matches_reference_golden_bit_exact(name):
    g = load_golden(name)
    out = run(g["counts"], g["gene_ids"], int(g["genes_seq_len"]), int(g["mask_idx"]))
    assert torch.equal(out["genes_subset"].cpu(), torch.from_numpy(g["genes_subset"]))
    assert torch.equal(out["counts_subset"].cpu(), torch.from_numpy(g["counts_subset"]))
    assert torch.equal(out["library_size"].cpu(), torch.from_numpy(g["library_size"]))
    assert out["genes"].shape == g["counts"].shape and torch.equal(out["genes"][1].cpu(), torch.from_numpy(g["gene_ids"]))


@pytest.mark.parametrize("N,G,S", [(1, 1, 1), (3, 255, 80), (7, 256, 256), (5, 1025, 300), (4, 4099, 4099), (0, 10, 4)])
def test_ragged_shapes_vs_oracle(N, G, S):
    rng = np.random.default_rng(N * 1000 + G)
    counts = (rng.poisson(1.0, (N, G)) * (rng.random((N, G)) < 0.2)).astype(np.float32)
    gene_ids = rng.permutation(G).astype(np.int64) + 1
    ref = tokenize_expressed(counts, gene_ids, S, 0) if N else None
    out = run(counts, gene_ids, S)
    assert out["genes_subset"].shape == (N, S)
    if N:
        assert np.array_equal(out["genes_subset"].cpu().numpy(), ref["genes_subset"])
        assert np.array_equal(out["counts_subset"].cpu().numpy(), ref["counts_subset"])
        assert np.array_equal(out["num_expressed"].cpu().numpy(), ref["num_expressed"])
        assert np.array_equal(out["library_size"].cpu().numpy(), ref["library_size"])


def test_per_cell_gene_rows_and_overflow_error():
    rng = np.random.default_rng(3)
    counts = (rng.random((4, 300)) < 0.3).astype(np.float32)
    gid = np.stack([rng.permutation(300) for _ in range(4)]).astype(np.int64) + 5
    out = run(counts, gid, 200, 2)
    for i in range(4):
        ref = tokenize_expressed(counts[i:i + 1], gid[i], 200, 2)
        assert np.array_equal(out["genes_subset"][i].cpu().numpy(), ref["genes_subset"][0])
    with pytest.raises(ValueError, match="genes_seq_len is smaller"):     # datamodule.py:707-708
        run(np.ones((2, 64), np.float32), np.arange(64, dtype=np.int64), 63)
    out = run(np.ones((2, 64), np.float32), np.arange(64, dtype=np.int64) + 1, 63, check=False)
    assert torch.equal(out["genes_subset"][0].cpu(), torch.arange(1, 64)) and int(out["num_expressed"][0]) == 64


@pytest.mark.parametrize("G,S", [(17002, 6147), (27997, 10186)])     # dentate_gyrus / hlca gene counts (datamodule/default.yaml:48,64)
def test_full_size_properties(G, S):
    """At BASELINE sizes (256 cells): every expressed gene appears once, in gene order, followed only by mask tokens; counts
    travel with their genes; the tokenised batch encodes to the same latents as the oracle-tokenised one would (same arrays)."""
    N = 256
    gen = torch.Generator(device="cuda").manual_seed(G)
    counts = (torch.poisson(torch.full((N, G), 0.8, device="cuda"), generator=gen) *
              (torch.rand((N, G), device="cuda", generator=gen) < 0.25)).float()
    gene_idx = torch.randperm(G, device="cuda", generator=gen) + 1
    from scldm_amd.datamodule import tokenize_cells_expressed
    out = tokenize_cells_expressed(counts, gene_idx, S)
    ne = (counts > 0).sum(1)
    assert torch.equal(out["num_expressed"].long(), ne)
    pos = torch.arange(S, device="cuda").unsqueeze(0)
    valid = pos < ne.unsqueeze(1)
    assert bool((out["genes_subset"][~valid] == 0).all()) and bool((out["counts_subset"][~valid] == 0).all())
    assert bool((out["counts_subset"][valid] > 0).all())
    assert torch.equal(out["counts_subset"].sum(1, keepdim=True), out["library_size"])            # integer counts: exact
    assert torch.equal(out["library_size"], counts.sum(1, keepdim=True))
    # order preserving + exactly the expressed set: scatter back and compare with the dense matrix
    inv = torch.empty(G + 1, dtype=torch.long, device="cuda")
    inv[gene_idx] = torch.arange(G, device="cuda")
    cols = inv[out["genes_subset"].clamp(min=1)]
    assert bool(((cols[:, 1:] > cols[:, :-1]) | ~valid[:, 1:]).all())
    dense = torch.zeros_like(counts)
    dense.scatter_(1, torch.where(valid, cols, torch.zeros_like(cols)), torch.where(valid, out["counts_subset"], torch.zeros_like(out["counts_subset"])))
    dense[:, 0] = torch.where(counts[:, 0] > 0, counts[:, 0], torch.zeros_like(counts[:, 0]))   # slot 0 absorbed the padding writes
    assert torch.equal(dense, counts)


@pytest.mark.parametrize("N,G,density", [(1, 1, 1.0), (5, 257, 0.3), (3, 1024, 0.0), (64, 2000, 0.12), (256, 17002, 0.15)])
def test_dense_to_csr_matches_scipy(N, G, density):
    """Output assembly: arrays identical to scipy.sparse.csr_matrix(dense) (the reference's call, _utils.py:192-197)."""
    from scipy import sparse
    from scldm_amd.datamodule import dense_to_csr
    rng = np.random.default_rng(G)
    dense = (rng.poisson(2.0, (N, G)) * (rng.random((N, G)) < density)).astype(np.float32)
    ref = sparse.csr_matrix(dense)
    indptr, indices, data = dense_to_csr(torch.from_numpy(dense).cuda())
    assert np.array_equal(indptr.cpu().numpy(), ref.indptr.astype(np.int64))
    assert np.array_equal(indices.cpu().numpy(), ref.indices) and indices.dtype == torch.int32
    assert np.array_equal(data.cpu().numpy(), ref.data)
    back = sparse.csr_matrix((data.cpu().numpy(), indices.cpu().numpy(), indptr.cpu().numpy()), shape=(N, G)).toarray()
    assert np.array_equal(back, dense)


# ---- the other sample_genes modes (src/scldm/datamodule.py:697-800): random samplers, so invariants and distributions -----
def _toy(N=64, G=300, seed=0):
    g = torch.Generator().manual_seed(seed)
    counts = torch.poisson(torch.full((N, G), 0.4), generator=g)
    counts[0] = 0                                     # a cell with nothing expressed
    gene_idx = torch.arange(G) + 5
    return counts.cuda(), gene_idx.cuda()


def test_tokenize_mode_none_and_errors():
    from scldm_amd.datamodule import tokenize_cells
    counts, gene_idx = _toy()
    out = tokenize_cells(counts, gene_idx, 50, "none")
    assert torch.equal(out["counts"], counts) and torch.equal(out["genes"][3], gene_idx) and torch.equal(out["library_size"], counts.sum(1, keepdim=True))
    with pytest.raises(ValueError):
        tokenize_cells(counts, gene_idx, 50, "all")                        # the reference raises on unknown modes too (:802-803)
    with pytest.raises(ValueError):
        tokenize_cells(counts, gene_idx, 50, "weighted")                   # needs gene means (:698-699)
    with pytest.raises(ValueError):
        tokenize_cells(counts, gene_idx, 301, "random")
    ref = tokenize_cells(counts[:, :60].contiguous(), gene_idx[:60].contiguous(), 60, "expressed", mask_token_idx=1)
    assert set(ref) == {"genes", "counts", "genes_subset", "counts_subset", "library_size"}


@pytest.mark.parametrize("mode", ["random", "weighted"])
def test_tokenize_random_subsets(mode):
    from scldm_amd.datamodule import tokenize_cells
    counts, gene_idx = _toy(N=2000, G=40, seed=1)
    means = torch.linspace(0.2, 2.0, 40).cuda()
    gen = torch.Generator(device="cuda").manual_seed(3)
    out = tokenize_cells(counts, gene_idx, 10, mode, gene_means=means, generator=gen)
    g, c = out["genes"], out["counts"]
    assert g.shape == (2000, 10) and c.shape == (2000, 10)
    assert all(len(set(row.tolist())) == 10 for row in g[:50])             # without replacement
    assert torch.equal(c, counts.gather(1, g - 5))                          # counts follow their genes
    freq = torch.bincount((g - 5).flatten(), minlength=40).double() / (2000 * 10)
    if mode == "random":
        assert float((freq - 1 / 40).abs().max()) < 0.006                  # uniform inclusion
    else:
        w = ((counts + 1) / means.view(1, -1)).double()
        first = w / w.sum(1, keepdim=True)                                  # exact probabilities of the FIRST draw
        got_first = torch.bincount((g[:, 0] - 5), minlength=40).double() / 2000
        assert float((got_first - first.mean(0)).abs().max()) < 0.02
        assert float(freq[:5].mean()) > float(freq[-5:].mean())           # rare-mean genes are favoured


def test_tokenize_expressed_zero_and_random_expressed():
    from scldm_amd.datamodule import tokenize_cells
    counts, gene_idx = _toy(N=256, G=300, seed=2)
    n_zero = (counts == 0).sum(1)
    out = tokenize_cells(counts, gene_idx, 200, "expressed_zero")
    cs = out["counts_subset"]
    for i in (0, 1, 17, 255):
        z = int(min(n_zero[i], 200))
        assert bool((cs[i, :z] == 0).all()) and bool((cs[i, z:] > 0).all())     # the non-expressed genes come first (:745-747)
        assert torch.equal(cs[i], counts[i].gather(0, out["genes_subset"][i] - 5))
    out2 = tokenize_cells(counts, gene_idx, 150, "random_expressed", mask_token_idx=1)
    n_exp = (counts > 0).sum(1)
    for i in (0, 1, 17, 255):
        k = int(min(n_exp[i], 150))
        assert bool((out2["counts"][i, :k] > 0).all()) and bool((out2["counts"][i, k:] == 0).all())
        assert bool((out2["genes"][i, k:] == 1).all())
        assert len(set(out2["genes"][i, :k].tolist())) == k
    # genes_seq_len > G: the reference pads with the mask token (np.pad, datamodule.py:759-768) - so does this
    out3 = tokenize_cells(counts, gene_idx, 340, "random_expressed", mask_token_idx=1)
    assert out3["genes"].shape == (256, 340) and out3["counts"].shape == (256, 340)
    for i in (0, 17, 255):
        k = int(n_exp[i])
        assert bool((out3["counts"][i, :k] > 0).all()) and bool((out3["counts"][i, k:] == 0).all()) and bool((out3["genes"][i, k:] == 1).all())
        assert sorted(out3["genes"][i, :k].tolist()) == gene_idx[counts[i] > 0].tolist()     # every expressed gene, once
