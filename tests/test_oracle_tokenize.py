"""Pin oracle/tokenize.py against outputs of the reference's own tokenize_cells (tests/golden/tok_*.npz).  CPU only."""
import numpy as np
import pytest

from conftest import load_golden
from oracle.tokenize import tokenize_expressed


@pytest.mark.parametrize("name", ["tok_small", "tok_dentate"])
def test_tokenize_expressed_matches_reference(name):
    g = load_golden(name)
    out = tokenize_expressed(g["counts"], g["gene_ids"], int(g["genes_seq_len"]), int(g["mask_idx"]))
    assert np.array_equal(out["genes_subset"], g["genes_subset"]) and out["genes_subset"].dtype == g["genes_subset"].dtype
    assert np.array_equal(out["counts_subset"], g["counts_subset"])
    assert np.array_equal(out["library_size"], g["library_size"])


def test_too_many_expressed_raises_like_reference():
    with pytest.raises(ValueError, match="genes_seq_len is smaller"):
        tokenize_expressed(np.ones((2, 9), np.float32), np.arange(9), 8)
