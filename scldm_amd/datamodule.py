"""Device-side counterpart of the reference's batch tokenizer for the encoder input
(`scldm.datamodule.tokenize_cells(sample_genes="expressed")`, src/scldm/datamodule.py:660-731).

The reference runs this per batch in NumPy on dense (N, G) matrices inside the DataLoader workers; here the dense counts
are tokenised where they already live (HBM) by one kernel (scldm_tokenize_expressed), and the result feeds
`TransformerVAE.encode(counts, genes, counts_subset, genes_subset)` directly.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def tokenize_cells_expressed(counts: torch.Tensor, gene_idx: torch.Tensor, genes_seq_len: int, mask_token_idx: int = 0,
                             check: bool = True) -> dict[str, torch.Tensor]:
    """counts (N,G) fp32 CUDA, gene_idx (G,) or (N,G) int64 CUDA -> the reference's batch dict entries
    {"genes", "counts", "genes_subset", "counts_subset", "library_size"} (datamodule.py:719-725) plus "num_expressed".
    With check=True a cell with more expressed genes than genes_seq_len raises ValueError like the reference (:707-708);
    that check is the only host synchronisation."""
    if counts.device.type != "cuda" or gene_idx.device != counts.device:
        raise RuntimeError("tokenize_cells_expressed works on CUDA (ROCm) tensors; there is no CPU path")
    if counts.dim() != 2 or counts.dtype != torch.float32:
        raise ValueError("counts must be (N, G) float32")
    N, G = counts.shape
    counts = counts.contiguous()
    gene_idx = gene_idx.to(torch.long).contiguous()
    if gene_idx.shape == (G,):
        stride = 0
    elif gene_idx.shape == (N, G):
        stride = G
    else:
        raise ValueError(f"gene_idx must be ({G},) or ({N},{G}), got {tuple(gene_idx.shape)}")
    S = int(genes_seq_len)
    genes_out = torch.empty((N, S), dtype=torch.long, device=counts.device)
    counts_out = torch.empty((N, S), dtype=torch.float32, device=counts.device)
    nexp = torch.empty((N,), dtype=torch.int32, device=counts.device)
    lib = torch.empty((N, 1), dtype=torch.float32, device=counts.device)
    L = _lib.lib()
    with torch.cuda.device(counts.device):
        if N:   # an empty batch has no buffers to hand over
            _lib.check(L.scldm_tokenize_expressed(counts.data_ptr(), gene_idx.data_ptr(), stride, N, G, S, int(mask_token_idx),
                                                  genes_out.data_ptr(), counts_out.data_ptr(), nexp.data_ptr(), lib.data_ptr(),
                                                  torch.cuda.current_stream().cuda_stream), "scldm_tokenize_expressed")
    if check and N and bool((nexp > S).any()):
        raise ValueError("genes_seq_len is smaller than number of expressed genes")
    return {"genes": gene_idx if stride else gene_idx.unsqueeze(0).expand(N, G), "counts": counts, "genes_subset": genes_out,
            "counts_subset": counts_out, "library_size": lib, "num_expressed": nexp}


def dense_to_csr(dense: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(N,G) fp32 CUDA -> (indptr (N+1) int64, indices (nnz) int32, data (nnz) fp32) on device: the arrays of
    `scipy.sparse.csr_matrix(dense.cpu().numpy())` (reference _utils.py:192-197), so only nnz entries cross PCIe.
    One host synchronisation (nnz sizes the outputs)."""
    if dense.device.type != "cuda" or dense.dtype != torch.float32 or dense.dim() != 2:
        raise ValueError("dense must be a 2-D float32 CUDA tensor; there is no CPU path")
    N, G = dense.shape
    dense = dense.contiguous()
    indptr = torch.zeros(N + 1, dtype=torch.long, device=dense.device)
    if N == 0 or G == 0:
        return indptr, torch.empty(0, dtype=torch.int32, device=dense.device), torch.empty(0, dtype=torch.float32, device=dense.device)
    L = _lib.lib()
    nnz_row = torch.empty(N, dtype=torch.int32, device=dense.device)
    st = torch.cuda.current_stream().cuda_stream
    with torch.cuda.device(dense.device):
        _lib.check(L.scldm_csr_count(dense.data_ptr(), N, G, nnz_row.data_ptr(), st), "scldm_csr_count")
        indptr[1:] = torch.cumsum(nnz_row, 0, dtype=torch.long)
        nnz = int(indptr[-1])
        indices = torch.empty(max(nnz, 1), dtype=torch.int32, device=dense.device)
        data = torch.empty(max(nnz, 1), dtype=torch.float32, device=dense.device)
        _lib.check(L.scldm_csr_fill(dense.data_ptr(), N, G, indptr.data_ptr(), indices.data_ptr(), data.data_ptr(), st), "scldm_csr_fill")
    return indptr, indices[:nnz], data[:nnz]
