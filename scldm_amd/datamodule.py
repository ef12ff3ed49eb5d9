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


def to_host(*tensors: torch.Tensor) -> tuple[torch.Tensor, ...]:
    """Device tensors -> host tensors through PINNED staging memory: every copy is queued asynchronously on the current stream and the
    host waits ONCE (the reference's `tree_map(lambda x: x.cpu(), batch)`, src/scldm/models.py:742, is one synchronous copy into
    pageable memory per tensor: ~19 GB/s on this host against ~45 GB/s pinned).  torch's pinned-memory allocator caches the blocks, so
    only the first call pays for pinning.  The returned tensors are ordinary CPU tensors (pinned storage)."""
    outs = []
    for t in tensors:
        if not t.is_cuda:
            outs.append(t)
            continue
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        h.copy_(t, non_blocking=True)
        outs.append(h)
    if any(t.is_cuda for t in tensors):
        torch.cuda.current_stream().synchronize()
    return tuple(outs)


def tokenize_cells(counts: torch.Tensor, gene_idx: torch.Tensor, genes_seq_len: int, sample_genes: str, mask_token_idx: int = 0,
                   gene_means: torch.Tensor | None = None, generator: torch.Generator | None = None) -> dict[str, torch.Tensor]:
    """Device-side `scldm.datamodule.tokenize_cells` for every `sample_genes` mode (src/scldm/datamodule.py:652-805); dict keys as
    in the reference ("genes", "counts", "library_size", and "genes_subset" / "counts_subset" where it has them).

      "expressed"        the HIP stream-compaction kernel above (deterministic: bit-exact with the reference)
      "none"             pass-through + library size (:795-800)
      "random"           `genes_seq_len` genes per cell, uniformly without replacement (:787-793)
      "weighted"         without replacement with probability ~ (count + 1) / gene mean (:697-709); needs `gene_means` (G,)
      "expressed_zero"   a random permutation stably sorted by the expressed flag, first `genes_seq_len` kept - the reference's
                         argsort puts the NON-expressed genes first (:733-752), and so does this
      "random_expressed" up to `genes_seq_len` expressed genes uniformly without replacement, padded with the mask token (:754-785)

    The random modes draw with torch's device generator instead of the reference's per-batch `np.random.default_rng(seed)`: the
    same distributions (uniform / Plackett-Luce subsets via random keys and the Gumbel top-k identity), not the same streams -
    parity for them is distributional by construction.  CUDA (ROCm) tensors only."""
    if sample_genes == "expressed":
        out = tokenize_cells_expressed(counts, gene_idx, genes_seq_len, mask_token_idx)
        out.pop("num_expressed")
        return out
    if counts.device.type != "cuda" or gene_idx.device != counts.device:
        raise RuntimeError("tokenize_cells works on CUDA (ROCm) tensors; there is no CPU path")
    if counts.dim() != 2:
        raise ValueError("counts must be (N, G)")
    N, G = counts.shape
    S = int(genes_seq_len)
    gene_idx = gene_idx.to(torch.long)
    genes = gene_idx if gene_idx.shape == (N, G) else gene_idx.unsqueeze(0).expand(N, G)
    lib = counts.sum(1, keepdim=True)
    rand = lambda: torch.rand((N, G), device=counts.device, generator=generator)
    if sample_genes == "none":
        return {"genes": genes, "counts": counts, "library_size": lib}
    if sample_genes in ("random", "weighted"):
        if S > G:
            raise ValueError("Cannot take a larger sample than population when 'replace=False'")
        if sample_genes == "random":
            keys = rand()
        else:
            if gene_means is None:
                raise ValueError("encoder.metadata_genes must be set for weighted sampling")
            p = (counts + 1) / gene_means.to(counts).view(1, G)
            keys = torch.log(p) - torch.log(-torch.log(rand().clamp_min(1e-20)))   # Gumbel top-k = successive sampling without replacement
            keys = -keys
        idx = keys.argsort(dim=1)[:, :S]
        return {"genes": genes.gather(1, idx), "counts": counts.gather(1, idx), "library_size": lib}
    expressed = counts > 0
    if sample_genes == "expressed_zero":
        order = (expressed.to(counts.dtype) + rand()).argsort(dim=1)[:, :S]          # zeros (flag 0) first, each group in random order
        return {"genes": genes, "counts": counts, "genes_subset": genes.gather(1, order), "counts_subset": counts.gather(1, order),
                "library_size": lib}
    if sample_genes == "random_expressed":
        order = ((~expressed).to(counts.dtype) * 2 + rand()).argsort(dim=1)[:, :S]   # expressed genes first, random order
        if S > G:   # the reference accepts genes_seq_len > G and pads with the mask token (np.pad, :759-768): pad the gather index
            order = torch.cat([order, order.new_zeros((N, S - G))], dim=1)
        take = torch.arange(S, device=counts.device).unsqueeze(0) < expressed.sum(1, keepdim=True)
        g = torch.where(take, genes.gather(1, order), torch.full_like(order, int(mask_token_idx)))
        c = torch.where(take, counts.gather(1, order), torch.zeros((), dtype=counts.dtype, device=counts.device))
        return {"genes": g, "counts": c, "library_size": lib}
    raise ValueError(f"Invalid sample_genes value: {sample_genes}")
