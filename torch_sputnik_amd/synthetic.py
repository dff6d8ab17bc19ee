"""Synthetic CSR inputs with the distribution of the reference's test helpers,
generated on the device (used by bench.py, smoke() and the full-size tests).

``random_csr`` restates tests/connectors.py:34-59 (``Uniform(sparsity,
round_to)``: the nonzero positions are a uniform sample without replacement,
their count rounded UP to a multiple of ``round_to``) followed by
tests/sparse_matrix.py:9-41 (row-major CSR, ascending columns,
``row_indices = argsort(-row_length)``).  The reference draws from the
unseeded numpy global RNG; here the seed is explicit.
"""
import torch


def nonzero_count(m, n, density, round_to=4):
    size = m * n
    num_dormant = int(round((1.0 - density) * size))
    nnz = size - num_dormant
    return (nnz + round_to - 1) // round_to * round_to


def random_csr(m, n, density, device, seed=0, round_to=4, order="descending"):
    """-> (row_indices, row_offsets, column_indices, nnz), int32 on `device`."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    size = m * n
    nnz = nonzero_count(m, n, density, round_to)
    idx = torch.sort(torch.randperm(size, device=device, generator=g)[:nnz]).values
    rows = torch.div(idx, n, rounding_mode="floor")
    cols = (idx - rows * n).to(torch.int32).contiguous()
    counts = torch.bincount(rows, minlength=m)
    row_offsets = torch.zeros(m + 1, dtype=torch.int64, device=device)
    row_offsets[1:] = torch.cumsum(counts, 0)
    row_indices = torch.argsort(counts, descending=(order == "descending"), stable=True)
    return row_indices.to(torch.int32), row_offsets.to(torch.int32), cols, nnz


def uniform(shape, device, seed):
    """U[0,1) float32, as tests/initializers.py:24-31."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return torch.rand(shape, device=device, generator=g, dtype=torch.float32)
