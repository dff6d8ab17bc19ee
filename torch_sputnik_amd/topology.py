"""CSR topology helpers the reference duplicates in every module
(modules/spmm.py:4-6, modules/sparse_linear.py:5-16,
modules/sparse_attention.py:12-36).  Pure torch, device-agnostic."""
import numpy as np
import torch


def diffsort(row_offsets):
    """``row_indices`` for a CSR matrix: rows ordered by ASCENDING length.

    Same order as the reference's ``argsort(offsets - roll(offsets, -1),
    descending=True)[:-1]`` (modules/spmm.py:4-6), always
    int32 (the reference forgets the cast in modules/sparse_linear.py:5-7).
    """
    lengths = row_offsets[1:] - row_offsets[:-1]
    return torch.argsort(lengths, stable=True).to(torch.int32)


def dense_to_sparse(matrix):
    """2-D dense tensor -> (values, row_indices, row_offsets, column_indices),
    int32 indices, as modules/sparse_linear.py:9-16 builds them."""
    csr = matrix.detach().to_sparse_csr()
    values = csr.values().clone()
    row_offsets = csr.crow_indices().to(torch.int32)
    column_indices = csr.col_indices().to(torch.int32)
    return values, diffsort(row_offsets), row_offsets, column_indices


def generate_mask(m, n, device="cpu", sparsity=0.9, round_to=4, generator=None):
    """Random 0/1 mask with the nonzero count of modules/sparse_attention.py:25-36:
    ``int(m*n*sparsity)`` zeros rounded DOWN to a multiple of ``round_to``.
    ``generator`` (numpy Generator) makes it reproducible; the reference uses
    the unseeded global numpy RNG."""
    num_elements = m * n
    remainder = int(num_elements * sparsity) % round_to
    num_zeros = int(num_elements * sparsity) - remainder
    num_ones = num_elements - num_zeros
    mask = np.zeros(num_elements, dtype=np.int64)
    mask[:num_ones] = 1
    (generator or np.random.default_rng()).shuffle(mask)
    return torch.from_numpy(mask).reshape(m, n).to(device)
