"""The five operators of the reference's ``torch_sputnik`` module
(src/sputnik.cpp:36-42), same names, positional signatures and return types,
backed by ``torch.ops.torch_sputnik.*`` (HIP kernels for gfx950).

GPU only: a CPU tensor raises from the dispatcher; there is no fallback.
"""
from ._native import load_ops

_ops = load_ops()


def spmm(m, k, values, row_indices, row_offsets, column_indices, dense_matrix):
    """Sparse (CSR) x dense.  src/spmm_cuda.cu:9-60.

    values [nnz] with dense [k,n] -> [m,n]; values [R,nnz] with dense [R,k,n]
    -> [R,m,n] (2-D when R == 1, as the reference does).
    """
    return _ops.spmm(int(m), int(k), values, row_indices, row_offsets, column_indices,
                     dense_matrix)


def left_spmm(m, k, values, row_indices, row_offsets, column_indices, dense_matrix):
    """One sparse matrix x R dense matrices -> always [R,m,n].
    src/left_replicated_spmm.cu:8-44."""
    return _ops.left_spmm(int(m), int(k), values, row_indices, row_offsets, column_indices,
                          dense_matrix)


# BASELINE.json names this op left_replicated_spmm; the binding exports left_spmm.
left_replicated_spmm = left_spmm


def sddmm(m, n, row_indices, row_offsets, column_indices, lhs_matrix, rhs_matrix):
    """(lhs @ rhs^T) sampled at the CSR pattern -> [nnz] / [R,nnz].
    src/sddmm_cuda.cu:7-57."""
    return _ops.sddmm(int(m), int(n), row_indices, row_offsets, column_indices, lhs_matrix,
                      rhs_matrix)


def sparse_softmax(values, row_indices, row_offsets, column_indices):
    """Row-wise softmax over the stored entries.  src/softmax_cuda.cu:7-46."""
    return _ops.sparse_softmax(values, row_indices, row_offsets, column_indices)


def csr_transpose(m, n, values, row_offsets, column_indices):
    """CSR(m x n) -> [values_t, row_offsets_t, column_indices_t] of the transpose.
    src/transpose_cuda.cu:45-102.  values may also be [R,nnz] (extension)."""
    return _ops.csr_transpose(int(m), int(n), values, row_offsets, column_indices)


def csr_transpose_with_permutation(m, n, values, row_offsets, column_indices):
    """Extension: as csr_transpose plus a 4th tensor,
    ``permutation`` with values_t == values[..., permutation], so a static
    topology's transpose can be cached by the caller."""
    return _ops.csr_transpose_with_permutation(int(m), int(n), values, row_offsets,
                                               column_indices)
