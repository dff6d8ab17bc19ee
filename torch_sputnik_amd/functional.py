"""autograd.Functions over the five ops, mirroring the reference's Python
layer: ``Spmm`` (modules/spmm.py:8-74), ``Sddmm`` (modules/sddmm.py:9-74),
``SparseLinearFunction`` (modules/sparse_linear.py:18-67).  Same ``apply``
signatures, same gradients returned in the same positions.

Beyond the reference:
  * 3-D (batched) backward works, because csr_transpose here accepts [R,nnz]
    values (the reference's is 1-D only, src/transpose_cuda.cu:50);
  * ``SparseSoftmax`` gives sparse_softmax a backward (the reference calls the
    raw op, modules/sparse_attention.py:76, which cuts the gradient);
  * the transposed topology of a static mask can be cached
    (``TransposeCache``), so a backward is one gather instead of a transpose.
"""
import torch

from . import ops
from .topology import diffsort


class TransposeCache:
    """Memoises the transposed topology (and the value permutation) of static
    CSR patterns, keyed by the identity and version of the index tensors."""

    def __init__(self):
        self._entries = {}

    @staticmethod
    def _key(m, n, row_offsets, column_indices):
        return (m, n, row_offsets.data_ptr(), column_indices.data_ptr(), row_offsets._version,
                column_indices._version, column_indices.numel(), str(column_indices.device))

    def lookup(self, m, n, row_offsets, column_indices, probe_values):
        key = self._key(m, n, row_offsets, column_indices)
        entry = self._entries.get(key)
        if entry is None:
            _, row_offsets_t, column_indices_t, permutation = ops.csr_transpose_with_permutation(
                m, n, probe_values.detach().reshape(-1, probe_values.shape[-1])[0].contiguous(),
                row_offsets, column_indices)
            entry = (diffsort(row_offsets_t), row_offsets_t, column_indices_t,
                     permutation.to(torch.int64), row_offsets, column_indices)
            self._entries[key] = entry
        return entry[:4]

    def clear(self):
        self._entries.clear()


_cache = None


def enable_transpose_cache(enabled=True):
    """Opt in to caching transposed topologies across backward calls."""
    global _cache
    _cache = TransposeCache() if enabled else None
    return _cache


def _transpose(m, n, values, row_offsets, column_indices):
    """(values_t, row_indices_t, row_offsets_t, column_indices_t)."""
    values = values.contiguous()
    if _cache is not None:
        row_indices_t, row_offsets_t, column_indices_t, perm = _cache.lookup(
            m, n, row_offsets, column_indices, values)
        return values.index_select(-1, perm), row_indices_t, row_offsets_t, column_indices_t
    values_t, row_offsets_t, column_indices_t = ops.csr_transpose(
        m, n, values, row_offsets, column_indices)
    return values_t, diffsort(row_offsets_t), row_offsets_t, column_indices_t


class Spmm(torch.autograd.Function):
    """sparse(values, CSR topology) @ dense.  ``apply(m, k, values, row_indices,
    row_offsets, column_indices, dense)``."""

    @staticmethod
    def forward(ctx, m, k, values, row_indices, row_offsets, column_indices, dense):
        ctx.shape = (m, k)
        ctx.topology = (row_indices, row_offsets, column_indices)
        ctx.save_for_backward(values, dense)
        return ops.spmm(m, k, values, row_indices, row_offsets, column_indices, dense)

    @staticmethod
    def backward(ctx, grad_output):
        m, k = ctx.shape
        row_indices, row_offsets, column_indices = ctx.topology
        values, dense = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        grad_values = grad_dense = None
        if ctx.needs_input_grad[2]:
            # dL/dA sampled at the pattern: <dC[i,:], B[j,:]>
            grad_values = ops.sddmm(m, k, row_indices, row_offsets, column_indices, grad_output,
                                    dense.contiguous())
        if ctx.needs_input_grad[6]:
            # dL/dB = A^T @ dC
            values_t, row_indices_t, row_offsets_t, column_indices_t = _transpose(
                m, k, values, row_offsets, column_indices)
            grad_dense = ops.spmm(k, m, values_t, row_indices_t, row_offsets_t, column_indices_t,
                                  grad_output)
        return None, None, grad_values, None, None, None, grad_dense


class Sddmm(torch.autograd.Function):
    """(lhs @ rhs^T) sampled at a CSR mask.  ``apply(m, n, row_indices,
    row_offsets, column_indices, lhs_matrix, rhs_matrix)``."""

    @staticmethod
    def forward(ctx, m, n, row_indices, row_offsets, column_indices, lhs_matrix, rhs_matrix):
        ctx.shape = (m, n)
        ctx.topology = (row_indices, row_offsets, column_indices)
        ctx.save_for_backward(lhs_matrix, rhs_matrix)
        return ops.sddmm(m, n, row_indices, row_offsets, column_indices, lhs_matrix, rhs_matrix)

    @staticmethod
    def backward(ctx, grad_output):
        m, n = ctx.shape
        row_indices, row_offsets, column_indices = ctx.topology
        lhs_matrix, rhs_matrix = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        grad_lhs = grad_rhs = None
        if ctx.needs_input_grad[5]:
            # dL/dlhs = dS @ rhs, dS sparse with the mask's pattern
            grad_lhs = ops.spmm(m, n, grad_output, row_indices, row_offsets, column_indices,
                                rhs_matrix.contiguous())
        if ctx.needs_input_grad[6]:
            # dL/drhs = dS^T @ lhs
            grad_t, row_indices_t, row_offsets_t, column_indices_t = _transpose(
                m, n, grad_output, row_offsets, column_indices)
            grad_rhs = ops.spmm(n, m, grad_t, row_indices_t, row_offsets_t, column_indices_t,
                                lhs_matrix.contiguous())
        return None, None, None, None, None, grad_lhs, grad_rhs


class SparseLinearFunction(torch.autograd.Function):
    """One sparse weight x a batch of dense matrices (left_spmm).  ``apply(m, k,
    values, row_indices, row_offsets, column_indices, dense)`` with dense
    [B,k,n] -> [B,m,n]."""

    @staticmethod
    def forward(ctx, m, k, values, row_indices, row_offsets, column_indices, dense):
        ctx.shape = (m, k)
        ctx.topology = (row_indices, row_offsets, column_indices)
        ctx.save_for_backward(values, dense)
        return ops.left_spmm(m, k, values, row_indices, row_offsets, column_indices, dense)

    @staticmethod
    def backward(ctx, grad_output):
        m, k = ctx.shape
        row_indices, row_offsets, column_indices = ctx.topology
        values, dense = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        grad_values = grad_dense = None
        if ctx.needs_input_grad[2]:
            # [B,nnz]; autograd sums it over B to match the shared `values`.
            grad_values = ops.sddmm(m, k, row_indices, row_offsets, column_indices, grad_output,
                                    dense.contiguous())
            if grad_values.dim() == 2:
                grad_values = grad_values.sum(dim=0)
        if ctx.needs_input_grad[6]:
            values_t, row_indices_t, row_offsets_t, column_indices_t = _transpose(
                m, k, values, row_offsets, column_indices)
            grad_dense = ops.left_spmm(k, m, values_t, row_indices_t, row_offsets_t,
                                       column_indices_t, grad_output)
            if dense.dim() == 2:
                grad_dense = grad_dense[0]
        return None, None, grad_values, None, None, None, grad_dense


class SparseSoftmax(torch.autograd.Function):
    """sparse_softmax with a gradient: dX = scale * Y * (dY - rowsum(dY * Y)), the
    row sums taken over the stored entries (extension).
    ``apply(values, row_indices, row_offsets, column_indices[, scale])`` computes
    softmax(scale * values)."""

    @staticmethod
    def forward(ctx, values, row_indices, row_offsets, column_indices, scale=1.0):
        if scale == 1.0:
            out = ops.sparse_softmax(values, row_indices, row_offsets, column_indices)
        else:
            out = ops.sparse_softmax_scaled(values, row_indices, row_offsets, column_indices,
                                            scale)
        ctx.scale = float(scale)
        ctx.save_for_backward(out, row_offsets)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        out, row_offsets = ctx.saved_tensors
        grad_values = ops.sparse_softmax_backward(out, grad_output.contiguous(), row_offsets,
                                                  ctx.scale)
        return grad_values, None, None, None, None


class SparseAttentionFunction(torch.autograd.Function):
    """softmax(scale * sddmm(q, k)) @ v with the ONE-kernel forward
    (ops.sparse_attention) and a backward built from the separate operators.
    Nothing of size [R, nnz] is kept between the passes: the backward recomputes
    the weights (sddmm + scaled softmax) and then runs the standard chain

        dV = P^T dO          dP = sddmm(dO, v)
        dS = softmax'(P, dP) (sparse_softmax_backward, carries the scale)
        dQ = dS k            dK = dS^T q

    on the mask and its transpose (one csr_transpose with permutation per call,
    or the cached transposed topology)."""

    @staticmethod
    def forward(ctx, query, key, value, row_indices, row_offsets, column_indices, scale):
        out = ops.sparse_attention(query, key, value, row_indices, row_offsets, column_indices,
                                   scale)
        ctx.scale = float(scale)
        ctx.save_for_backward(query, key, value, row_indices, row_offsets, column_indices)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        query, key, value, row_indices, row_offsets, column_indices = ctx.saved_tensors
        topo = (row_indices, row_offsets, column_indices)
        m, n = query.size(-2), key.size(-2)
        grad_output = grad_output.contiguous()
        scores = ops.sddmm(m, n, *topo, query, key)
        weights = ops.sparse_softmax_scaled(scores, *topo, ctx.scale)
        grad_weights = ops.sddmm(m, n, *topo, grad_output, value)
        grad_scores = ops.sparse_softmax_backward(weights, grad_weights, row_offsets, ctx.scale)
        grad_query = grad_key = grad_value = None
        if ctx.needs_input_grad[0]:
            grad_query = ops.spmm(m, n, grad_scores, *topo, key)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            grad_scores_t, row_offsets_t, column_indices_t, perm = \
                ops.csr_transpose_with_permutation(m, n, grad_scores, row_offsets, column_indices)
            row_indices_t = diffsort(row_offsets_t)
            if ctx.needs_input_grad[1]:
                grad_key = ops.spmm(n, m, grad_scores_t, row_indices_t, row_offsets_t,
                                    column_indices_t, query)
            if ctx.needs_input_grad[2]:
                weights_t = weights.index_select(-1, perm.to(torch.int64))
                grad_value = ops.spmm(n, m, weights_t, row_indices_t, row_offsets_t,
                                      column_indices_t, grad_output)
        return grad_query, grad_key, grad_value, None, None, None, None


# ---------------------------------------------------------------------------
# many-mask family: one mask per batch element, shared by its heads.  Same
# ``apply`` signatures and gradient positions as the reference's sketches in
# tests/transformer/functions.py (Spmm :5-69, CsrSoftmax :70-120, Sddmm :122-188).
# ---------------------------------------------------------------------------
def diffsort_many_mask(row_offsets, masks):
    """Per-mask ``diffsort`` of stacked / concatenated offsets
    (tests/transformer/utils.py:51-62) -> flat [masks * rows]."""
    per_mask = row_offsets.reshape(masks, -1)
    return torch.cat([diffsort(per_mask[i]) for i in range(masks)])


class SpmmManyMask(torch.autograd.Function):
    """tests/transformer/functions.py:5-69."""

    @staticmethod
    def forward(ctx, b, m, k, nonzeros, values, row_indices, row_offsets, column_indices, dense):
        ctx.dims = (b, m, k)
        ctx.nonzeros = nonzeros
        ctx.save_for_backward(values, row_indices, row_offsets, column_indices, dense)
        return ops.spmm_many_mask(b, m, k, nonzeros, values, row_indices, row_offsets,
                                  column_indices, dense)

    @staticmethod
    def backward(ctx, grad_output):
        b, m, k = ctx.dims
        nonzeros = ctx.nonzeros
        values, row_indices, row_offsets, column_indices, dense = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        grad_values = grad_dense = None
        if ctx.needs_input_grad[4]:
            grad_values = ops.sddmm_many_mask(b, m, k, nonzeros, row_indices, row_offsets,
                                              column_indices, grad_output, dense)
            if grad_values.shape[-1] != values.shape[-1]:  # values rows were padded
                grad_values = torch.nn.functional.pad(
                    grad_values, (0, values.shape[-1] - grad_values.shape[-1]))
        if ctx.needs_input_grad[8]:
            values_t, row_offsets_t, column_indices_t = ops.csr_transpose_many_mask(
                b, m, k, nonzeros, values.detach(), row_offsets, column_indices)
            row_indices_t = diffsort_many_mask(row_offsets_t, b)
            grad_dense = ops.spmm_many_mask(b, k, m, nonzeros, values_t, row_indices_t,
                                            row_offsets_t, column_indices_t, grad_output)
        return None, None, None, None, grad_values, None, None, None, grad_dense


class SddmmManyMask(torch.autograd.Function):
    """tests/transformer/functions.py:122-188."""

    @staticmethod
    def forward(ctx, b, m, n, nonzeros, row_indices, row_offsets, column_indices, lhs_matrix,
                rhs_matrix):
        ctx.dims = (b, m, n)
        ctx.nonzeros = nonzeros
        ctx.save_for_backward(row_indices, row_offsets, column_indices, lhs_matrix, rhs_matrix)
        return ops.sddmm_many_mask(b, m, n, nonzeros, row_indices, row_offsets, column_indices,
                                   lhs_matrix, rhs_matrix)

    @staticmethod
    def backward(ctx, grad_output):
        b, m, n = ctx.dims
        nonzeros = ctx.nonzeros
        row_indices, row_offsets, column_indices, lhs_matrix, rhs_matrix = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        grad_lhs = grad_rhs = None
        if ctx.needs_input_grad[7]:
            grad_lhs = ops.spmm_many_mask(b, m, n, nonzeros, grad_output, row_indices,
                                          row_offsets, column_indices, rhs_matrix)
        if ctx.needs_input_grad[8]:
            grad_t, row_offsets_t, column_indices_t = ops.csr_transpose_many_mask(
                b, m, n, nonzeros, grad_output, row_offsets, column_indices)
            row_indices_t = diffsort_many_mask(row_offsets_t, b)
            grad_rhs = ops.spmm_many_mask(b, n, m, nonzeros, grad_t, row_indices_t,
                                          row_offsets_t, column_indices_t, lhs_matrix)
        return None, None, None, None, None, None, None, grad_lhs, grad_rhs


class CsrSoftmaxManyMask(torch.autograd.Function):
    """tests/transformer/functions.py:70-120 with the softmax Jacobian in the
    backward (the sketch there returns ``s * (1 - s)`` of a dense softmax of the
    incoming gradient).  Optional trailing ``scale``."""

    @staticmethod
    def forward(ctx, b, m, nonzeros, scores, row_indices, row_offsets, column_indices,
                scale=1.0):
        out = ops.sparse_softmax_many_mask(b, m, nonzeros, scores, row_indices, row_offsets,
                                           column_indices, None if scale == 1.0 else scale)
        ctx.dims = (b, m, float(scale))
        ctx.nonzeros = nonzeros
        ctx.save_for_backward(out, row_offsets)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        b, m, scale = ctx.dims
        out, row_offsets = ctx.saved_tensors
        grad_scores = ops.sparse_softmax_backward_many_mask(
            b, m, ctx.nonzeros, out, grad_output.contiguous(), row_offsets, scale)
        return None, None, None, grad_scores, None, None, None, None
