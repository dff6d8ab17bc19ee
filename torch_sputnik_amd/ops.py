"""The five operators of the reference's ``torch_sputnik`` module
(src/sputnik.cpp:36-42), same names, positional signatures and return types,
backed by ``torch.ops.torch_sputnik.*`` (HIP kernels for gfx950), followed by
the extensions whose call sites exist only in the reference's tests
(``spmm_bias``, the ``*_many_mask`` family) or not at all (softmax gradient).

GPU only: a CPU tensor raises from the dispatcher; there is no fallback.
"""
import torch

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


def left_spmm_half_tiles(m, k, values, row_offsets, column_indices, dense_matrix, tile_dtype):
    """left_spmm as a dense contraction on the matrix cores (the half-storage extension,
    csrc/spmm_mfma.hip): the densified weight against dense [R, k, n] on tiles of
    ``tile_dtype`` (float16 / bfloat16); `values` and `dense_matrix` float32 or of that
    type -- a float32 operand enters as half planes, not rounded.  -> [R, m, n] float32,
    or None where the route does not serve the call (take left_spmm then)."""
    import torch
    code = {torch.float16: 1, torch.bfloat16: 2}[tile_dtype]
    out = _ops.left_spmm_half_tiles(int(m), int(k), values, row_offsets, column_indices,
                                    dense_matrix, code)
    return out if out.numel() else None


# ---- a sparse layer on half-stored activations: the three products on the matrix cores with
# no layout pass (csrc/sparse_linear_half.hip; include/sputnik_hip.h: sparse_linear_half_*) ----
_HALF_CODES = None


def _half_code(dtype):
    global _HALF_CODES
    if _HALF_CODES is None:
        import torch
        _HALF_CODES = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
    return _HALF_CODES[dtype]


def half_linear_supported(out_features, in_features, seq, batch, nonzeros, values_dtype, tile_dtype):
    """Whether forward, weight gradient and input gradient of a sparse layer with these
    sizes take the matrix-core route (they share the weight's image and dy's planes)."""
    from . import capi
    try:
        vt, tt = _half_code(values_dtype), _half_code(tile_dtype)
    except KeyError:
        return False
    if tt == 0:
        return False
    return bool(capi.lib().sputnik_hip_sparse_linear_half_supported(
        int(out_features), int(in_features), int(seq), int(batch), int(nonzeros), vt, tt))


def half_linear_image(out_features, in_features, values, row_offsets, column_indices, tile_dtype):
    """The weight as a zeroed [planes, out, in] image of `tile_dtype` with the CSR values
    scattered in (float32 values: half planes whose sum is the value)."""
    return _ops.half_linear_image(int(out_features), int(in_features), values, row_offsets,
                                  column_indices, _half_code(tile_dtype))


def half_planes(t, tile_dtype):
    """A float32 tensor as half planes whose (scaled) sum is the value."""
    return _ops.half_planes(t, _half_code(tile_dtype))


def half_linear_forward(out_features, image, values_dtype, x):
    """y [batch, out, seq] float32 from x [batch, seq, in] (float16 / bfloat16), no layout pass."""
    return _ops.half_linear_forward(int(out_features), image, _half_code(values_dtype), x)


def half_linear_plan(out_features, in_features, row_offsets, column_indices):
    return _ops.half_linear_plan(int(out_features), int(in_features), row_offsets, column_indices)


def half_linear_weight_gradient(out_features, row_offsets, column_indices, grad, grad_is_planes, x,
                                plan=None):
    """dW [nnz] float32 = sum over the batch of dy x, sampled at the weight's mask; `grad` is
    dy [batch, out, seq] in x's type, or the planes of the float32 dy (`half_planes`)."""
    return _ops.half_linear_weight_gradient(int(out_features), row_offsets, column_indices, grad,
                                            bool(grad_is_planes), x, plan)


def half_linear_input_gradient(out_features, in_features, grad, grad_is_planes, image, values_dtype,
                               like, batch, seq):
    """dx [batch, seq, in] in `like`'s type, or None where the route does not serve the call."""
    out = _ops.half_linear_input_gradient(int(out_features), int(in_features), grad, bool(grad_is_planes),
                                          image, _half_code(values_dtype), like, int(batch), int(seq))
    return out if out.numel() else None


# BASELINE.json names this op left_replicated_spmm; the binding exports left_spmm.
left_replicated_spmm = left_spmm


def sddmm(m, n, row_indices, row_offsets, column_indices, lhs_matrix, rhs_matrix):
    """(lhs @ rhs^T) sampled at the CSR pattern -> [nnz] / [R,nnz].
    src/sddmm_cuda.cu:7-57."""
    return _ops.sddmm(int(m), int(n), row_indices, row_offsets, column_indices, lhs_matrix,
                      rhs_matrix)


def sddmm_narrow(m, n, row_indices, row_offsets, column_indices, lhs_matrix, rhs_matrix):
    """sddmm whose result is stored in the operands' type: float16 / bfloat16 operands
    give float16 / bfloat16 scores (float32 sums, rounded once); float32 as sddmm."""
    return _ops.sddmm_narrow(int(m), int(n), row_indices, row_offsets, column_indices, lhs_matrix,
                             rhs_matrix)


def sparse_softmax(values, row_indices, row_offsets, column_indices):
    """Row-wise softmax over the stored entries.  src/softmax_cuda.cu:7-46."""
    return _ops.sparse_softmax(values, row_indices, row_offsets, column_indices)


def csr_transpose(m, n, values, row_offsets, column_indices):
    """CSR(m x n) -> [values_t, row_offsets_t, column_indices_t] of the transpose.
    src/transpose_cuda.cu:45-102.  values may also be [R,nnz] (extension)."""
    return _ops.csr_transpose(int(m), int(n), values, row_offsets, column_indices)


def csr_transpose_with_permutation(m, n, values, row_offsets, column_indices, checked=True):
    """Extension: as csr_transpose plus a 4th tensor,
    ``permutation`` with values_t == values[..., permutation], so a static
    topology's transpose can be cached by the caller.  ``checked`` (what a caller
    that KEEPS the result wants): wait for the stream and raise on a pattern the
    transpose is not defined for; ``checked=False`` is asynchronous like
    csr_transpose itself (no host round trip, legal inside a stream capture)."""
    return _ops.csr_transpose_with_permutation(int(m), int(n), values, row_offsets,
                                               column_indices, bool(checked))


# ---------------------------------------------------------------------------
# extensions
# ---------------------------------------------------------------------------
def spmm_bias(m, k, values, row_indices, row_offsets, column_indices, bias, dense_matrix):
    """spmm with ``bias[i]`` added to output row i in the kernel's epilogue.
    Call site tests/test_spmm_bias_relu.py:35-37."""
    return _ops.spmm_bias(int(m), int(k), values, row_indices, row_offsets, column_indices, bias,
                          dense_matrix)


def spmm_bias_relu(m, k, values, row_indices, row_offsets, column_indices, bias, dense_matrix):
    """``relu(spmm + bias)`` in one pass."""
    return _ops.spmm_bias_relu(int(m), int(k), values, row_indices, row_offsets, column_indices,
                               bias, dense_matrix)


def sparse_softmax_scaled(values, row_indices, row_offsets, column_indices, scale):
    """softmax(scale * values): the 1/sqrt(d) of modules/sparse_attention.py:72 folded in."""
    return _ops.sparse_softmax_scaled(values, row_indices, row_offsets, column_indices,
                                      float(scale))


def sparse_softmax_backward(softmax_out, grad_out, row_offsets, scale=1.0):
    """Gradient of softmax(scale * x) w.r.t. x given the forward output."""
    return _ops.sparse_softmax_backward(softmax_out, grad_out, row_offsets, float(scale))


def sparse_attention(query, key, value, row_indices, row_offsets, column_indices, scale):
    """softmax(scale * sddmm(query, key)) @ value over a fixed mask in ONE kernel
    (forward only): the chain of modules/sparse_attention.py:66-82 without the
    [replicas, nnz] intermediates.  query [R,S,D], key/value [R,S',D] -> [R,S,D]."""
    return _ops.sparse_attention(query, key, value, row_indices, row_offsets, column_indices,
                                 float(scale))


def sparse_attention_with_lse(query, key, value, row_indices, row_offsets, column_indices, scale):
    """As sparse_attention, returning [out, lse]: lse[r, i] = log-sum-exp of row i's
    scaled scores (-inf for rows without entries).  Head dimension 64 only."""
    return _ops.sparse_attention_with_lse(query, key, value, row_indices, row_offsets,
                                          column_indices, float(scale))


def _counts(nonzeros):
    import torch
    return nonzeros if torch.is_tensor(nonzeros) else torch.tensor(list(nonzeros),
                                                                    dtype=torch.int64)


def spmm_many_mask(b, m, k, nonzeros, values, row_indices, row_offsets, column_indices,
                   dense_matrix):
    """One mask per batch element, shared by its heads: replica r of
    values [R, max nnz] / dense [R,k,n] uses mask r // (R // b).
    tests/transformer/functions.py:20; layout tests/transformer/utils.py:17-38."""
    return _ops.spmm_many_mask(int(b), int(m), int(k), _counts(nonzeros), values, row_indices,
                               row_offsets, column_indices, dense_matrix)


def sddmm_many_mask(b, m, n, nonzeros, row_indices, row_offsets, column_indices, lhs_matrix,
                    rhs_matrix):
    """-> [R, max(nonzeros)], zero past a replica's own count.
    tests/transformer/functions.py:135; tests/test_attention_many_masks.py:120-127."""
    return _ops.sddmm_many_mask(int(b), int(m), int(n), _counts(nonzeros), row_indices,
                                row_offsets, column_indices, lhs_matrix, rhs_matrix)


def sparse_softmax_many_mask(b, m, nonzeros, values, row_indices, row_offsets, column_indices,
                             scale=None):
    """tests/transformer/functions.py:81; tests/test_attention_many_masks.py:132-138."""
    if scale is None:
        return _ops.sparse_softmax_many_mask(int(b), int(m), _counts(nonzeros), values,
                                             row_indices, row_offsets, column_indices)
    return _ops.sparse_softmax_many_mask_scaled(int(b), int(m), _counts(nonzeros), values,
                                                row_indices, row_offsets, column_indices,
                                                float(scale))


def sparse_softmax_backward_many_mask(b, m, nonzeros, softmax_out, grad_out, row_offsets,
                                      scale=1.0):
    return _ops.sparse_softmax_backward_many_mask(int(b), int(m), _counts(nonzeros), softmax_out,
                                                  grad_out, row_offsets, float(scale))


def csr_transpose_many_mask(b, m, n, nonzeros, values, row_offsets, column_indices):
    """-> [values_t [R, max nnz], row_offsets_t [b, n+1], column_indices_t [sum nnz]].
    tests/transformer/functions.py:50,165."""
    return _ops.csr_transpose_many_mask(int(b), int(m), int(n), _counts(nonzeros), values,
                                        row_offsets, column_indices)


# ---------------------------------------------------------------------------
# static topologies: plan once, run many times (no counterpart in the reference,
# which re-derives everything per call, src/spmm_cuda.cu:48-57)
# ---------------------------------------------------------------------------
def spmm_plan(m, k, n, row_indices, row_offsets, column_indices):
    """Topology-only pre-pass for spmm / left_spmm with a dense operand of width
    `n`; returns the plan (a uint8 tensor) for spmm_planned / left_spmm_planned.
    Valid as long as the three index tensors, m, k and n do not change."""
    return _ops.spmm_plan(int(m), int(k), int(n), row_indices, row_offsets, column_indices)


def spmm_planned(m, k, values, row_indices, row_offsets, column_indices, dense_matrix, plan):
    return _ops.spmm_planned(int(m), int(k), values, row_indices, row_offsets, column_indices,
                             dense_matrix, plan)


def left_spmm_planned(m, k, values, row_indices, row_offsets, column_indices, dense_matrix, plan):
    return _ops.left_spmm_planned(int(m), int(k), values, row_indices, row_offsets,
                                  column_indices, dense_matrix, plan)


def sddmm_plan(m, n, k, row_indices, row_offsets, column_indices):
    """Pre-pass for sddmm over an m x n mask with inner dimension k."""
    return _ops.sddmm_plan(int(m), int(n), int(k), row_indices, row_offsets, column_indices)


def sddmm_planned(m, n, row_indices, row_offsets, column_indices, lhs_matrix, rhs_matrix, plan):
    return _ops.sddmm_planned(int(m), int(n), row_indices, row_offsets, column_indices,
                              lhs_matrix, rhs_matrix, plan)


def permute_band_size():
    return _ops.permute_band_size()


def banded_lists(permutation):
    """(dest_list, source_in_band) of sputnik_hip_permute_banded_batched for a
    permutation (out[i] = in[permutation[i]]): the output positions grouped by the
    band of their source, ascending inside a band."""
    band = permute_band_size()
    perm = permutation.long()
    order = torch.argsort(torch.div(perm, band, rounding_mode="floor"), stable=True)
    return (order.to(torch.int32).contiguous(),
            torch.remainder(perm[order], band).to(torch.int32).contiguous())


def permute_last_banded(values, dest_list, source_in_band):
    """permute_last for many rows of values, through LDS (lists: banded_lists)."""
    return _ops.permute_last_banded(values, dest_list, source_in_band)


def spmm_permuted_fused(m, k, n, nonzeros):
    """True where spmm_permuted runs as one kernel (and that is the faster form)."""
    from . import capi
    return bool(capi.lib().sputnik_hip_spmm_permuted_supported(int(m), int(k), int(n), int(nonzeros)))


def spmm_permuted(m, k, values, permutation, row_indices, row_offsets, column_indices,
                  dense_matrix, plan=None, left=False):
    """spmm / left_spmm over a topology whose values are stored in another order of
    the same entries (entry p takes ``values[..., permutation[p]]``): the product
    with a transpose whose topology and permutation are cached.  One kernel where
    the panel kernel serves the shape, else permute_last + the planned product."""
    op = _ops.left_spmm_permuted if left else _ops.spmm_permuted
    return op(int(m), int(k), values, permutation, row_indices, row_offsets, column_indices,
              dense_matrix, plan)


def spmm_transposed_out(m, k, values, row_indices, row_offsets, column_indices, dense_matrix,
                        block_rows, permutation=None, plan=None, left=False):
    """spmm / left_spmm with the product stored as the transposes of its blocks of
    ``block_rows`` rows -> [replicas * m / block_rows, n, block_rows]: the head
    split behind a projection (block_rows = head_dim,
    modules/sparse_attention.py:38-45) or C^T (block_rows = m), written by the
    kernel's store phase where the panel kernel serves the shape."""
    return _ops.spmm_transposed_out(int(m), int(k), values, permutation, row_indices,
                                    row_offsets, column_indices, dense_matrix, int(block_rows),
                                    bool(left), plan)


def left_spmm_group(m, k, values, row_indices, row_offsets, column_indices, dense_matrix,
                    block_rows=0):
    """Several sparse weights of one shape times ONE batch of dense matrices in one
    launch (lists with one entry per weight) -> list of products, each head split
    as in spmm_transposed_out when ``block_rows`` > 0: the q, k and v projections of
    a self-attention block (modules/sparse_attention.py:108-110)."""
    return _ops.left_spmm_group(int(m), int(k), list(values), list(row_indices),
                                list(row_offsets), list(column_indices), dense_matrix,
                                int(block_rows))


def left_spmm_group_sum(m, k, values, permutations, row_indices, row_offsets, column_indices,
                        dense_matrices):
    """sum_p A_p @ dense_p in one launch, accumulated in registers (values gathered
    through ``permutations`` when the list is not empty): the input gradient of a
    group of projections."""
    return _ops.left_spmm_group_sum(int(m), int(k), list(values), list(permutations),
                                    list(row_indices), list(row_offsets), list(column_indices),
                                    list(dense_matrices))


def sddmm_sum(m, n, row_indices, row_offsets, column_indices, lhs_matrix, rhs_matrix):
    """sum over the replicas of sddmm(...) -> [nnz]: the gradient of sparse values
    shared by a batch (what autograd makes of the [R, nnz] result of
    tests/test_linear_3d.py:64-69), summed inside the call in replica order."""
    return _ops.sddmm_sum(int(m), int(n), row_indices, row_offsets, column_indices, lhs_matrix,
                          rhs_matrix)


def sddmm_sum_plan(m, n, k, row_indices, row_offsets, column_indices):
    """Pre-pass for sddmm_sum_planned (its own: not interchangeable with sddmm_plan)."""
    return _ops.sddmm_sum_plan(int(m), int(n), int(k), row_indices, row_offsets, column_indices)


def sddmm_sum_planned(m, n, row_indices, row_offsets, column_indices, lhs_matrix, rhs_matrix,
                      plan):
    return _ops.sddmm_sum_planned(int(m), int(n), row_indices, row_offsets, column_indices,
                                  lhs_matrix, rhs_matrix, plan)


def sddmm_sum_group_planned(m, n, row_indices, row_offsets, column_indices, lhs_matrices, rhs_matrix,
                            plans):
    """Up to four summed SDDMMs of one shape against ONE rhs in one call (lists with an entry
    per product; float32, [replicas, rows, k]): the weight gradients of a group of projections
    (modules/sparse_attention.py:108-110), whose partial sums one launch adds.  Bit-identical
    to sddmm_sum_planned product by product."""
    return _ops.sddmm_sum_group_planned(int(m), int(n), list(row_indices), list(row_offsets),
                                        list(column_indices), list(lhs_matrices), rhs_matrix,
                                        list(plans))


def sparse_attention_plan(m, n, d, row_indices, row_offsets, column_indices):
    """Pre-pass for the fused attention over an m x n mask with head dimension d."""
    return _ops.sparse_attention_plan(int(m), int(n), int(d), row_indices, row_offsets,
                                      column_indices)


def sparse_attention_planned(query, key, value, row_indices, row_offsets, column_indices, scale,
                             plan):
    return _ops.sparse_attention_planned(query, key, value, row_indices, row_offsets,
                                         column_indices, float(scale), plan)


_TYPE_CODES = None


def transpose_last2(x, dtype=None):
    """``x.transpose(-1, -2).contiguous()`` as one tiled HIP kernel: the layout
    pass in front of / behind every SparseLinear (modules/sparse_linear.py:89,
    modules/sparse_attention.py:108-126).  ``dtype`` (float32 / float16 /
    bfloat16) changes the storage type inside the same pass."""
    global _TYPE_CODES
    if dtype is None or dtype == x.dtype:
        return _ops.transpose_last2(x)
    if _TYPE_CODES is None:
        import torch
        _TYPE_CODES = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
    return _ops.transpose_last2_as(x, _TYPE_CODES[dtype])


def permute_last(values, permutation):
    """``values[..., permutation]`` for value arrays [nnz] / [R, nnz] sharing one
    int32 permutation (a static pattern's transposed order): one kernel, the
    permutation read once for all rows.  float32 out."""
    return _ops.permute_last(values, permutation)
