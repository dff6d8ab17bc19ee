"""Registers the numpy oracle as the CPU kernels of torch.ops.torch_sputnik.*
-- TEST INFRASTRUCTURE ONLY.

The product registers HIP kernels only (csrc/torch_binding.cpp) and raises on
CPU tensors.  Tests of the *host logic* (autograd wrappers, modules, replica
sharding, the reference's own modules imported against this package) need the
ops to run without a GPU; they call ``install()`` from tests/conftest.py, which
adds the oracle under the CPU dispatch key for the life of the test process.
Nothing outside tests/ and oracle/make_golden.py imports this module.
"""
import numpy as np
import torch

from . import sputnik_oracle as O

_lib = None


def _np(t):
    return t.detach().cpu().numpy()


def _spmm(m, k, values, row_indices, row_offsets, column_indices, dense):
    assert values.dim() == dense.dim() - 1
    out = O.spmm(m, k, _np(values), _np(row_indices), _np(row_offsets), _np(column_indices),
                 _np(dense))
    if out.ndim == 3 and out.shape[0] == 1:
        out = out[0]  # src/spmm_cuda.cu:46: 2-D whenever replication == 1
    return torch.from_numpy(out.astype(np.float32))


def _left_spmm(m, k, values, row_indices, row_offsets, column_indices, dense):
    out = O.left_spmm(m, k, _np(values), _np(row_indices), _np(row_offsets),
                      _np(column_indices), _np(dense))
    return torch.from_numpy(out.astype(np.float32))


def _sddmm(m, n, row_indices, row_offsets, column_indices, lhs, rhs):
    out = O.sddmm(m, n, _np(row_indices), _np(row_offsets), _np(column_indices), _np(lhs),
                  _np(rhs))
    if out.ndim == 2 and out.shape[0] == 1:
        out = out[0]  # src/sddmm_cuda.cu:43
    return torch.from_numpy(out.astype(np.float32))


def _sparse_softmax(values, row_indices, row_offsets, column_indices):
    out = O.sparse_softmax(_np(values), _np(row_indices), _np(row_offsets), _np(column_indices))
    return torch.from_numpy(out.astype(np.float32))


def _csr_transpose(m, n, values, row_offsets, column_indices):
    v, ro, ci = O.csr_transpose(m, n, _np(values), _np(row_offsets), _np(column_indices))
    return [torch.from_numpy(np.ascontiguousarray(v)), torch.from_numpy(ro), torch.from_numpy(ci)]


def _csr_transpose_with_permutation(m, n, values, row_offsets, column_indices, checked=True):
    out = _csr_transpose(m, n, values, row_offsets, column_indices)
    perm = np.argsort(_np(column_indices).astype(np.int64), kind="stable").astype(np.int32)
    return out + [torch.from_numpy(perm)]


def _f32(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def _spmm_bias(m, k, values, row_indices, row_offsets, column_indices, bias, dense, relu=False):
    out = O.spmm_bias(m, k, _np(values), _np(row_indices), _np(row_offsets), _np(column_indices),
                      _np(bias), _np(dense), relu=relu)
    if out.ndim == 3 and out.shape[0] == 1:
        out = out[0]
    return _f32(out)


def _spmm_bias_relu(m, k, values, row_indices, row_offsets, column_indices, bias, dense):
    return _spmm_bias(m, k, values, row_indices, row_offsets, column_indices, bias, dense, True)


def _sparse_softmax_scaled(values, row_indices, row_offsets, column_indices, scale):
    return _f32(O.sparse_softmax_scaled(_np(values), _np(row_indices), _np(row_offsets),
                                        _np(column_indices), scale))


def _sparse_softmax_backward(softmax_out, grad_out, row_offsets, scale):
    return _f32(O.sparse_softmax_backward(_np(softmax_out), _np(grad_out), _np(row_offsets),
                                          scale))


def _sparse_attention(q, k, v, row_indices, row_offsets, column_indices, scale):
    return _f32(O.sparse_attention(_np(q), _np(k), _np(v), _np(row_indices), _np(row_offsets),
                                   _np(column_indices), scale))


def _sparse_attention_with_lse(q, k, v, row_indices, row_offsets, column_indices, scale):
    out = _sparse_attention(q, k, v, row_indices, row_offsets, column_indices, scale)
    qn, kn = _np(q).astype(np.float64), _np(k).astype(np.float64)
    ro = _np(row_offsets).astype(np.int64)
    scores = O.sddmm(qn.shape[-2], kn.shape[-2], _np(row_indices), ro, _np(column_indices), qn,
                     kn) * scale
    scores = scores.reshape(-1, scores.shape[-1])
    lse = np.full((scores.shape[0], ro.shape[0] - 1), -np.inf)
    for i in range(ro.shape[0] - 1):
        if ro[i + 1] > ro[i]:
            seg = scores[:, ro[i]:ro[i + 1]]
            mx = seg.max(axis=1)
            lse[:, i] = mx + np.log(np.exp(seg - mx[:, None]).sum(axis=1))
    lse = lse if q.dim() == 3 else lse[0]
    return [out, _f32(lse)]


def _spmm_many_mask(b, m, k, nonzeros, values, row_indices, row_offsets, column_indices, dense):
    return _f32(O.spmm_many_mask(b, m, k, _np(nonzeros), _np(values), _np(row_indices),
                                 _np(row_offsets), _np(column_indices), _np(dense)))


def _sddmm_many_mask(b, m, n, nonzeros, row_indices, row_offsets, column_indices, lhs, rhs):
    return _f32(O.sddmm_many_mask(b, m, n, _np(nonzeros), _np(row_indices), _np(row_offsets),
                                  _np(column_indices), _np(lhs), _np(rhs)))


def _sparse_softmax_many_mask_scaled(b, m, nonzeros, values, row_indices, row_offsets,
                                     column_indices, scale):
    return _f32(O.sparse_softmax_many_mask(b, m, _np(nonzeros), _np(values), _np(row_indices),
                                           _np(row_offsets), _np(column_indices), scale))


def _sparse_softmax_many_mask(b, m, nonzeros, values, row_indices, row_offsets, column_indices):
    return _sparse_softmax_many_mask_scaled(b, m, nonzeros, values, row_indices, row_offsets,
                                            column_indices, 1.0)


def _sparse_softmax_backward_many_mask(b, m, nonzeros, softmax_out, grad_out, row_offsets, scale):
    return _f32(O.sparse_softmax_backward_many_mask(b, m, _np(nonzeros), _np(softmax_out),
                                                    _np(grad_out), _np(row_offsets), scale))


def _csr_transpose_many_mask(b, m, n, nonzeros, values, row_offsets, column_indices):
    v, ro, ci = O.csr_transpose_many_mask(b, m, n, _np(nonzeros), _np(values), _np(row_offsets),
                                          _np(column_indices))
    return [_f32(v), torch.from_numpy(ro), torch.from_numpy(ci)]


def _permute_last(values, permutation):
    return values.float().index_select(-1, permutation.long())


def _transpose_last2(x):
    return x.transpose(-1, -2).contiguous()


def _transpose_last2_as(x, out_type):
    t = x.transpose(-1, -2).contiguous()
    return t if out_type < 0 else t.to({0: torch.float32, 1: torch.float16, 2: torch.bfloat16}[out_type])


def _plan(*_args):
    return torch.zeros(16, dtype=torch.uint8)  # the CPU checker has nothing to pre-compute


def _spmm_planned(m, k, values, row_indices, row_offsets, column_indices, dense, plan):
    return _spmm(m, k, values, row_indices, row_offsets, column_indices, dense)


def _left_spmm_planned(m, k, values, row_indices, row_offsets, column_indices, dense, plan):
    return _left_spmm(m, k, values, row_indices, row_offsets, column_indices, dense)


def _sddmm_planned(m, n, row_indices, row_offsets, column_indices, lhs, rhs, plan):
    return _sddmm(m, n, row_indices, row_offsets, column_indices, lhs, rhs)


def _spmm_permuted(m, k, values, permutation, row_indices, row_offsets, column_indices, dense,
                   plan=None):
    return _spmm(m, k, values[..., permutation.long()].contiguous(), row_indices, row_offsets,
                 column_indices, dense)


def _left_spmm_permuted(m, k, values, permutation, row_indices, row_offsets, column_indices,
                        dense, plan=None):
    return _left_spmm(m, k, values[permutation.long()].contiguous(), row_indices, row_offsets,
                      column_indices, dense)


def _permute_last_banded(values, dest_list, source_in_band):
    import torch_sputnik_amd.ops as product_ops
    band = product_ops.permute_band_size()
    t = torch.arange(dest_list.numel())
    source = torch.div(t, band, rounding_mode="floor") * band + source_in_band.long()
    out = torch.empty_like(values)
    out[..., dest_list.long()] = values[..., source]
    return out


def _spmm_transposed_out(m, k, values, permutation, row_indices, row_offsets, column_indices,
                         dense, block_rows, left, plan=None):
    if permutation is not None:
        values = values[..., permutation.long()].contiguous()
    c = (_left_spmm if left else _spmm)(m, k, values, row_indices, row_offsets, column_indices,
                                        dense)
    n = c.shape[-1]
    return c.reshape(-1, block_rows, n).transpose(1, 2).contiguous()


def _left_spmm_group(m, k, values, row_indices, row_offsets, column_indices, dense, block_rows):
    outs = []
    for v, ri, ro, ci in zip(values, row_indices, row_offsets, column_indices):
        if block_rows > 0:
            outs.append(_spmm_transposed_out(m, k, v, None, ri, ro, ci, dense, block_rows, True))
        else:
            outs.append(_left_spmm(m, k, v, ri, ro, ci, dense))
    return outs


def _left_spmm_group_sum(m, k, values, permutations, row_indices, row_offsets, column_indices,
                         dense):
    total = None
    for p, (v, ri, ro, ci, d) in enumerate(zip(values, row_indices, row_offsets, column_indices,
                                               dense)):
        if len(permutations):
            v = v[permutations[p].long()].contiguous()
        part = _left_spmm(m, k, v, ri, ro, ci, d)
        total = part if total is None else total + part
    return total


def _sddmm_narrow(m, n, row_indices, row_offsets, column_indices, lhs, rhs):
    out = _sddmm(m, n, row_indices, row_offsets, column_indices, lhs.float(), rhs.float())
    return out.to(torch.promote_types(lhs.dtype, rhs.dtype))


def _sddmm_sum(m, n, row_indices, row_offsets, column_indices, lhs, rhs):
    out = _sddmm(m, n, row_indices, row_offsets, column_indices, lhs, rhs)
    return out.sum(dim=0) if out.dim() == 2 else out


def _sddmm_sum_planned(m, n, row_indices, row_offsets, column_indices, lhs, rhs, plan):
    return _sddmm_sum(m, n, row_indices, row_offsets, column_indices, lhs, rhs)


def _sddmm_sum_group_planned(m, n, row_indices, row_offsets, column_indices, lhs, rhs, plans):
    return [_sddmm_sum(m, n, ri, ro, ci, left, rhs)
            for ri, ro, ci, left in zip(row_indices, row_offsets, column_indices, lhs)]


def _sparse_attention_planned(q, k, v, row_indices, row_offsets, column_indices, scale, plan):
    return _sparse_attention(q, k, v, row_indices, row_offsets, column_indices, scale)


def install():
    """Idempotent.  Needs the product's op schemas, so it imports the package
    (which loads the native libraries; no GPU is touched)."""
    global _lib
    if _lib is not None:
        return
    import torch_sputnik_amd  # noqa: F401  (defines the torch_sputnik:: schemas)
    _lib = torch.library.Library("torch_sputnik", "IMPL")
    _lib.impl("spmm", _spmm, "CPU")
    _lib.impl("left_spmm", _left_spmm, "CPU")
    _lib.impl("sddmm", _sddmm, "CPU")
    _lib.impl("sddmm_narrow", _sddmm_narrow, "CPU")
    _lib.impl("sparse_softmax", _sparse_softmax, "CPU")
    _lib.impl("csr_transpose", _csr_transpose, "CPU")
    _lib.impl("csr_transpose_with_permutation", _csr_transpose_with_permutation, "CPU")
    _lib.impl("spmm_bias", _spmm_bias, "CPU")
    _lib.impl("spmm_bias_relu", _spmm_bias_relu, "CPU")
    _lib.impl("sparse_softmax_scaled", _sparse_softmax_scaled, "CPU")
    _lib.impl("sparse_softmax_backward", _sparse_softmax_backward, "CPU")
    _lib.impl("sparse_attention", _sparse_attention, "CPU")
    _lib.impl("sparse_attention_with_lse", _sparse_attention_with_lse, "CPU")
    _lib.impl("spmm_plan", _plan, "CPU")
    _lib.impl("sddmm_plan", _plan, "CPU")
    _lib.impl("sddmm_sum_plan", _plan, "CPU")
    _lib.impl("sparse_attention_plan", _plan, "CPU")
    _lib.impl("spmm_planned", _spmm_planned, "CPU")
    _lib.impl("left_spmm_planned", _left_spmm_planned, "CPU")
    _lib.impl("sddmm_planned", _sddmm_planned, "CPU")
    _lib.impl("sddmm_sum", _sddmm_sum, "CPU")
    _lib.impl("permute_last_banded", _permute_last_banded, "CPU")
    _lib.impl("spmm_permuted", _spmm_permuted, "CPU")
    _lib.impl("spmm_transposed_out", _spmm_transposed_out, "CPU")
    _lib.impl("left_spmm_group", _left_spmm_group, "CPU")
    _lib.impl("left_spmm_group_sum", _left_spmm_group_sum, "CPU")
    _lib.impl("left_spmm_permuted", _left_spmm_permuted, "CPU")
    _lib.impl("sddmm_sum_planned", _sddmm_sum_planned, "CPU")
    _lib.impl("sddmm_sum_group_planned", _sddmm_sum_group_planned, "CPU")
    _lib.impl("sparse_attention_planned", _sparse_attention_planned, "CPU")
    _lib.impl("spmm_many_mask", _spmm_many_mask, "CPU")
    _lib.impl("sddmm_many_mask", _sddmm_many_mask, "CPU")
    _lib.impl("sparse_softmax_many_mask", _sparse_softmax_many_mask, "CPU")
    _lib.impl("sparse_softmax_many_mask_scaled", _sparse_softmax_many_mask_scaled, "CPU")
    _lib.impl("sparse_softmax_backward_many_mask", _sparse_softmax_backward_many_mask, "CPU")
    _lib.impl("csr_transpose_many_mask", _csr_transpose_many_mask, "CPU")
    _lib.impl("permute_last", _permute_last, "CPU")
    _lib.impl("transpose_last2", _transpose_last2, "CPU")
    _lib.impl("transpose_last2_as", _transpose_last2_as, "CPU")
