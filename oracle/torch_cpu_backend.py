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


def _csr_transpose_with_permutation(m, n, values, row_offsets, column_indices):
    out = _csr_transpose(m, n, values, row_offsets, column_indices)
    perm = np.argsort(_np(column_indices).astype(np.int64), kind="stable").astype(np.int32)
    return out + [torch.from_numpy(perm)]


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
    _lib.impl("sparse_softmax", _sparse_softmax, "CPU")
    _lib.impl("csr_transpose", _csr_transpose, "CPU")
    _lib.impl("csr_transpose_with_permutation", _csr_transpose_with_permutation, "CPU")
