"""ctypes wrapper around oracle/libsputnik_oracle.so (TEST INFRASTRUCTURE ONLY).

The C restatement is the checker at sizes where the numpy one is too slow
(BASELINE.json's 4096^3 configs) and the sparse CPU baseline bench.py times.
Build it with ``make -C oracle`` (``__graft_entry__.build()`` does).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsputnik_oracle.so")
_lib = None

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_int = ctypes.c_int


def available():
    return os.path.exists(_LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError(f"{_LIB_PATH} missing: run `make -C oracle`")
        L = ctypes.CDLL(_LIB_PATH)
        L.oracle_num_threads.restype = _int
        L.oracle_set_num_threads.argtypes = [_int]
        for name in ("oracle_spmm_f64acc", "oracle_spmm_f32"):
            getattr(L, name).argtypes = [_int, _int, _int, _i32p, _i32p, _f32p, _f32p, _f32p]
            getattr(L, name).restype = None
        L.oracle_sddmm_f64acc.argtypes = [_int, _int, _int, _i32p, _i32p, _f32p, _f32p, _f32p]
        L.oracle_sddmm_f64acc.restype = None
        L.oracle_softmax_f64.argtypes = [_int, _i32p, _f32p, _f32p]
        L.oracle_softmax_f64.restype = None
        L.oracle_csr_transpose.argtypes = [_int, _int, _int, _int, _f32p, _i32p, _i32p,
                                           _f32p, _i32p, _i32p]
        L.oracle_csr_transpose.restype = None
        _lib = L
    return _lib


def _f32(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def _i32(x):
    return np.ascontiguousarray(x, dtype=np.int32)


def num_threads():
    return lib().oracle_num_threads()


def set_num_threads(t):
    lib().oracle_set_num_threads(int(t))


def spmm(m, k, values, row_offsets, column_indices, dense, f32_accumulate=False):
    """2-D or batched SpMM; ``values`` [nnz] may be shared by a 3-D ``dense``."""
    values, dense = _f32(values), _f32(dense)
    row_offsets, column_indices = _i32(row_offsets), _i32(column_indices)
    fn = lib().oracle_spmm_f32 if f32_accumulate else lib().oracle_spmm_f64acc
    n = dense.shape[-1]
    if dense.ndim == 2:
        out = np.empty((m, n), np.float32)
        fn(m, k, n, row_offsets, column_indices, values, dense, out)
        return out
    out = np.empty((dense.shape[0], m, n), np.float32)
    for r in range(dense.shape[0]):
        v = values if values.ndim == 1 else values[r]
        fn(m, k, n, row_offsets, column_indices, np.ascontiguousarray(v), dense[r], out[r])
    return out


def sddmm(m, n, row_offsets, column_indices, lhs, rhs):
    lhs, rhs = _f32(lhs), _f32(rhs)
    row_offsets, column_indices = _i32(row_offsets), _i32(column_indices)
    nnz, k = column_indices.shape[0], lhs.shape[-1]
    if lhs.ndim == 2:
        out = np.empty((nnz,), np.float32)
        lib().oracle_sddmm_f64acc(m, k, n, row_offsets, column_indices, lhs, rhs, out)
        return out
    out = np.empty((lhs.shape[0], nnz), np.float32)
    for r in range(lhs.shape[0]):
        lib().oracle_sddmm_f64acc(m, k, n, row_offsets, column_indices, lhs[r], rhs[r], out[r])
    return out


def sparse_softmax(values, row_offsets, column_indices):
    values = _f32(values)
    row_offsets = _i32(row_offsets)
    m = row_offsets.shape[0] - 1
    out = np.zeros_like(values)
    if values.ndim == 1:
        lib().oracle_softmax_f64(m, row_offsets, values, out)
    else:
        for r in range(values.shape[0]):
            lib().oracle_softmax_f64(m, row_offsets, values[r], out[r])
    return out


def csr_transpose(m, n, values, row_offsets, column_indices):
    values = _f32(values)
    row_offsets, column_indices = _i32(row_offsets), _i32(column_indices)
    nnz = column_indices.shape[0]
    replicas = 1 if values.ndim == 1 else values.shape[0]
    values_t = np.empty_like(values)
    row_offsets_t = np.empty((n + 1,), np.int32)
    column_indices_t = np.empty((nnz,), np.int32)
    lib().oracle_csr_transpose(m, n, nnz, replicas, values, row_offsets, column_indices,
                               values_t, row_offsets_t, column_indices_t)
    return values_t, row_offsets_t, column_indices_t
