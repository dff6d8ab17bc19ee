"""ctypes view of the C ABI (include/sputnik_hip.h) for callers that hold raw
device memory: the parity tests and bench.py call the kernels through this, on
torch-allocated HIP buffers, without going through the TORCH_LIBRARY layer.

Every wrapper takes already-valid int32 / float32 contiguous GPU tensors,
passes their pointers plus the current HIP stream, and raises on a nonzero
status.  No shape inference or casting happens here (that is the op layer's
job, csrc/torch_binding.cpp).
"""
import ctypes

import torch

from ._native import kernel_lib

_c_int = ctypes.c_int
_c_i64 = ctypes.c_int64
_c_ptr = ctypes.c_void_p
_c_size = ctypes.c_size_t
_c_float = ctypes.c_float

# name -> (restype, argtypes); must list every symbol include/sputnik_hip.h declares.
SIGNATURES = {
    "sputnik_hip_version": (ctypes.c_char_p, []),
    "sputnik_hip_build_id": (ctypes.c_char_p, []),
    "sputnik_hip_spmm_kernel_name": (ctypes.c_char_p, [_c_int] * 5),
    "sputnik_hip_sddmm_kernel_name": (ctypes.c_char_p, [_c_int] * 7),
    "sputnik_hip_reload_options": (None, []),
    "sputnik_hip_spmm": (_c_int, [_c_int] * 4 + [_c_ptr] * 7),
    "sputnik_hip_spmm_workspace_bytes": (_c_size, [_c_int] * 4),
    "sputnik_hip_spmm_batched": (_c_int, [_c_int] * 5 + [_c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr,
                                                        _c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr,
                                                        _c_size, _c_ptr]),
    "sputnik_hip_spmm_plan": (_c_int, [_c_int] * 4 + [_c_ptr] * 4 + [_c_size, _c_ptr]),
    "sputnik_hip_spmm_batched_planned": (_c_int, [_c_int] * 5 + [_c_ptr, _c_ptr, _c_i64, _c_ptr,
                                                                _c_ptr, _c_ptr, _c_i64, _c_ptr,
                                                                _c_i64, _c_ptr, _c_size, _c_ptr]),
    "sputnik_hip_sddmm": (_c_int, [_c_int] * 4 + [_c_ptr] * 7),
    "sputnik_hip_sddmm_workspace_bytes": (_c_size, [_c_int] * 4),
    "sputnik_hip_sddmm_many_mask_workspace_bytes": (_c_size, [_c_int] * 5),
    "sputnik_hip_csr_transpose_many_mask_workspace_bytes": (_c_size, [_c_int] * 4),
    "sputnik_hip_sddmm_batched": (_c_int, [_c_int] * 5 + [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64,
                                                         _c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr,
                                                         _c_size, _c_ptr]),
    "sputnik_hip_sddmm_plan": (_c_int, [_c_int] * 4 + [_c_ptr] * 4 + [_c_size, _c_ptr]),
    "sputnik_hip_sddmm_batched_planned": (_c_int, [_c_int] * 5 + [_c_ptr, _c_ptr, _c_ptr, _c_ptr,
                                                                 _c_i64, _c_ptr, _c_i64, _c_ptr,
                                                                 _c_i64, _c_ptr, _c_size, _c_ptr]),
    "sputnik_hip_permute_band_size": (_c_int, []),
    "sputnik_hip_permute_banded_batched": (_c_int, [_c_int, _c_int, _c_ptr, _c_i64, _c_ptr, _c_ptr,
                                                   _c_ptr, _c_i64, _c_ptr]),
    "sputnik_hip_spmm_permuted_supported": (_c_int, [_c_int] * 4),
    "sputnik_hip_spmm_permuted_batched": (_c_int, [_c_int] * 5 + [_c_ptr, _c_ptr, _c_i64, _c_ptr,
                                                                 _c_ptr, _c_ptr, _c_ptr, _c_i64,
                                                                 _c_ptr, _c_i64, _c_ptr]),
    "sputnik_hip_spmm_group_supported": (_c_int, [_c_int] * 6),
    "sputnik_hip_spmm_group_batched": (_c_int, [_c_int] * 5 + [_c_ptr, _c_i64, _c_i64, _c_int, _c_int,
                                                              _c_ptr]),
    "sputnik_hip_spmm_transposed_out_supported": (_c_int, [_c_int] * 5),
    "sputnik_hip_spmm_transposed_out_batched": (_c_int, [_c_int] * 5 + [_c_ptr, _c_i64, _c_ptr, _c_ptr,
                                                                       _c_ptr, _c_ptr, _c_i64, _c_ptr,
                                                                       _c_int, _c_int, _c_ptr, _c_i64,
                                                                       _c_ptr]),
    "sputnik_hip_sddmm_sum_scratch_bytes": (_c_size, [_c_int] * 5),
    "sputnik_hip_sddmm_sum_workspace_bytes": (_c_size, [_c_int] * 4),
    "sputnik_hip_sddmm_sum_plan": (_c_int, [_c_int] * 4 + [_c_ptr] * 4 + [_c_size, _c_ptr]),
    "sputnik_hip_sddmm_sum_batched": (_c_int, [_c_int] * 5 + [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64,
                                                             _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_size,
                                                             _c_ptr, _c_size, _c_ptr]),
    "sputnik_hip_sddmm_sum_batched_planned": (_c_int, [_c_int] * 5 + [_c_ptr, _c_ptr, _c_ptr, _c_ptr,
                                                                     _c_i64, _c_ptr, _c_i64, _c_ptr,
                                                                     _c_ptr, _c_size, _c_ptr, _c_size,
                                                                     _c_ptr]),
    "sputnik_hip_sparse_softmax": (_c_int, [_c_int] * 3 + [_c_ptr] * 6),
    "sputnik_hip_sparse_softmax_batched": (_c_int, [_c_int] * 4 + [_c_ptr, _c_i64, _c_ptr, _c_ptr,
                                                                  _c_ptr, _c_ptr, _c_i64, _c_ptr]),
    "sputnik_hip_csr_transpose_workspace_bytes": (_c_size, [_c_int] * 3),
    "sputnik_hip_csr_transpose": (_c_int, [_c_int] * 4 + [_c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr,
                                                         _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr,
                                                         _c_size, _c_ptr]),
    "sputnik_hip_csr_transpose_checked": (_c_int, [_c_int] * 4 + [_c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr,
                                                         _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr,
                                                         _c_size, _c_ptr]),
    # extensions
    "sputnik_hip_spmm_bias_batched": (_c_int, [_c_int] * 5 + [_c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr,
                                                             _c_ptr, _c_i64, _c_ptr, _c_int, _c_ptr,
                                                             _c_i64, _c_ptr, _c_size, _c_ptr]),
    "sputnik_hip_sparse_softmax_scaled_batched": (_c_int, [_c_int] * 4 + [
        _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_float, _c_ptr, _c_i64, _c_ptr]),
    "sputnik_hip_sparse_softmax_backward_batched": (_c_int, [_c_int] * 3 + [
        _c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr, _c_float, _c_ptr, _c_i64, _c_ptr]),
    "sputnik_hip_csr_transpose_typed": (_c_int, [_c_int] * 4 + [_c_ptr, _c_int, _c_i64, _c_ptr, _c_ptr,
                                                               _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr,
                                                               _c_ptr, _c_size, _c_int, _c_ptr]),
    "sputnik_hip_spmm_typed_workspace_bytes": (_c_size, [_c_int] * 6 + [_c_i64, _c_int]),
    "sputnik_hip_spmm_typed": (_c_int, [_c_int] * 5 + [_c_ptr, _c_ptr, _c_int, _c_i64, _c_ptr, _c_ptr,
                                                      _c_ptr, _c_int, _c_i64, _c_ptr, _c_int, _c_ptr,
                                                      _c_i64, _c_ptr, _c_size, _c_ptr]),
    "sputnik_hip_sddmm_typed": (_c_int, [_c_int] * 5 + [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64,
                                                       _c_ptr, _c_i64, _c_int, _c_ptr, _c_i64,
                                                       _c_int, _c_ptr, _c_size, _c_int, _c_ptr]),
    "sputnik_hip_sddmm_sum_typed": (_c_int, [_c_int] * 5 + [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64,
                                                           _c_ptr, _c_i64, _c_int, _c_ptr, _c_ptr,
                                                           _c_size, _c_int, _c_ptr, _c_size, _c_ptr]),
    "sputnik_hip_left_spmm_half_tiles_workspace_bytes": (_c_size, [_c_int] * 8),
    "sputnik_hip_left_spmm_half_tiles": (_c_int, [_c_int] * 5 + [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr,
                                                                _c_int, _c_i64, _c_int, _c_ptr, _c_int,
                                                                _c_ptr, _c_i64, _c_ptr, _c_size, _c_ptr]),
    "sputnik_hip_sparse_linear_half_supported": (_c_int, [_c_int] * 7),
    "sputnik_hip_sparse_linear_half_image_bytes": (_c_size, [_c_int] * 4),
    "sputnik_hip_sparse_linear_half_image": (_c_int, [_c_int] * 3 + [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_int,
                                                                    _c_ptr, _c_size, _c_ptr]),
    "sputnik_hip_half_planes_bytes": (_c_size, [_c_i64, _c_int]),
    "sputnik_hip_half_planes": (_c_int, [_c_i64, _c_ptr, _c_int, _c_ptr, _c_ptr]),
    "sputnik_hip_sparse_linear_half_forward": (_c_int, [_c_int] * 4 + [_c_ptr, _c_int, _c_ptr, _c_int,
                                                                      _c_ptr, _c_int, _c_ptr, _c_ptr]),
    "sputnik_hip_sparse_linear_half_plan_bytes": (_c_size, [_c_int] * 2),
    "sputnik_hip_sparse_linear_half_plan": (_c_int, [_c_int] * 2 + [_c_ptr, _c_ptr, _c_ptr, _c_ptr]),
    "sputnik_hip_sparse_linear_half_scratch_bytes": (_c_size, [_c_int] * 7),
    "sputnik_hip_sparse_linear_half_weight_gradient": (_c_int, [_c_int] * 5 + [_c_ptr, _c_ptr, _c_ptr, _c_int,
                                                                              _c_ptr, _c_int, _c_ptr, _c_ptr,
                                                                              _c_ptr, _c_size, _c_ptr]),
    "sputnik_hip_sparse_linear_half_input_gradient": (_c_int, [_c_int] * 4 + [_c_ptr, _c_int, _c_ptr, _c_int,
                                                                             _c_int, _c_ptr, _c_int, _c_ptr]),
    "sputnik_hip_sddmm_sum_group_planned": (_c_int, [_c_int] * 5 + [_c_ptr, _c_i64, _c_i64, _c_ptr]),
    "sputnik_hip_sddmm_sum_mixed_scratch_bytes": (_c_size, [_c_int] * 7),
    "sputnik_hip_sddmm_sum_mixed": (_c_int, [_c_int] * 5 + [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int,
                                                           _c_i64, _c_ptr, _c_int, _c_i64, _c_ptr,
                                                           _c_ptr, _c_size, _c_int, _c_ptr, _c_size,
                                                           _c_ptr]),
    "sputnik_hip_sparse_softmax_typed": (_c_int, [_c_int] * 4 + [
        _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_float, _c_ptr, _c_i64, _c_int, _c_ptr]),
    "sputnik_hip_sparse_softmax_backward_typed": (_c_int, [_c_int] * 3 + [
        _c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr, _c_float, _c_ptr, _c_i64, _c_int, _c_ptr]),
    "sputnik_hip_sparse_attention_supported": (_c_int, [_c_int] * 4),
    "sputnik_hip_sparse_attention_workspace_bytes": (_c_size, [_c_int] * 4),
    "sputnik_hip_sparse_attention_forward": (_c_int, [_c_int] * 5 + [
        _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_float, _c_ptr,
        _c_i64, _c_ptr, _c_i64, _c_ptr, _c_size, _c_ptr]),
    "sputnik_hip_sparse_attention_plan": (_c_int, [_c_int] * 4 + [_c_ptr] * 4 + [_c_size, _c_ptr]),
    "sputnik_hip_sparse_attention_forward_planned": (_c_int, [_c_int] * 5 + [
        _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_float, _c_ptr,
        _c_i64, _c_ptr, _c_i64, _c_ptr, _c_size, _c_ptr]),
    "sputnik_hip_spmm_many_mask": (_c_int, [_c_int] * 4 + [_c_ptr, _c_int, _c_ptr, _c_ptr, _c_i64,
                                                          _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr,
                                                          _c_i64, _c_ptr, _c_size, _c_ptr]),
    "sputnik_hip_sddmm_many_mask": (_c_int, [_c_int] * 4 + [_c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr,
                                                           _c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr,
                                                           _c_i64, _c_ptr, _c_size, _c_ptr]),
    "sputnik_hip_sparse_softmax_many_mask": (_c_int, [_c_int] * 2 + [
        _c_ptr, _c_int, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_float, _c_ptr, _c_i64, _c_ptr]),
    "sputnik_hip_sparse_softmax_backward_many_mask": (_c_int, [_c_int] * 2 + [
        _c_ptr, _c_int, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr, _c_float, _c_ptr, _c_i64, _c_ptr]),
    "sputnik_hip_transpose_batched": (_c_int, [_c_int] * 3 + [_c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr]),
    "sputnik_hip_permute_last_batched": (_c_int, [_c_int] * 2 + [_c_ptr, _c_i64, _c_ptr, _c_ptr, _c_i64, _c_ptr]),
    "sputnik_hip_transpose_cast_batched": (_c_int, [_c_int] * 3 + [_c_ptr, _c_int, _c_i64, _c_ptr, _c_int,
                                                                  _c_i64, _c_ptr]),
    "sputnik_hip_csr_transpose_many_mask": (_c_int, [_c_int] * 3 + [
        _c_ptr, _c_int, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr,
        _c_ptr, _c_size, _c_ptr]),
}

_bound = None


def lib():
    global _bound
    if _bound is None:
        L = kernel_lib()
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library lacks the symbol
            fn.restype = restype
            fn.argtypes = argtypes
        _bound = L
    return _bound


def version():
    return lib().sputnik_hip_version().decode()


def build_id():
    """Hash of the kernel sources the loaded library was built from."""
    return lib().sputnik_hip_build_id().decode()


def spmm_kernel_name(m, k, n, nonzeros, replicas=1):
    """Device kernel the dispatcher picks for an SpMM call of this shape."""
    return lib().sputnik_hip_spmm_kernel_name(m, k, n, nonzeros, replicas).decode()


def sddmm_kernel_name(m, k, n, nonzeros, replicas=1, elem_bytes=4, planned=False):
    """Device kernel the dispatcher picks for an SDDMM call of this shape."""
    return lib().sputnik_hip_sddmm_kernel_name(m, k, n, nonzeros, replicas, elem_bytes,
                                               int(bool(planned))).decode()


def reload_options():
    """Re-read the SPUTNIK_HIP_* knobs from the environment (read once otherwise)."""
    lib().sputnik_hip_reload_options()


def _ptr(t):
    return None if t is None else _c_ptr(t.data_ptr())


def _stream(t):
    return _c_ptr(torch.cuda.current_stream(t.device).cuda_stream)


def _check(status, what):
    if status != 0:
        raise RuntimeError(f"{what} failed with status {status}")


def _require(t, dtype, name):
    if not (t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise ValueError(f"{name}: expected a contiguous {dtype} GPU tensor, got "
                         f"{t.dtype} on {t.device}, contiguous={t.is_contiguous()}")


def spmm_workspace_bytes(m, k, n, nonzeros):
    return lib().sputnik_hip_spmm_workspace_bytes(m, k, n, nonzeros)


def spmm(m, k, n, row_indices, values, row_offsets, column_indices, dense, out):
    """sputnik_hip_spmm: one 2-D SpMM, arguments in sputnik::CudaSpmm order."""
    nonzeros = column_indices.numel()
    _check(lib().sputnik_hip_spmm(m, k, n, nonzeros, _ptr(row_indices), _ptr(values),
                                  _ptr(row_offsets), _ptr(column_indices), _ptr(dense),
                                  _ptr(out), _stream(out)), "sputnik_hip_spmm")
    return out


def spmm_batched(m, k, n, replicas, row_indices, values, values_stride, row_offsets,
                 column_indices, dense, out, workspace=None):
    nonzeros = column_indices.numel()
    for t, d, nm in ((row_indices, torch.int32, "row_indices"), (values, torch.float32, "values"),
                     (row_offsets, torch.int32, "row_offsets"),
                     (column_indices, torch.int32, "column_indices"),
                     (dense, torch.float32, "dense"), (out, torch.float32, "out")):
        _require(t, d, nm)
    ws_bytes = 0 if workspace is None else workspace.numel() * workspace.element_size()
    _check(lib().sputnik_hip_spmm_batched(
        m, k, n, nonzeros, replicas, _ptr(row_indices), _ptr(values), values_stride,
        _ptr(row_offsets), _ptr(column_indices), _ptr(dense), k * n, _ptr(out), m * n,
        _ptr(workspace), ws_bytes, _stream(out)), "sputnik_hip_spmm_batched")
    return out


def spmm_plan(m, k, n, row_indices, row_offsets, column_indices, workspace):
    """Topology-only pre-pass into `workspace` (reusable by spmm_batched_planned)."""
    nonzeros = column_indices.numel()
    ws_bytes = 0 if workspace is None else workspace.numel() * workspace.element_size()
    _check(lib().sputnik_hip_spmm_plan(m, k, n, nonzeros, _ptr(row_indices), _ptr(row_offsets),
                                       _ptr(column_indices), _ptr(workspace), ws_bytes,
                                       _stream(row_offsets)), "sputnik_hip_spmm_plan")
    return workspace


def spmm_batched_planned(m, k, n, replicas, row_indices, values, values_stride, row_offsets,
                         column_indices, dense, out, workspace):
    nonzeros = column_indices.numel()
    ws_bytes = 0 if workspace is None else workspace.numel() * workspace.element_size()
    _check(lib().sputnik_hip_spmm_batched_planned(
        m, k, n, nonzeros, replicas, _ptr(row_indices), _ptr(values), values_stride,
        _ptr(row_offsets), _ptr(column_indices), _ptr(dense), k * n, _ptr(out), m * n,
        _ptr(workspace), ws_bytes, _stream(out)), "sputnik_hip_spmm_batched_planned")
    return out


def sddmm_workspace_bytes(m, k, n, nonzeros):
    return lib().sputnik_hip_sddmm_workspace_bytes(m, k, n, nonzeros)


def sddmm_batched(m, k, n, replicas, row_indices, row_offsets, column_indices, lhs, rhs, out,
                  workspace=None):
    nonzeros = column_indices.numel()
    for t, d, nm in ((row_indices, torch.int32, "row_indices"),
                     (row_offsets, torch.int32, "row_offsets"),
                     (column_indices, torch.int32, "column_indices"),
                     (lhs, torch.float32, "lhs"), (rhs, torch.float32, "rhs"),
                     (out, torch.float32, "out")):
        _require(t, d, nm)
    _check(lib().sputnik_hip_sddmm_batched(
        m, k, n, nonzeros, replicas, _ptr(row_indices), _ptr(row_offsets), _ptr(column_indices),
        _ptr(lhs), m * k, _ptr(rhs), n * k, _ptr(out), nonzeros, _ptr(workspace),
        0 if workspace is None else workspace.numel() * workspace.element_size(), _stream(out)),
        "sputnik_hip_sddmm_batched")
    return out


def sparse_softmax_batched(m, replicas, values, row_indices, row_offsets, column_indices, out):
    nonzeros = column_indices.numel()
    for t, d, nm in ((values, torch.float32, "values"), (row_indices, torch.int32, "row_indices"),
                     (row_offsets, torch.int32, "row_offsets"),
                     (column_indices, torch.int32, "column_indices"),
                     (out, torch.float32, "out")):
        _require(t, d, nm)
    _check(lib().sputnik_hip_sparse_softmax_batched(
        m, -1, nonzeros, replicas, _ptr(values), nonzeros, _ptr(row_indices), _ptr(row_offsets),
        _ptr(column_indices), _ptr(out), nonzeros, _stream(out)),
        "sputnik_hip_sparse_softmax_batched")
    return out


def csr_transpose_workspace_bytes(m, n, nonzeros):
    return lib().sputnik_hip_csr_transpose_workspace_bytes(m, n, nonzeros)


def csr_transpose(m, n, replicas, values, row_offsets, column_indices, out_values,
                  out_row_offsets, out_column_indices, out_permutation, workspace, checked=False):
    """``checked``: the synchronising entry, which raises for a pattern the transpose
    is not defined for (a row storing a column twice, a column out of range)."""
    nonzeros = column_indices.numel()
    for t, d, nm in ((values, torch.float32, "values"), (row_offsets, torch.int32, "row_offsets"),
                     (column_indices, torch.int32, "column_indices"),
                     (out_values, torch.float32, "out_values"),
                     (out_row_offsets, torch.int32, "out_row_offsets"),
                     (out_column_indices, torch.int32, "out_column_indices")):
        _require(t, d, nm)
    ws_bytes = 0 if workspace is None else workspace.numel() * workspace.element_size()
    entry = lib().sputnik_hip_csr_transpose_checked if checked else lib().sputnik_hip_csr_transpose
    _check(entry(
        m, n, nonzeros, replicas, _ptr(values), nonzeros, _ptr(row_offsets), _ptr(column_indices),
        _ptr(out_values), nonzeros, _ptr(out_row_offsets), _ptr(out_column_indices),
        _ptr(out_permutation), _ptr(workspace), ws_bytes, _stream(out_values)),
        "sputnik_hip_csr_transpose")
    return out_values, out_row_offsets, out_column_indices


# ---------------------------------------------------------------------------
# extensions
# ---------------------------------------------------------------------------
def _ws_bytes(workspace):
    return 0 if workspace is None else workspace.numel() * workspace.element_size()


def spmm_bias_batched(m, k, n, replicas, row_indices, values, values_stride, row_offsets,
                      column_indices, dense, bias, relu, out, workspace=None):
    nonzeros = column_indices.numel()
    for t, d, nm in ((row_indices, torch.int32, "row_indices"), (values, torch.float32, "values"),
                     (row_offsets, torch.int32, "row_offsets"),
                     (column_indices, torch.int32, "column_indices"),
                     (dense, torch.float32, "dense"), (out, torch.float32, "out")):
        _require(t, d, nm)
    if bias is not None:
        _require(bias, torch.float32, "bias")
        if bias.numel() != m:
            raise ValueError(f"bias: expected {m} elements, got {bias.numel()}")
    _check(lib().sputnik_hip_spmm_bias_batched(
        m, k, n, nonzeros, replicas, _ptr(row_indices), _ptr(values), values_stride,
        _ptr(row_offsets), _ptr(column_indices), _ptr(dense), k * n, _ptr(bias), int(bool(relu)),
        _ptr(out), m * n, _ptr(workspace), _ws_bytes(workspace), _stream(out)),
        "sputnik_hip_spmm_bias_batched")
    return out


def sparse_softmax_scaled_batched(m, replicas, values, row_indices, row_offsets, column_indices,
                                  scale, out):
    nonzeros = column_indices.numel()
    for t, d, nm in ((values, torch.float32, "values"), (row_indices, torch.int32, "row_indices"),
                     (row_offsets, torch.int32, "row_offsets"),
                     (column_indices, torch.int32, "column_indices"),
                     (out, torch.float32, "out")):
        _require(t, d, nm)
    _check(lib().sputnik_hip_sparse_softmax_scaled_batched(
        m, -1, nonzeros, replicas, _ptr(values), nonzeros, _ptr(row_indices), _ptr(row_offsets),
        _ptr(column_indices), float(scale), _ptr(out), nonzeros, _stream(out)),
        "sputnik_hip_sparse_softmax_scaled_batched")
    return out


def sparse_softmax_backward_batched(m, replicas, softmax_out, grad_out, row_offsets, scale,
                                    grad_values):
    nonzeros = softmax_out.shape[-1]
    for t, d, nm in ((softmax_out, torch.float32, "softmax_out"),
                     (grad_out, torch.float32, "grad_out"),
                     (row_offsets, torch.int32, "row_offsets"),
                     (grad_values, torch.float32, "grad_values")):
        _require(t, d, nm)
    _check(lib().sputnik_hip_sparse_softmax_backward_batched(
        m, nonzeros, replicas, _ptr(softmax_out), nonzeros, _ptr(grad_out), nonzeros,
        _ptr(row_offsets), float(scale), _ptr(grad_values), nonzeros, _stream(grad_values)),
        "sputnik_hip_sparse_softmax_backward_batched")
    return grad_values


TYPE_CODES = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}   # SPUTNIK_HIP_F32 / F16 / BF16


def _type_code(*tensors):
    dtype = tensors[0].dtype
    if dtype not in TYPE_CODES or any(t.dtype != dtype for t in tensors):
        raise TypeError("operands must share one of float32 / float16 / bfloat16, got "
                        + ", ".join(str(t.dtype) for t in tensors))
    return TYPE_CODES[dtype]


def csr_transpose_typed(m, n, replicas, values, row_offsets, column_indices, out_values,
                        out_row_offsets, out_column_indices, out_permutation, workspace,
                        checked=False):
    """csr_transpose of float32 / float16 / bfloat16 values; transposed values float32."""
    nonzeros = column_indices.numel()
    _require(out_values, torch.float32, "out_values")
    _check(lib().sputnik_hip_csr_transpose_typed(
        m, n, nonzeros, replicas, _ptr(values), _type_code(values), nonzeros, _ptr(row_offsets),
        _ptr(column_indices), _ptr(out_values), nonzeros, _ptr(out_row_offsets),
        _ptr(out_column_indices), _ptr(out_permutation), _ptr(workspace), _ws_bytes(workspace),
        int(bool(checked)), _stream(out_values)), "sputnik_hip_csr_transpose_typed")
    return out_values


def spmm_typed_workspace_bytes(m, k, n, nonzeros, replicas, values, values_stride, dense):
    return lib().sputnik_hip_spmm_typed_workspace_bytes(m, k, n, nonzeros, replicas,
                                                        _type_code(values), values_stride,
                                                        _type_code(dense))


def spmm_typed(m, k, n, replicas, row_indices, values, values_stride, row_offsets, column_indices,
               dense, out, workspace=None, bias=None, relu=False):
    """SpMM with values / dense stored as float32, float16 or bfloat16; out float32."""
    nonzeros = column_indices.numel()
    for t, nm in ((row_indices, "row_indices"), (row_offsets, "row_offsets"),
                  (column_indices, "column_indices")):
        _require(t, torch.int32, nm)
    _require(out, torch.float32, "out")
    for t, nm in ((values, "values"), (dense, "dense")):
        if not (t.is_cuda and t.is_contiguous()):
            raise ValueError(f"{nm}: expected a contiguous GPU tensor")
    _check(lib().sputnik_hip_spmm_typed(
        m, k, n, nonzeros, replicas, _ptr(row_indices), _ptr(values), _type_code(values),
        values_stride, _ptr(row_offsets), _ptr(column_indices), _ptr(dense), _type_code(dense),
        k * n, _ptr(bias), int(bool(relu)), _ptr(out), m * n, _ptr(workspace),
        _ws_bytes(workspace), _stream(out)), "sputnik_hip_spmm_typed")
    return out


def sddmm_typed(m, k, n, replicas, row_indices, row_offsets, column_indices, lhs, rhs, out,
                workspace=None, planned=False):
    """SDDMM on float32 / float16 / bfloat16 operands (lhs, rhs alike); out float32 or
    the operands' type."""
    nonzeros = column_indices.numel()
    for t, nm in ((row_indices, "row_indices"), (row_offsets, "row_offsets"),
                  (column_indices, "column_indices")):
        _require(t, torch.int32, nm)
    in_code = _type_code(lhs, rhs)
    out_code = _type_code(out)
    for t, nm in ((lhs, "lhs"), (rhs, "rhs"), (out, "out")):
        if not (t.is_cuda and t.is_contiguous()):
            raise ValueError(f"{nm}: expected a contiguous GPU tensor")
    _check(lib().sputnik_hip_sddmm_typed(
        m, k, n, nonzeros, replicas, _ptr(row_indices), _ptr(row_offsets), _ptr(column_indices),
        _ptr(lhs), m * k, _ptr(rhs), n * k, in_code, _ptr(out), nonzeros, out_code, _ptr(workspace),
        _ws_bytes(workspace), int(bool(planned)), _stream(out)), "sputnik_hip_sddmm_typed")
    return out


def sparse_softmax_typed(m, replicas, values, row_indices, row_offsets, column_indices, scale, out):
    """softmax(scale * values) on float32 / float16 / bfloat16 storage (in and out alike)."""
    nonzeros = column_indices.numel()
    for t, nm in ((row_indices, "row_indices"), (row_offsets, "row_offsets"),
                  (column_indices, "column_indices")):
        _require(t, torch.int32, nm)
    code = _type_code(values, out)
    _check(lib().sputnik_hip_sparse_softmax_typed(
        m, -1, nonzeros, replicas, _ptr(values), nonzeros, _ptr(row_indices), _ptr(row_offsets),
        _ptr(column_indices), float(scale), _ptr(out), nonzeros, code, _stream(out)),
        "sputnik_hip_sparse_softmax_typed")
    return out


def sparse_softmax_backward_typed(m, replicas, softmax_out, grad_out, row_offsets, scale,
                                  grad_values):
    nonzeros = softmax_out.shape[-1]
    _require(row_offsets, torch.int32, "row_offsets")
    code = _type_code(softmax_out, grad_out, grad_values)
    _check(lib().sputnik_hip_sparse_softmax_backward_typed(
        m, nonzeros, replicas, _ptr(softmax_out), nonzeros, _ptr(grad_out), nonzeros,
        _ptr(row_offsets), float(scale), _ptr(grad_values), nonzeros, code, _stream(grad_values)),
        "sputnik_hip_sparse_softmax_backward_typed")
    return grad_values


def _host_counts(nonzeros):
    """[masks] nonzero counts as a C int array living on the host."""
    counts = [int(x) for x in (nonzeros.tolist() if torch.is_tensor(nonzeros) else nonzeros)]
    return (ctypes.c_int * len(counts))(*counts), counts


def spmm_many_mask(masks, m, k, n, nonzeros, replicas, row_indices, values, row_offsets,
                   column_indices, dense, out, workspace=None):
    arr, _ = _host_counts(nonzeros)
    _check(lib().sputnik_hip_spmm_many_mask(
        masks, m, k, n, arr, replicas, _ptr(row_indices), _ptr(values), values.shape[-1],
        _ptr(row_offsets), _ptr(column_indices), _ptr(dense), k * n, _ptr(out), m * n,
        _ptr(workspace), _ws_bytes(workspace), _stream(out)), "sputnik_hip_spmm_many_mask")
    return out


def sddmm_many_mask_workspace_bytes(masks, m, k, n, largest_nonzeros):
    """One plan per mask: with it all masks run on the LDS-tiled kernel in one launch."""
    return lib().sputnik_hip_sddmm_many_mask_workspace_bytes(masks, m, k, n, largest_nonzeros)


def sddmm_many_mask(masks, m, k, n, nonzeros, replicas, row_indices, row_offsets, column_indices,
                    lhs, rhs, out, workspace=None):
    arr, _ = _host_counts(nonzeros)
    _check(lib().sputnik_hip_sddmm_many_mask(
        masks, m, k, n, arr, replicas, _ptr(row_indices), _ptr(row_offsets), _ptr(column_indices),
        _ptr(lhs), m * k, _ptr(rhs), n * k, _ptr(out), out.shape[-1], _ptr(workspace),
        _ws_bytes(workspace), _stream(out)), "sputnik_hip_sddmm_many_mask")
    return out


def sparse_softmax_many_mask(masks, m, nonzeros, replicas, values, row_indices, row_offsets,
                             column_indices, scale, out):
    arr, _ = _host_counts(nonzeros)
    _check(lib().sputnik_hip_sparse_softmax_many_mask(
        masks, m, arr, replicas, _ptr(values), values.shape[-1], _ptr(row_indices),
        _ptr(row_offsets), _ptr(column_indices), float(scale), _ptr(out), out.shape[-1],
        _stream(out)), "sputnik_hip_sparse_softmax_many_mask")
    return out


def sparse_softmax_backward_many_mask(masks, m, nonzeros, replicas, softmax_out, grad_out,
                                      row_offsets, scale, grad_values):
    arr, _ = _host_counts(nonzeros)
    _check(lib().sputnik_hip_sparse_softmax_backward_many_mask(
        masks, m, arr, replicas, _ptr(softmax_out), softmax_out.shape[-1], _ptr(grad_out),
        grad_out.shape[-1], _ptr(row_offsets), float(scale), _ptr(grad_values),
        grad_values.shape[-1], _stream(grad_values)),
        "sputnik_hip_sparse_softmax_backward_many_mask")
    return grad_values


def csr_transpose_many_mask_workspace_bytes(masks, m, n, largest_nonzeros):
    """A region of tables per mask: with it all masks are transposed by the same three launches."""
    return lib().sputnik_hip_csr_transpose_many_mask_workspace_bytes(masks, m, n, largest_nonzeros)


def csr_transpose_many_mask(masks, m, n, nonzeros, replicas, values, row_offsets, column_indices,
                            out_values, out_row_offsets, out_column_indices, out_permutation,
                            workspace):
    arr, _ = _host_counts(nonzeros)
    _check(lib().sputnik_hip_csr_transpose_many_mask(
        masks, m, n, arr, replicas, _ptr(values), 0 if values is None else values.shape[-1],
        _ptr(row_offsets), _ptr(column_indices), _ptr(out_values),
        0 if out_values is None else out_values.shape[-1], _ptr(out_row_offsets),
        _ptr(out_column_indices), _ptr(out_permutation), _ptr(workspace), _ws_bytes(workspace),
        _stream(out_row_offsets)), "sputnik_hip_csr_transpose_many_mask")
    return out_values, out_row_offsets, out_column_indices


def sparse_attention_supported(m, n, d, nonzeros):
    return bool(lib().sputnik_hip_sparse_attention_supported(m, n, d, nonzeros))


def sparse_attention_workspace_bytes(m, n, d, nonzeros):
    return lib().sputnik_hip_sparse_attention_workspace_bytes(m, n, d, nonzeros)


def sparse_attention_forward(m, n, d, replicas, row_indices, row_offsets, column_indices, q, k, v,
                             scale, out, lse=None, workspace=None):
    """Fused softmax(scale * q k^T at the mask) v.  q [R,m,d], k and v [R,n,d]."""
    nonzeros = column_indices.numel()
    for t, dt, nm in ((row_indices, torch.int32, "row_indices"),
                      (row_offsets, torch.int32, "row_offsets"),
                      (column_indices, torch.int32, "column_indices"), (q, torch.float32, "q"),
                      (k, torch.float32, "k"), (v, torch.float32, "v"),
                      (out, torch.float32, "out")):
        _require(t, dt, nm)
    _check(lib().sputnik_hip_sparse_attention_forward(
        m, n, d, nonzeros, replicas, _ptr(row_indices), _ptr(row_offsets), _ptr(column_indices),
        _ptr(q), m * d, _ptr(k), n * d, _ptr(v), n * d, float(scale), _ptr(out), m * d, _ptr(lse),
        m, _ptr(workspace), _ws_bytes(workspace), _stream(out)),
        "sputnik_hip_sparse_attention_forward")
    return out


def sddmm_plan(m, k, n, row_indices, row_offsets, column_indices, workspace):
    """Topology-only pre-pass of the tiled SDDMM kernels into `workspace`."""
    _check(lib().sputnik_hip_sddmm_plan(m, k, n, column_indices.numel(), _ptr(row_indices),
                                        _ptr(row_offsets), _ptr(column_indices), _ptr(workspace),
                                        _ws_bytes(workspace), _stream(row_offsets)),
           "sputnik_hip_sddmm_plan")
    return workspace


def sddmm_batched_planned(m, k, n, replicas, row_indices, row_offsets, column_indices, lhs, rhs,
                          out, workspace):
    nonzeros = column_indices.numel()
    _check(lib().sputnik_hip_sddmm_batched_planned(
        m, k, n, nonzeros, replicas, _ptr(row_indices), _ptr(row_offsets), _ptr(column_indices),
        _ptr(lhs), m * k, _ptr(rhs), n * k, _ptr(out), nonzeros, _ptr(workspace),
        _ws_bytes(workspace), _stream(out)), "sputnik_hip_sddmm_batched_planned")
    return out


def spmm_permuted_batched(m, k, n, replicas, row_indices, values, values_stride, permutation,
                          row_offsets, column_indices, dense, out):
    """out = A @ dense with A's values taken as values[permutation[p]]; returns the
    status (SPUTNIK_HIP_UNSUPPORTED = -2 when the shape is not served)."""
    nonzeros = column_indices.numel()
    st = lib().sputnik_hip_spmm_permuted_batched(
        m, k, n, nonzeros, replicas, _ptr(row_indices), _ptr(values), values_stride,
        _ptr(permutation), _ptr(row_offsets), _ptr(column_indices), _ptr(dense), k * n, _ptr(out),
        m * n, _stream(out))
    if st != -2:
        _check(st, "sputnik_hip_spmm_permuted_batched")
    return st


class SpmmProblem(ctypes.Structure):
    """sputnik_hip_spmm_problem (include/sputnik_hip.h)."""
    _fields_ = [("row_indices", ctypes.c_void_p), ("row_offsets", ctypes.c_void_p),
                ("column_indices", ctypes.c_void_p), ("values", ctypes.c_void_p),
                ("value_permutation", ctypes.c_void_p), ("dense", ctypes.c_void_p),
                ("out", ctypes.c_void_p), ("nonzeros", ctypes.c_int)]


def spmm_group_batched(m, k, n, replicas, problems, block_rows=0, accumulate=False):
    """`problems`: list of dicts with row_indices (or None), row_offsets,
    column_indices, values, permutation (or None), dense, out.  Returns the status
    (SPUTNIK_HIP_UNSUPPORTED = -2 when the combination is not served)."""
    array = (SpmmProblem * len(problems))()
    for slot, p in zip(array, problems):
        slot.row_indices = _ptr(p.get("row_indices"))
        slot.row_offsets = _ptr(p["row_offsets"])
        slot.column_indices = _ptr(p["column_indices"])
        slot.values = _ptr(p["values"])
        slot.value_permutation = _ptr(p.get("permutation"))
        slot.dense = _ptr(p["dense"])
        slot.out = _ptr(p["out"])
        slot.nonzeros = p["column_indices"].numel()
    st = lib().sputnik_hip_spmm_group_batched(
        m, k, n, replicas, len(problems), ctypes.cast(array, ctypes.c_void_p), k * n, m * n,
        block_rows, int(bool(accumulate)), _stream(problems[0]["out"]))
    if st != -2:
        _check(st, "sputnik_hip_spmm_group_batched")
    return st


class SddmmSumProblem(ctypes.Structure):
    """sputnik_hip_sddmm_sum_problem (include/sputnik_hip.h)."""
    _fields_ = [("row_indices", ctypes.c_void_p), ("row_offsets", ctypes.c_void_p),
                ("column_indices", ctypes.c_void_p), ("lhs", ctypes.c_void_p),
                ("rhs", ctypes.c_void_p), ("out", ctypes.c_void_p),
                ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t),
                ("scratch", ctypes.c_void_p), ("scratch_bytes", ctypes.c_size_t),
                ("nonzeros", ctypes.c_int)]


def sddmm_sum_group_planned(m, k, n, replicas, problems):
    """`problems`: list of dicts with row_indices, row_offsets, column_indices, lhs, rhs, out,
    workspace (planned: sddmm_sum_plan), scratch -- up to four summed SDDMMs of one shape,
    their partial vectors added by one launch."""
    array = (SddmmSumProblem * len(problems))()
    for slot, p in zip(array, problems):
        slot.row_indices = _ptr(p["row_indices"])
        slot.row_offsets = _ptr(p["row_offsets"])
        slot.column_indices = _ptr(p["column_indices"])
        slot.lhs = _ptr(p["lhs"])
        slot.rhs = _ptr(p["rhs"])
        slot.out = _ptr(p["out"])
        slot.workspace = _ptr(p["workspace"])
        slot.workspace_bytes = _ws_bytes(p["workspace"])
        slot.scratch = _ptr(p.get("scratch"))
        slot.scratch_bytes = _ws_bytes(p.get("scratch"))
        slot.nonzeros = p["column_indices"].numel()
    _check(lib().sputnik_hip_sddmm_sum_group_planned(
        m, k, n, replicas, len(problems), ctypes.cast(array, ctypes.c_void_p), m * k, n * k,
        _stream(problems[0]["out"])), "sputnik_hip_sddmm_sum_group_planned")


def sddmm_sum_workspace_bytes(m, k, n, nonzeros):
    return lib().sputnik_hip_sddmm_sum_workspace_bytes(m, k, n, nonzeros)


def sddmm_sum_plan(m, k, n, row_indices, row_offsets, column_indices, workspace):
    """Topology-only pre-pass of the SUMMED tiled SDDMM into `workspace`."""
    _check(lib().sputnik_hip_sddmm_sum_plan(m, k, n, column_indices.numel(), _ptr(row_indices),
                                            _ptr(row_offsets), _ptr(column_indices),
                                            _ptr(workspace), _ws_bytes(workspace),
                                            _stream(row_offsets)),
           "sputnik_hip_sddmm_sum_plan")
    return workspace


def sddmm_sum_scratch_bytes(m, k, n, nonzeros, replicas):
    return lib().sputnik_hip_sddmm_sum_scratch_bytes(m, k, n, nonzeros, replicas)


def sddmm_sum_batched(m, k, n, replicas, row_indices, row_offsets, column_indices, lhs, rhs, out,
                      workspace, scratch, planned=False):
    """out[nnz] = sum over the replicas of sddmm(lhs_r, rhs_r)."""
    nonzeros = column_indices.numel()
    fn = lib().sputnik_hip_sddmm_sum_batched_planned if planned else lib().sputnik_hip_sddmm_sum_batched
    _check(fn(m, k, n, nonzeros, replicas, _ptr(row_indices), _ptr(row_offsets),
              _ptr(column_indices), _ptr(lhs), m * k, _ptr(rhs), n * k, _ptr(out), _ptr(workspace),
              _ws_bytes(workspace), _ptr(scratch), _ws_bytes(scratch), _stream(out)),
           "sputnik_hip_sddmm_sum_batched")
    return out


def sddmm_sum_typed(m, k, n, replicas, row_indices, row_offsets, column_indices, lhs, rhs, out,
                    workspace, scratch, planned=False):
    """sddmm_sum_batched on float32 / float16 / bfloat16 operands; out float32."""
    nonzeros = column_indices.numel()
    _require(out, torch.float32, "out")
    _check(lib().sputnik_hip_sddmm_sum_typed(
        m, k, n, nonzeros, replicas, _ptr(row_indices), _ptr(row_offsets), _ptr(column_indices),
        _ptr(lhs), m * k, _ptr(rhs), n * k, _type_code(lhs, rhs), _ptr(out), _ptr(workspace),
        _ws_bytes(workspace), int(bool(planned)), _ptr(scratch), _ws_bytes(scratch), _stream(out)),
        "sputnik_hip_sddmm_sum_typed")
    return out


def left_spmm_half_tiles_workspace_bytes(m, k, n, nonzeros, replicas, values, dense, tile_dtype):
    return lib().sputnik_hip_left_spmm_half_tiles_workspace_bytes(
        m, k, n, nonzeros, replicas, _type_code(values), _type_code(dense),
        TYPE_CODES[tile_dtype])


def left_spmm_half_tiles(m, k, n, replicas, row_offsets, column_indices, values, dense, tile_dtype, out,
                         workspace, bias=None, relu=False):
    """left_spmm on the matrix cores (values shared, dense [R, k, n], out [R, m, n] float32);
    raises (status -2) where the route does not serve the call."""
    _require(out, torch.float32, "out")
    _check(lib().sputnik_hip_left_spmm_half_tiles(
        m, k, n, column_indices.numel(), replicas, _ptr(row_offsets), _ptr(column_indices),
        _ptr(values), _type_code(values), _ptr(dense), _type_code(dense), k * n,
        TYPE_CODES[tile_dtype], _ptr(bias), int(bool(relu)), _ptr(out), m * n,
        _ptr(workspace), _ws_bytes(workspace), _stream(out)), "sputnik_hip_left_spmm_half_tiles")
    return out


def sddmm_sum_mixed_scratch_bytes(m, k, n, nonzeros, replicas, lhs, rhs):
    return lib().sputnik_hip_sddmm_sum_mixed_scratch_bytes(m, k, n, nonzeros, replicas,
                                                           _type_code(lhs), _type_code(rhs))


def sddmm_sum_mixed(m, k, n, replicas, row_indices, row_offsets, column_indices, lhs, rhs, out,
                    workspace, scratch, planned=False):
    """The summed SDDMM on a (float32, half) pair of operands: the float32 one enters the
    matrix-core product as two half planes (not rounded).  Raises (status -2) where that
    route does not serve the shape."""
    nonzeros = column_indices.numel()
    _require(out, torch.float32, "out")
    _check(lib().sputnik_hip_sddmm_sum_mixed(
        m, k, n, nonzeros, replicas, _ptr(row_indices), _ptr(row_offsets), _ptr(column_indices),
        _ptr(lhs), _type_code(lhs), m * k, _ptr(rhs), _type_code(rhs), n * k, _ptr(out),
        _ptr(workspace), _ws_bytes(workspace), int(bool(planned)), _ptr(scratch), _ws_bytes(scratch),
        _stream(out)), "sputnik_hip_sddmm_sum_mixed")
    return out


def sparse_attention_plan(m, n, d, row_indices, row_offsets, column_indices, workspace):
    _check(lib().sputnik_hip_sparse_attention_plan(
        m, n, d, column_indices.numel(), _ptr(row_indices), _ptr(row_offsets),
        _ptr(column_indices), _ptr(workspace), _ws_bytes(workspace), _stream(row_offsets)),
        "sputnik_hip_sparse_attention_plan")
    return workspace


def sparse_attention_forward_planned(m, n, d, replicas, row_indices, row_offsets, column_indices,
                                     q, k, v, scale, out, lse, workspace):
    nonzeros = column_indices.numel()
    _check(lib().sputnik_hip_sparse_attention_forward_planned(
        m, n, d, nonzeros, replicas, _ptr(row_indices), _ptr(row_offsets), _ptr(column_indices),
        _ptr(q), m * d, _ptr(k), n * d, _ptr(v), n * d, float(scale), _ptr(out), m * d, _ptr(lse),
        m, _ptr(workspace), _ws_bytes(workspace), _stream(out)),
        "sputnik_hip_sparse_attention_forward_planned")
    return out


def transpose_batched(batches, rows, cols, inp, out):
    """out[b][c][r] = inp[b][r][c] for contiguous [batches, rows, cols] fp32."""
    _require(inp, torch.float32, "inp")
    _require(out, torch.float32, "out")
    _check(lib().sputnik_hip_transpose_batched(batches, rows, cols, _ptr(inp), rows * cols,
                                               _ptr(out), rows * cols, _stream(out)),
           "sputnik_hip_transpose_batched")
    return out
