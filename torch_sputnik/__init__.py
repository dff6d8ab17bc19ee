"""Drop-in for the reference's ``torch_sputnik`` extension module
(src/sputnik.cpp:36-42, built by setup.py:14-17): the same five callables,
so ``import torch_sputnik`` in modules/spmm.py, modules/sddmm.py,
modules/sparse_linear.py and modules/sparse_attention.py keeps working
unchanged on an MI355X.  The implementation lives in ``torch_sputnik_amd``.

Also exported: the functions the reference's test scripts call but its
committed binding does not define (``spmm_bias``, tests/test_spmm_bias_relu.py:37;
the ``*_many_mask`` family, tests/transformer/functions.py), and the softmax
gradient.
"""
from torch_sputnik_amd.ops import (  # noqa: F401
    csr_transpose,
    csr_transpose_many_mask,
    csr_transpose_with_permutation,
    left_replicated_spmm,
    left_spmm,
    left_spmm_group,
    left_spmm_group_sum,
    sddmm,
    sddmm_many_mask,
    sddmm_plan,
    sddmm_planned,
    sddmm_sum,
    sddmm_sum_plan,
    sddmm_sum_planned,
    sparse_attention,
    sparse_attention_plan,
    sparse_attention_planned,
    sparse_attention_with_lse,
    sparse_softmax,
    sparse_softmax_backward,
    sparse_softmax_backward_many_mask,
    sparse_softmax_many_mask,
    sparse_softmax_scaled,
    spmm,
    spmm_bias,
    spmm_bias_relu,
    spmm_many_mask,
    spmm_permuted,
    spmm_transposed_out,
    permute_last,
    spmm_plan,
    spmm_planned,
    left_spmm_planned,
)

__all__ = ["spmm", "left_spmm", "left_replicated_spmm", "sddmm", "sparse_softmax",
           "csr_transpose", "csr_transpose_with_permutation", "spmm_bias", "spmm_bias_relu",
           "sparse_softmax_scaled", "sparse_softmax_backward", "spmm_many_mask",
           "sddmm_many_mask", "sparse_softmax_many_mask", "sparse_softmax_backward_many_mask",
           "csr_transpose_many_mask", "sparse_attention", "sparse_attention_with_lse", "spmm_plan", "spmm_planned",
           "left_spmm_planned", "sddmm_plan", "sddmm_planned", "sddmm_sum", "sddmm_sum_plan", "spmm_permuted", "left_spmm_group", "left_spmm_group_sum", "spmm_transposed_out", "permute_last", "sddmm_sum_planned", "sparse_attention_plan",
           "sparse_attention_planned"]
