"""Drop-in for the reference's ``torch_sputnik`` extension module
(src/sputnik.cpp:36-42, built by setup.py:14-17): the same five callables,
so ``import torch_sputnik`` in modules/spmm.py, modules/sddmm.py,
modules/sparse_linear.py and modules/sparse_attention.py keeps working
unchanged on an MI355X.  The implementation lives in ``torch_sputnik_amd``.
"""
from torch_sputnik_amd.ops import (  # noqa: F401
    csr_transpose,
    csr_transpose_with_permutation,
    left_replicated_spmm,
    left_spmm,
    sddmm,
    sparse_softmax,
    spmm,
)

__all__ = ["spmm", "left_spmm", "left_replicated_spmm", "sddmm", "sparse_softmax",
           "csr_transpose", "csr_transpose_with_permutation"]
