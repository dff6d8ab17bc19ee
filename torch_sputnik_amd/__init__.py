"""MI355X-native (gfx950) implementation of the torch_sputnik operator surface.

Layers, bottom up:
  csrc/*.hip            hand-written HIP kernels behind the C ABI include/sputnik_hip.h
  csrc/torch_binding    torch.ops.torch_sputnik.* (TORCH_LIBRARY, HIP tensors only)
  ops                   the reference's five callables (src/sputnik.cpp:36-42)
  functional / modules  autograd.Functions and nn.Modules mirroring modules/*.py
  sharding              replica-dimension sharding over the GPUs of a node (RCCL)

Importing this package loads the native libraries and fails loudly when they
have not been built (``python torch_sputnik_amd/build.py``).
"""
from . import ops  # noqa: F401  (loads libsputnik_hip.so + libtorch_sputnik_ops.so)
from .ops import (  # noqa: F401
    csr_transpose,
    csr_transpose_with_permutation,
    left_replicated_spmm,
    left_spmm,
    sddmm,
    sparse_softmax,
    spmm,
)
from .topology import dense_to_sparse, diffsort, generate_mask  # noqa: F401
from .functional import Sddmm, SparseLinearFunction, SparseSoftmax, Spmm  # noqa: F401
from .modules import SparseAttention, SparseLinear  # noqa: F401

__version__ = "0.3.0"
