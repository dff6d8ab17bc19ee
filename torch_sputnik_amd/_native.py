"""Loads the two in-tree native libraries.  There is no fallback: if either is
missing the import fails and says how to build it."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
KERNEL_LIB = os.path.join(LIB_DIR, "libsputnik_hip.so")
OPS_LIB = os.path.join(LIB_DIR, "libtorch_sputnik_ops.so")

_BUILD_HINT = ("build it with `python torch_sputnik_amd/build.py` "
               "(hipcc --offload-arch=gfx950; needs no GPU)")

_kernel_lib = None
_ops_loaded = False


def kernel_lib():
    """ctypes handle of libsputnik_hip.so (the C ABI of include/sputnik_hip.h)."""
    global _kernel_lib
    if _kernel_lib is None:
        if not os.path.exists(KERNEL_LIB):
            raise ImportError(f"torch_sputnik_amd: {KERNEL_LIB} is missing; {_BUILD_HINT}")
        # `import torch` above has already mapped libamdhip64.so.7, so the
        # kernels and PyTorch share one HIP runtime (and its streams).
        _kernel_lib = ctypes.CDLL(KERNEL_LIB, mode=ctypes.RTLD_GLOBAL)
    return _kernel_lib


def load_ops():
    """Registers torch.ops.torch_sputnik.* (TORCH_LIBRARY in csrc/torch_binding.cpp)."""
    global _ops_loaded
    if not _ops_loaded:
        kernel_lib()
        if not os.path.exists(OPS_LIB):
            raise ImportError(f"torch_sputnik_amd: {OPS_LIB} is missing; {_BUILD_HINT}")
        torch.ops.load_library(OPS_LIB)
        _ops_loaded = True
    return torch.ops.torch_sputnik
