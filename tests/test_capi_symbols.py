"""CPU: the C-ABI library loads and exports every symbol include/sputnik_hip.h
declares; the ctypes signature table covers exactly that set.  No compute."""
import ctypes
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(REPO, "include", "sputnik_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sputnik_hip_\w+)\s*\(", text)))


def test_header_declares_the_surface():
    names = declared_symbols()
    for must in ("sputnik_hip_spmm", "sputnik_hip_spmm_batched", "sputnik_hip_sddmm",
                 "sputnik_hip_sddmm_batched", "sputnik_hip_sparse_softmax",
                 "sputnik_hip_sparse_softmax_batched", "sputnik_hip_csr_transpose",
                 "sputnik_hip_csr_transpose_workspace_bytes", "sputnik_hip_spmm_workspace_bytes"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from torch_sputnik_amd import _native
    lib = ctypes.CDLL(_native.KERNEL_LIB)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in sputnik_hip.h but not exported"


def test_ctypes_table_matches_header():
    from torch_sputnik_amd import capi
    assert sorted(capi.SIGNATURES) == declared_symbols()
    capi.lib()  # binds every signature; AttributeError on a missing symbol
    assert "gfx950" in capi.version()


def test_workspace_queries_are_host_only():
    from torch_sputnik_amd import capi
    assert capi.spmm_workspace_bytes(4096, 4096, 4096, 1677724) > 0
    assert capi.spmm_workspace_bytes(64, 64, 64, 2048) > 0       # 64-column tiled kernel
    assert capi.spmm_workspace_bytes(72, 64, 72, 464) > 0        # n = 64 + 8: partial column tile
    assert capi.spmm_workspace_bytes(72, 64, 70, 464) > 0        # any n >= 64 stays on the 64-column kernel
    assert capi.spmm_workspace_bytes(72, 64, 7, 464) == 0        # narrower than a tile: row-gather kernel
    assert capi.sddmm_workspace_bytes(1024, 64, 1024, 104860) > 0
    assert capi.sddmm_workspace_bytes(72, 72, 72, 500) == 0
    assert capi.csr_transpose_workspace_bytes(2048, 2048, 838864) >= 4 * 2048
    assert capi.csr_transpose_workspace_bytes(0, 0, 0) == 0


def test_ops_registered_without_cpu_kernels():
    """The product registers HIP kernels only: CPU tensors must raise."""
    import pytest
    import torch
    import torch_sputnik
    for name in ("spmm", "left_spmm", "left_replicated_spmm", "sddmm", "sparse_softmax",
                 "csr_transpose"):
        assert callable(getattr(torch_sputnik, name))
    has_cpu = torch._C._dispatch_has_kernel_for_dispatch_key("torch_sputnik::spmm", "CPU")
    if not has_cpu:  # (another test module may have installed the oracle backend)
        with pytest.raises((NotImplementedError, RuntimeError)):
            torch_sputnik.spmm(2, 2, torch.ones(2), torch.arange(2, dtype=torch.int32),
                               torch.tensor([0, 1, 2], dtype=torch.int32),
                               torch.tensor([0, 1], dtype=torch.int32), torch.ones(2, 2))
    assert torch._C._dispatch_has_kernel_for_dispatch_key("torch_sputnik::spmm", "CUDA")
