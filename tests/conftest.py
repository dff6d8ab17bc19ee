import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def _ensure_built():
    """A fresh checkout has no native libraries (they are git-ignored): build
    them once, exactly as __graft_entry__.build() does.  On the GPU box the
    snapshot carries the built files, so this is a no-op there."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_graft_entry_for_tests",
                                                  os.path.join(REPO, "__graft_entry__.py"))
    entry = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(entry)
    entry.ensure_built()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _ensure_built()


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) where no GPU exists, so a plain
    `pytest tests/` works in the CPU container too."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this process")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="module")
def cpu_ops():
    """torch.ops.torch_sputnik.* answered on CPU by the oracle -- for tests of
    the host logic only (see oracle/torch_cpu_backend.py)."""
    from oracle import torch_cpu_backend
    torch_cpu_backend.install()
    import torch_sputnik_amd
    return torch_sputnik_amd


def _reload_library_options():
    """The kernel library reads its SPUTNIK_HIP_* knobs once; the fixtures below
    change them per test and tell it to look again."""
    from torch_sputnik_amd import capi
    capi.reload_options()


@pytest.fixture(params=["auto", "wide", "wide512", "flat", "narrow", "gather", "panel"])
def spmm_kernel(request, monkeypatch):
    """Small inputs take the single-launch row-gather kernel on their own; the
    library's test knob steers them onto each tiled kernel in turn (a kernel
    that does not apply to a shape falls through to the next one)."""
    if request.param == "auto":
        monkeypatch.delenv("SPUTNIK_HIP_SPMM_KERNEL", raising=False)
    else:
        monkeypatch.setenv("SPUTNIK_HIP_SPMM_KERNEL", request.param)
    _reload_library_options()
    yield request.param
    monkeypatch.delenv("SPUTNIK_HIP_SPMM_KERNEL", raising=False)
    _reload_library_options()


@pytest.fixture(params=["plan_slabs_auto", "plan_slabs_80_rows"])
def sddmm_sum_slab(request, monkeypatch):
    """Slab rows of the summed SDDMM's 256-wide panels: large masks take 80-row slabs (two
    workgroups per CU) on their own; the knob puts the small test shapes on them too."""
    if request.param == "plan_slabs_80_rows":
        monkeypatch.setenv("SPUTNIK_HIP_SDDMM_SLAB", "80")
    else:
        monkeypatch.delenv("SPUTNIK_HIP_SDDMM_SLAB", raising=False)
    _reload_library_options()
    yield request.param
    monkeypatch.delenv("SPUTNIK_HIP_SDDMM_SLAB", raising=False)
    _reload_library_options()


@pytest.fixture(params=["auto", "tiled", "wave"])
def sddmm_kernel(request, monkeypatch):
    """As spmm_kernel, for the SDDMM dispatch (LDS-tiled / row-wave)."""
    if request.param == "auto":
        monkeypatch.delenv("SPUTNIK_HIP_SDDMM_KERNEL", raising=False)
    else:
        monkeypatch.setenv("SPUTNIK_HIP_SDDMM_KERNEL", request.param)
    _reload_library_options()
    yield request.param
    monkeypatch.delenv("SPUTNIK_HIP_SDDMM_KERNEL", raising=False)
    _reload_library_options()
