"""GPU: regression guard for the SpMM dispatcher.  The choice
of kernel is a handful of measured thresholds (csrc/spmm.hip takes_panel,
csrc/spmm_tiled.hip choose_kernel / use_flat, csrc/spmm_flat.hip flat_mode); this
test times EVERY kernel the knob can force on a fixed grid of shapes -- the
benchmark's, the modules' and the corners between them -- and requires the
automatic choice to be within 15 % of the best one (plus 3 us: launch noise on
the smallest calls).  A kernel that does not apply to a shape falls through to
the next one, so its time is simply not better.

This is the only test of the suite that asserts on wall-clock time.  The file is
named so that it collects LAST: a timing flake on a noisy box
must never stop `pytest -x` in front of the parity tests.  Each kernel is measured
twice and the better median counts."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

KERNELS = ["auto", "flat", "wide512", "wide", "narrow", "panel", "gather"]

SHAPES = [
    # m, k, n, density, replicas
    (4096, 4096, 4096, 0.10, 1),     # BASELINE config 2, headline density
    (4096, 4096, 4096, 0.50, 1),     # ... its dense end (group loop)
    (4096, 4096, 4096, 0.05, 1),     # ... its sparse end
    (4096, 4096, 4096, 0.10, 4),     # config 4's batched form (a few replicas of it)
    (2048, 2048, 512, 0.20, 8),      # config 5: SparseLinear forward, batch 8 x seq 512
    (2048, 2048, 2048, 0.20, 8),     # config 5 at its stated size (seq 2048): flat kernel by replica count
    (1024, 1024, 64, 0.10, 64),      # config 3: attention weights @ V
    (512, 512, 1024, 0.10, 8),       # config 3: a projection
    (2048, 2048, 2048, 0.10, 1),     # one mid-size product
    (4096, 4096, 256, 0.10, 1),      # narrow dense operand
    (64, 64, 64, 0.50, 1),           # config 1: launch-latency bound
    # round 5 (tools/spmm_dispatch_sweep.py): the bands where the thresholds lost 1.4-2 x
    (1024, 1024, 64, 0.30, 1),       # long rows over a short k: the K split, not the row gather
    (4096, 4096, 64, 0.02, 8),       # 2.6 entries per (row, chunk) visit: the row gather
    (2048, 2048, 256, 0.02, 1),      # ... and for one replica
]


def _median_ms(fn, iters=25, warmup=6):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    for s, e in zip(starts, ends):
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s, e in zip(starts, ends))[iters // 2]


@pytest.mark.parametrize("m,k,n,density,replicas", SHAPES)
def test_automatic_choice_is_close_to_the_best_kernel(m, k, n, density, replicas):
    from torch_sputnik_amd import capi
    from torch_sputnik_amd.synthetic import random_csr, uniform
    dev = torch.device("cuda:0")
    ri, ro, ci, nnz = random_csr(m, k, density, dev, seed=11)
    values = uniform((replicas, nnz) if replicas > 1 else (nnz,), dev, 12)
    dense = uniform((replicas, k, n) if replicas > 1 else (k, n), dev, 13)
    out = torch.empty((replicas, m, n) if replicas > 1 else (m, n), device=dev)
    times = {}
    try:
        for kern in KERNELS:
            if kern == "auto":
                os.environ.pop("SPUTNIK_HIP_SPMM_KERNEL", None)
            else:
                os.environ["SPUTNIK_HIP_SPMM_KERNEL"] = kern
            capi.reload_options()
            ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, nnz) + 16, dtype=torch.uint8, device=dev)
            # the whole call, pre-pass included: what the dispatcher's thresholds were measured on
            call = lambda: capi.spmm_batched(
                m, k, n, replicas, ri, values, nnz if replicas > 1 else 0, ro, ci, dense, out, ws)
            times[kern] = min(_median_ms(call), _median_ms(call))
        # (round 5) "auto" runs first, on a GPU that has just been handed new operands -- a
        # sweep of 180 shapes read it 5-15 % slow against the SAME kernel forced later: once
        # more at the end, the better median counts
        os.environ.pop("SPUTNIK_HIP_SPMM_KERNEL", None)
        capi.reload_options()
        ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, nnz) + 16, dtype=torch.uint8, device=dev)
        times["auto"] = min(times["auto"], _median_ms(call))
    finally:
        os.environ.pop("SPUTNIK_HIP_SPMM_KERNEL", None)
        capi.reload_options()
    best = min(times, key=times.get)
    assert times["auto"] <= 1.15 * times[best] + 0.003, (
        f"auto picks {capi.spmm_kernel_name(m, k, n, nnz, replicas)} at {times['auto']:.4f} ms; "
        f"'{best}' runs {times[best]:.4f} ms; all: " + ", ".join(f"{a} {b:.4f}" for a, b in times.items()))
