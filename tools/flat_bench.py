#!/usr/bin/env python3
"""Flat-stream SpMM kernel against the visit-per-row form of the same tile
(developer tool): per-call time, planned (kernel-only) time and pre-pass time at
4096^3 for a random pattern and for a pattern with the SAME number of entries in
every (row, 32-column chunk) -- the second shows what the per-chunk rendezvous
costs when waves carry unequal work.

    python tools/flat_bench.py [--densities 0.1,0.05] [--kernels flat,wide512]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from torch_sputnik_amd import capi  # noqa: E402


def timeit(fn, iters=60, warmup=10):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    for i in range(iters):
        starts[i].record()
        fn()
        ends[i].record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in zip(starts, ends))
    return ts[len(ts) // 2]


def balanced_csr(m, k, density, dev):
    """Every row holds round(32 * density) entries in every 32-column chunk, at
    random columns of the chunk."""
    per = max(1, round(32 * density))
    rng = np.random.default_rng(5)
    chunks = k // 32
    cols = np.empty((m, chunks, per), dtype=np.int32)
    for c in range(chunks):
        r = rng.random((m, 32)).argsort(axis=1)[:, :per]
        cols[:, c, :] = np.sort(r, axis=1) + 32 * c
    ci = cols.reshape(m, -1)
    ro = (np.arange(m + 1) * ci.shape[1]).astype(np.int32)
    ri = np.arange(m, dtype=np.int32)
    return (torch.from_numpy(ri).to(dev), torch.from_numpy(ro).to(dev),
            torch.from_numpy(ci.reshape(-1).copy()).to(dev), ci.size)


def random_csr(m, k, density, dev):
    from torch_sputnik_amd.synthetic import random_csr as rc
    return rc(m, k, density, dev, seed=int(density * 1000), round_to=4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--densities", default="0.1")
    ap.add_argument("--kernels", default="flat,wide512")
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--loops", default="", help="flat kernel only: comma list of 2,3,4 = entry / group straight / "
                    "group diagonal loop (SPUTNIK_HIP_SPMM_SPARSE); empty = the dispatcher's choice")
    ap.add_argument("--patterns", default="random,balanced")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    m = k = n = args.size
    torch.manual_seed(0)
    b = torch.rand(k, n, device=dev)
    out = torch.empty(m, n, device=dev)
    for d in [float(x) for x in args.densities.split(",")]:
        for pattern in args.patterns.split(","):
            ri, ro, ci, nnz = (random_csr if pattern == "random" else balanced_csr)(m, k, d, dev)
            vals = torch.rand(nnz, device=dev)
            for kern, loop in [(kk, ll) for kk in args.kernels.split(",")
                               for ll in (args.loops.split(",") if kk == "flat" and args.loops else [""])]:
                os.environ["SPUTNIK_HIP_SPMM_KERNEL"] = kern
                if loop:
                    os.environ["SPUTNIK_HIP_SPMM_SPARSE"] = loop
                else:
                    os.environ.pop("SPUTNIK_HIP_SPMM_SPARSE", None)
                capi.reload_options()
                ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, nnz) + 16, dtype=torch.uint8,
                                 device=dev)
                call = timeit(lambda: capi.spmm_batched(m, k, n, 1, ri, vals, 0, ro, ci, b, out, ws))
                plan = timeit(lambda: capi.spmm_plan(m, k, n, ri, ro, ci, ws))
                capi.spmm_plan(m, k, n, ri, ro, ci, ws)
                kernel = timeit(lambda: capi.spmm_batched_planned(m, k, n, 1, ri, vals, 0, ro, ci, b,
                                                                  out, ws))
                print(json.dumps(dict(density=d, pattern=pattern, kernel=kern + (":" + loop if loop else ""), nnz=nnz,
                                      call_ms=round(call, 4), plan_ms=round(plan, 4),
                                      kernel_ms=round(kernel, 4),
                                      kernel_tflops=round(2.0 * nnz * n / kernel / 1e9, 2))), flush=True)
    os.environ.pop("SPUTNIK_HIP_SPMM_KERNEL", None)


if __name__ == "__main__":
    main()
