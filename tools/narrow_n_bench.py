#!/usr/bin/env python3
"""4096 x 4096 at density 0.1 against n columns, one replica (bench.py's
spmm_4096x4096_d010_by_n), whole call: the automatic kernel with and without the K split
of the 64-column kernel (SPUTNIK_HIP_SPMM_DEBUG=64 switches it off).

    python tools/narrow_n_bench.py [--ns 64,72,128,200,256,1000]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from torch_sputnik_amd import capi  # noqa: E402
from torch_sputnik_amd.synthetic import random_csr, uniform  # noqa: E402
from tools.flat_bench import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ns", default="64,72,128,200,256,1000")
    ap.add_argument("--m", type=int, default=4096)
    ap.add_argument("--density", type=float, default=0.1)
    ap.add_argument("--splits", type=lambda v: [int(x) for x in v.split(",") if x], default=[])
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    m = k = args.m
    ri, ro, ci, nnz = random_csr(m, k, args.density, dev, seed=21)
    vals = uniform((nnz,), dev, 22)
    for n in [int(x) for x in args.ns.split(",")]:
        b = uniform((k, n), dev, 23)
        out = torch.empty(m, n, device=dev)
        row = dict(n=n, kernel=capi.spmm_kernel_name(m, k, n, nnz, 1))
        for name, dbg, tile in [("split", 0, 0), ("no_split", 64, 0)] + [
                (f"split{sp}", 0, 16 + sp) for sp in args.splits]:
            os.environ["SPUTNIK_HIP_SPMM_DEBUG"] = str(dbg)
            os.environ["SPUTNIK_HIP_SPMM_MEDIUM"] = str(tile)
            capi.reload_options()
            ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, nnz) + 16, dtype=torch.uint8, device=dev)
            t = timeit(lambda: capi.spmm_batched(m, k, n, 1, ri, vals, 0, ro, ci, b, out, ws), iters=50, warmup=10)
            row[name + "_us"] = round(1000 * t, 1)
            if name == "split":   # (the topology pre-pass done once: a static pattern)
                capi.spmm_plan(m, k, n, ri, ro, ci, ws)
                tp = timeit(lambda: capi.spmm_batched_planned(m, k, n, 1, ri, vals, 0, ro, ci, b, out, ws),
                            iters=50, warmup=10)
                row[name + "_planned_us"] = round(1000 * tp, 1)
            row[name + "_tflops"] = round(2.0 * nnz * n / t / 1e9, 2)
        print(json.dumps(row), flush=True)
    os.environ.pop("SPUTNIK_HIP_SPMM_DEBUG", None)
    os.environ.pop("SPUTNIK_HIP_SPMM_MEDIUM", None)


if __name__ == "__main__":
    main()
