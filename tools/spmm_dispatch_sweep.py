#!/usr/bin/env python3
"""Every SpMM kernel the knob can force against the automatic choice on a grid of shapes
(whole call, pre-pass included) -- the wide version of tests/test_zz_gpu_dispatch.py, to
find the bands where the dispatcher's thresholds lose.  One JSON line per shape.

    python tools/spmm_dispatch_sweep.py [--sizes 512,1024,2048,4096] [--ks 256,1024] [--ns 64,128,256,512,1024]
                                        [--replicas 1,8,64] [--densities 0.02,0.1,0.3]
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

KERNELS = ["auto", "flat", "wide512", "wide", "narrow", "panel", "gather"]


def ints(v):
    return [int(x) for x in v.split(",") if x]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", type=ints, default=[512, 1024, 2048, 4096])
    ap.add_argument("--ks", type=ints, default=[], help="inner dimensions (default: k = m)")
    ap.add_argument("--ns", type=ints, default=[64, 128, 256, 512, 1024])
    ap.add_argument("--replicas", type=ints, default=[1, 8, 64])
    ap.add_argument("--densities", type=lambda v: [float(x) for x in v.split(",")], default=[0.02, 0.1, 0.3])
    ap.add_argument("--max-elements", type=float, default=3e8, help="skip shapes with more output elements")
    ap.add_argument("--half", action="store_true",
                    help="float16 values and dense operand through sputnik_hip_spmm_typed (its own "
                         "dispatch: matrix-core tiles / panel / row gather / widening + float kernels)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for m, k in [(m, k) for m in args.sizes for k in (args.ks or [m])]:
        for d in args.densities:
            ri, ro, ci, nnz = random_csr(m, k, d, dev, seed=11)
            for n in args.ns:
                for reps in args.replicas:
                    if float(reps) * max(m, k) * n > args.max_elements:
                        continue
                    values = uniform((reps, nnz) if reps > 1 else (nnz,), dev, 12)
                    dense = uniform((reps, k, n) if reps > 1 else (k, n), dev, 13)
                    if args.half:
                        values, dense = values.half(), dense.half()
                    out = torch.empty((reps, m, n) if reps > 1 else (m, n), device=dev)
                    row = {"m": m, "k": k, "n": n, "replicas": reps, "density": d, "nnz": nnz}
                    for kern in KERNELS:
                        if kern == "auto":
                            os.environ.pop("SPUTNIK_HIP_SPMM_KERNEL", None)
                        else:
                            os.environ["SPUTNIK_HIP_SPMM_KERNEL"] = kern
                        capi.reload_options()
                        if args.half:
                            vs = nnz if reps > 1 else 0
                            ws = torch.empty(capi.spmm_typed_workspace_bytes(m, k, n, nnz, reps, values, vs,
                                                                             dense) + 256, dtype=torch.uint8,
                                             device=dev)
                            t = timeit(lambda: capi.spmm_typed(m, k, n, reps, ri, values, vs, ro, ci, dense, out,
                                                               ws), iters=15, warmup=4)
                        else:
                            ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, nnz) + 16, dtype=torch.uint8,
                                             device=dev)
                            t = timeit(lambda: capi.spmm_batched(m, k, n, reps, ri, values,
                                                                 nnz if reps > 1 else 0, ro, ci, dense, out, ws),
                                       iters=15, warmup=4)
                        row[kern] = round(1000 * t, 1)
                        if kern == "auto":
                            row["auto_kernel"] = capi.spmm_kernel_name(m, k, n, nnz, reps)[:40]
                    best = min(KERNELS[1:], key=lambda kk: row[kk])
                    row["best"] = best
                    row["regret"] = round(row["auto"] / row[best] - 1, 3)
                    print(json.dumps(row), flush=True)
                    del values, dense, out
    os.environ.pop("SPUTNIK_HIP_SPMM_KERNEL", None)
    capi.reload_options()


if __name__ == "__main__":
    main()
