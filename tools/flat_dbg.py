#!/usr/bin/env python3
"""Timing experiments on the flat-stream SpMM kernel (developer tool): the
planned (kernel-only) time at 4096^3 with parts of the loop switched off through
SPUTNIK_HIP_SPMM_DEBUG (bit 0 no B copies, 1 no rendezvous, 2 no LDS drain at the
boundary, 3 no window loads).  Results are wrong with any bit set.

    python tools/flat_dbg.py [--density 0.1] [--bits 0,1,2,4,8,15]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from torch_sputnik_amd import capi  # noqa: E402
from tools.flat_bench import random_csr, timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--density", type=float, default=0.1)
    ap.add_argument("--bits", default="0,1,2,4,8,3,7,15")
    ap.add_argument("--kernel", default="flat")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    m = k = n = 4096
    ri, ro, ci, nnz = random_csr(m, k, args.density, dev)
    vals = torch.rand(nnz, device=dev)
    b = torch.rand(k, n, device=dev)
    out = torch.empty(m, n, device=dev)
    os.environ["SPUTNIK_HIP_SPMM_KERNEL"] = args.kernel
    for bits in [int(x) for x in args.bits.split(",")]:
        os.environ["SPUTNIK_HIP_SPMM_DEBUG"] = str(bits)
        capi.reload_options()
        ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, nnz) + 16, dtype=torch.uint8, device=dev)
        capi.spmm_plan(m, k, n, ri, ro, ci, ws)
        t = timeit(lambda: capi.spmm_batched_planned(m, k, n, 1, ri, vals, 0, ro, ci, b, out, ws),
                   iters=100, warmup=30)
        print(json.dumps(dict(density=args.density, debug=bits, kernel_ms=round(t, 4))), flush=True)


if __name__ == "__main__":
    main()
