#!/usr/bin/env python3
"""Time of the flat-stream SpMM's pre-pass alone (sputnik_hip_spmm_plan: count + fill
kernels) at 4096^2, with parts of the fill kernel switched off through
SPUTNIK_HIP_SPMM_DEBUG: 0x100 phases 1-2 only (rows, masks), 0x200 phases 1-3, 0x400 no
stream writes, 0x1000 stream written straight to memory (no LDS staging).  Results of a
plan made with any bit but 0x1000 set are wrong.

    python tools/flat_prepass.py [--densities 0.05,0.1,0.5]
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
    ap.add_argument("--densities", default="0.05,0.1,0.5")
    ap.add_argument("--bits", default="0,4096,256,512,1024")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    m = k = n = 4096
    os.environ["SPUTNIK_HIP_SPMM_KERNEL"] = "flat"
    for d in [float(x) for x in args.densities.split(",")]:
        ri, ro, ci, nnz = random_csr(m, k, d, dev)
        for bits in [int(x) for x in args.bits.split(",")]:
            os.environ["SPUTNIK_HIP_SPMM_DEBUG"] = str(bits)
            capi.reload_options()
            ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, nnz) + 16, dtype=torch.uint8, device=dev)
            t = timeit(lambda: capi.spmm_plan(m, k, n, ri, ro, ci, ws), iters=100, warmup=30)
            print(json.dumps(dict(density=d, debug=hex(bits), prepass_us=round(1000 * t, 2))), flush=True)
    os.environ.pop("SPUTNIK_HIP_SPMM_DEBUG", None)


if __name__ == "__main__":
    main()
