#!/usr/bin/env python3
"""Config 5's left_spmm alone (2048^2 weight at density 0.2 against batch 8 x seq columns):
kernel name and time of the call, for A/B runs under SPUTNIK_HIP_SPMM_KERNEL /
SPUTNIK_HIP_SPMM_MEDIUM (1: flat kernel with 8 rows per wave, 2: with 4).

    python tools/c5_spmm_bench.py [--seq 512]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from bench import event_time_ms  # noqa: E402
from torch_sputnik_amd import capi  # noqa: E402
from torch_sputnik_amd.synthetic import random_csr, uniform  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--density", type=float, default=0.2)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    m = n = 2048
    ri, ro, ci, nnz = random_csr(m, n, args.density, dev, seed=5)
    vals = uniform((nnz,), dev, 20)
    x = uniform((args.batch, n, args.seq), dev, 21)
    y = torch.empty(args.batch, m, args.seq, device=dev)
    ws = torch.empty(capi.spmm_workspace_bytes(m, n, args.seq, nnz) + 16, dtype=torch.uint8, device=dev)
    t = event_time_ms(lambda: capi.spmm_batched(m, n, args.seq, args.batch, ri, vals, 0, ro, ci, x, y, ws), 30)
    capi.spmm_plan(m, n, args.seq, ri, ro, ci, ws)
    tp = event_time_ms(lambda: capi.spmm_batched_planned(m, n, args.seq, args.batch, ri, vals, 0, ro, ci, x, y, ws), 30)
    print(f"seq {args.seq} KERNEL={os.environ.get('SPUTNIK_HIP_SPMM_KERNEL', 'auto')} "
          f"MEDIUM={os.environ.get('SPUTNIK_HIP_SPMM_MEDIUM', '0')}: "
          f"{capi.spmm_kernel_name(m, n, args.seq, nnz, args.batch)}  per call {t * 1e3:.1f} us "
          f"({2.0 * nnz * args.seq * args.batch / t / 1e9:.1f} TFLOP/s), planned {tp * 1e3:.1f} us")


if __name__ == "__main__":
    main()
