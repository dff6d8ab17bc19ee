#!/usr/bin/env python3
"""Config 5 (SparseLinear forward + backward, 2048 x 2048 weight at density 0.2)
broken down by operator (dev tool).  batch and seq are not given by BASELINE.json;
defaults 8 x 512."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch_sputnik  # noqa: E402
from torch_sputnik_amd.synthetic import random_csr  # noqa: E402
from torch_sputnik_amd.topology import diffsort  # noqa: E402


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--features", type=int, default=2048)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--density", type=float, default=0.2)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    f, b, s = a.features, a.batch, a.seq
    ri, ro, ci, nnz = random_csr(f, f, a.density, dev, seed=5)
    w = torch.rand(nnz, device=dev)
    x = torch.rand(b, f, s, device=dev)        # [B, in, seq] as left_spmm sees it
    gy = torch.rand(b, f, s, device=dev)       # grad of [B, out, seq]
    rows = {}
    rows["fwd left_spmm"] = (timeit(lambda: torch_sputnik.left_spmm(f, f, w, ri, ro, ci, x)),
                             2.0 * nnz * s * b)
    # backward, modules/sparse_linear.py:40-65: grad_values = sum_b sddmm(grad_out_b, x_b)
    rows["bwd sddmm (k=seq)"] = (timeit(lambda: torch_sputnik.sddmm(f, f, ri, ro, ci, gy, x)),
                                 2.0 * nnz * s * b)
    rows["bwd csr_transpose"] = (timeit(lambda: torch_sputnik.csr_transpose(f, f, w, ro, ci)), 0.0)
    wt, rot, cit = torch_sputnik.csr_transpose(f, f, w, ro, ci)
    rit = diffsort(rot)
    rows["bwd left_spmm (transposed)"] = (
        timeit(lambda: torch_sputnik.left_spmm(f, f, wt, rit, rot, cit, gy)), 2.0 * nnz * s * b)
    for k, (ms, fl) in rows.items():
        print(json.dumps(dict(op=k, ms=ms, tflops=fl / ms / 1e9 if fl else None)), flush=True)


if __name__ == "__main__":
    main()
