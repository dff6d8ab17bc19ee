#!/usr/bin/env python3
"""The many-mask operators at the attention size (bench.py's other_ops.many_mask_*):
b masks x heads, S x S, head_dim 64, a density per mask.

    python tools/many_mask_bench.py [--densities 0.1,0.2,0.05,0.5] [--b 8] [--heads 8]
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
    ap.add_argument("--densities", default="0.1,0.2,0.05,0.5")
    ap.add_argument("--b", type=int, default=8)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--s", type=int, default=1024)
    ap.add_argument("--d", type=int, default=64)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    s, d, b, h = args.s, args.d, args.b, args.heads
    dens = [float(x) for x in args.densities.split(",")]
    topo = [random_csr(s, s, dens[i % len(dens)], dev, seed=70 + i) for i in range(b)]
    nn = torch.tensor([t4[3] for t4 in topo], dtype=torch.int32)
    width = int(nn.max())
    mri = torch.cat([t4[0] for t4 in topo])
    mro = torch.cat([t4[1] for t4 in topo])
    mci = torch.cat([t4[2] for t4 in topo])
    r = b * h
    q = uniform((r, s, d), dev, 1)
    k = uniform((r, s, d), dev, 2)
    v = uniform((r, s, d), dev, 3)
    ctx = torch.empty(r, s, d, device=dev)
    scores = torch.zeros(r, width, device=dev)
    probs = torch.zeros_like(scores)
    ws = torch.empty(max(capi.sddmm_many_mask_workspace_bytes(b, s, d, s, width),
                         capi.spmm_workspace_bytes(s, s, d, width)) + 16, dtype=torch.uint8, device=dev)
    total = float(nn.sum()) * h
    row = {"densities": dens, "b": b, "heads": h, "entries_x_heads": total}
    # (SPUTNIK_HIP_SDDMM_DEBUG=64: the masks in their own order instead of largest first)
    os.environ["SPUTNIK_HIP_SDDMM_DEBUG"] = "64"
    capi.reload_options()
    t = timeit(lambda: capi.sddmm_many_mask(b, s, d, s, nn, r, mri, mro, mci, q, k, scores, ws), iters=40)
    row["sddmm_mask_order_us"] = round(1000 * t, 1)
    os.environ.pop("SPUTNIK_HIP_SDDMM_DEBUG")
    capi.reload_options()
    t = timeit(lambda: capi.sddmm_many_mask(b, s, d, s, nn, r, mri, mro, mci, q, k, scores, ws), iters=40)
    row["sddmm_us"] = round(1000 * t, 1)
    row["sddmm_tflops"] = round(2.0 * total * d / t / 1e9, 2)
    t = timeit(lambda: capi.sparse_softmax_many_mask(b, s, nn, r, scores, mri, mro, mci, d ** -0.5, probs), iters=40)
    row["softmax_us"] = round(1000 * t, 1)
    t = timeit(lambda: capi.spmm_many_mask(b, s, s, d, nn, r, mri, probs, mro, mci, v, ctx, ws), iters=40)
    row["spmm_us"] = round(1000 * t, 1)
    row["spmm_tflops"] = round(2.0 * total * d / t / 1e9, 2)
    print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
