#!/usr/bin/env python3
"""Developer timing: sparse softmax (forward, backward) at config 3's mask
(S = 1024, density 0.1) for several replica counts, beside a device copy of the
same bytes (what the memory system gives a perfectly streaming kernel of that
size).  GPU only."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from torch_sputnik_amd import capi
    from torch_sputnik_amd.synthetic import random_csr, uniform
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from kbench import timeit
    dev = torch.device("cuda:0")
    s = 1024
    ri, ro, ci, nnz = random_csr(s, s, 0.1, dev, seed=7)
    for reps in (64, 512):
        x = uniform((reps, nnz), dev, 1) * 8 - 4
        y = torch.empty_like(x)
        g = uniform((reps, nnz), dev, 2)
        dx = torch.empty_like(x)
        fwd, fwd_min = timeit(lambda: capi.sparse_softmax_batched(s, reps, x, ri, ro, ci, y), 100, 20)
        bwd, bwd_min = timeit(lambda: capi.sparse_softmax_backward_batched(s, reps, y, g, ro, 1.0, dx), 100, 20)
        cp, cp_min = timeit(lambda: y.copy_(x), 100, 20)
        by = reps * 8.0 * nnz
        print(json.dumps({"replicas": reps, "nnz": nnz, "fwd_us": fwd * 1e6, "fwd_min_us": fwd_min * 1e6,
                          "fwd_hbm_frac": by / fwd / 8e12, "bwd_us": bwd * 1e6,
                          "bwd_hbm_frac": reps * 12.0 * nnz / bwd / 8e12, "copy_us": cp * 1e6,
                          "copy_hbm_frac": by / cp / 8e12}), flush=True)


if __name__ == "__main__":
    main()
