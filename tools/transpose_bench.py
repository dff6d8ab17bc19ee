#!/usr/bin/env python3
"""Developer timing: csr_transpose at config 5's weight (2048^2, density 0.2) and
two other shapes.  GPU only."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    from kbench import timeit
    from torch_sputnik_amd import capi
    from torch_sputnik_amd.synthetic import random_csr, uniform
    dev = torch.device("cuda:0")
    for m, n, d in ((2048, 2048, 0.2), (1024, 1024, 0.1), (4096, 4096, 0.1), (512, 16384, 0.05)):
        ri, ro, ci, nnz = random_csr(m, n, d, dev, seed=9)
        vals = uniform((nnz,), dev, 10)
        ov, oro = torch.empty_like(vals), torch.empty(n + 1, dtype=torch.int32, device=dev)
        oci = torch.empty(nnz, dtype=torch.int32, device=dev)
        perm = torch.empty(nnz, dtype=torch.int32, device=dev)
        ws = torch.empty(capi.csr_transpose_workspace_bytes(m, n, nnz), dtype=torch.uint8, device=dev)
        t, tmin = timeit(lambda: capi.csr_transpose(m, n, 1, vals, ro, ci, ov, oro, oci, None, ws), 100, 20)
        tp, _ = timeit(lambda: capi.csr_transpose(m, n, 1, vals, ro, ci, ov, oro, oci, perm, ws), 100, 20)
        by = 16.0 * nnz + 4.0 * (m + n + 2)
        print(json.dumps({"m": m, "n": n, "density": d, "nnz": nnz, "us": t * 1e6, "us_min": tmin * 1e6,
                          "us_with_permutation": tp * 1e6, "alg_gbs": by / t / 1e9,
                          "hbm_frac": by / t / 8e12}), flush=True)


if __name__ == "__main__":
    main()
