#!/usr/bin/env python3
"""Config 3's attention SpMM (1024^2 mask at density 0.1 times [1024, 64], 64 replicas:
the two-panel panel-resident kernel).  SPUTNIK_HIP_SPMM_DEBUG=16: rows are never cut at
the panel boundary (the masked walk that serves any column order, round 2-3's only form).
Host timing; run ONE variant under rocprofv3 --kernel-trace --stats (tools/prof_variants.sh)
for device-side times.

    python tools/spmm_c3_bench.py [--variants 0,16] [--permuted]

--permuted: the TRANSPOSED product of the backward pass (dV = P^T dO, dK = dS^T Q) with the
values permuted by a pass of their own (LDS-banded permutation, then the product) against the
kernel gathering them through the cached permutation (round 5: two panels with cut rows).
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
    ap.add_argument("--variants", default="0,16")
    ap.add_argument("--replicas", type=int, default=64)
    ap.add_argument("--seq", type=int, default=1024)
    ap.add_argument("--density", type=float, default=0.1)
    ap.add_argument("--permuted", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    s, d, reps = args.seq, 64, args.replicas
    ri, ro, ci, nnz = random_csr(s, s, args.density, dev, seed=3)
    p = uniform((reps, nnz), dev, 4)
    v = uniform((reps, s, d), dev, 5)
    out = torch.empty(reps, s, d, device=dev)
    ws = torch.empty(capi.spmm_workspace_bytes(s, s, d, nnz) + 16, dtype=torch.uint8, device=dev)
    for dbg in [int(x, 0) for x in args.variants.split(",")]:
        os.environ["SPUTNIK_HIP_SPMM_DEBUG"] = str(dbg)
        capi.reload_options()
        t = timeit(lambda: capi.spmm_batched(s, s, d, reps, ri, p, nnz, ro, ci, v, out, ws), iters=50, warmup=10)
        print(json.dumps(dict(kernel=capi.spmm_kernel_name(s, s, d, nnz, reps), debug=dbg, us=round(1000 * t, 2),
                              tflops=round(2.0 * nnz * d * reps / t / 1e9, 2))), flush=True)
    os.environ.pop("SPUTNIK_HIP_SPMM_DEBUG", None)
    capi.reload_options()
    if args.permuted:
        from torch_sputnik_amd import ops
        from torch_sputnik_amd.topology import diffsort
        _, rot, cit, perm = ops.csr_transpose_with_permutation(s, s, p[0].contiguous(), ro, ci)
        rit = diffsort(rot)
        lists = ops.banded_lists(perm)

        def separate():
            return ops.spmm_transposed_out(s, s, ops.permute_last_banded(p, *lists), rit, rot, cit, v, d)

        def fused():
            return ops.spmm_transposed_out(s, s, p, rit, rot, cit, v, d, permutation=perm)

        print(json.dumps(dict(permute_then_product_us=round(1000 * timeit(separate, iters=40), 1),
                              gathered_in_the_kernel_us=round(1000 * timeit(fused, iters=40), 1),
                              bit_identical=bool(torch.equal(separate(), fused())),
                              fused_by_rule=bool(ops.spmm_permuted_fused(s, s, d, nnz)))), flush=True)


if __name__ == "__main__":
    main()
