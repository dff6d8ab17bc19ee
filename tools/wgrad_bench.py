#!/usr/bin/env python3
"""The weight gradient of a sparse layer shared by a batch (modules/sparse_linear.py:44-49:
the summed SDDMM) on half-storage operands: the matrix-core route (csrc/sddmm_mfma.hip)
against the vector kernels (SPUTNIK_HIP_SDDMM_KERNEL=tiled), and float32 for reference.

    python tools/wgrad_bench.py [--m 2048] [--seqs 512,2048] [--densities 0.2] [--replicas 8]

One JSON line per (density, seq, type, route): whole C-ABI call (kernel + the sum of the
partial vectors), median of 30 HIP-event pairs.
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
    ap.add_argument("--m", type=int, default=2048)
    ap.add_argument("--seqs", default="512,2048")
    ap.add_argument("--densities", default="0.2")
    ap.add_argument("--replicas", type=int, default=8)
    ap.add_argument("--types", default="f16,bf16,f32,f32xf16,f32xbf16")
    ap.add_argument("--routes", default="auto,tiled")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    m = n = args.m
    reps = args.replicas
    for density in (float(d) for d in args.densities.split(",")):
        ri, ro, ci, nnz = random_csr(m, n, density, dev, seed=9)
        for seq in (int(s) for s in args.seqs.split(",")):
            gy = uniform((reps, m, seq), dev, 22) - 0.5
            x = uniform((reps, n, seq), dev, 21) - 0.5
            for name, dt in (("f16", torch.float16), ("bf16", torch.bfloat16), ("f32", torch.float32),
                             ("f32xf16", torch.float16), ("f32xbf16", torch.bfloat16)):
                if name not in args.types.split(","):
                    continue
                mixed = name.startswith("f32x")
                a, b = (gy if mixed else gy.to(dt)), x.to(dt)
                out = torch.empty(nnz, device=dev)
                for route in (["auto"] if mixed else args.routes.split(",")):
                    if route == "auto":
                        os.environ.pop("SPUTNIK_HIP_SDDMM_KERNEL", None)
                    else:
                        os.environ["SPUTNIK_HIP_SDDMM_KERNEL"] = route
                    capi.reload_options()
                    ws = torch.empty(capi.sddmm_sum_workspace_bytes(m, seq, n, nnz) + 16, dtype=torch.uint8,
                                     device=dev)
                    scratch = torch.empty((capi.sddmm_sum_mixed_scratch_bytes(m, seq, n, nnz, reps, a, b)
                                           if mixed else capi.sddmm_sum_scratch_bytes(m, seq, n, nnz, reps)) + 16,
                                          dtype=torch.uint8, device=dev)
                    capi.sddmm_sum_plan(m, seq, n, ri, ro, ci, ws)
                    call = capi.sddmm_sum_mixed if mixed else capi.sddmm_sum_typed
                    t = timeit(lambda: call(m, seq, n, reps, ri, ro, ci, a, b, out, ws, scratch,
                                            planned=True), iters=30, warmup=5)
                    print(json.dumps(dict(m=m, density=density, seq=seq, replicas=reps, type=name, route=route,
                                          us=round(1000 * t, 1),
                                          sampled_tflops=round(2.0 * nnz * seq * reps / t / 1e9, 1),
                                          dense_tflops=round(2.0 * m * n * seq * reps / t / 1e9, 1))),
                          flush=True)
    os.environ.pop("SPUTNIK_HIP_SDDMM_KERNEL", None)
    capi.reload_options()


if __name__ == "__main__":
    main()
