#!/usr/bin/env python3
"""left_spmm of a layer weight (2048 x 2048, shared by the batch) against [8, 2048, seq] on
half-storage operands: the matrix-core route (csrc/spmm_mfma.hip) for every pairing of
float32 / half values and dense operand, against the vector kernels on the same operands
(SPUTNIK_HIP_SPMM_KERNEL=wide forces them) and the float32 product.

    python tools/spmm_mfma_bench.py [--seqs 512,2048] [--densities 0.2]
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
    ap.add_argument("--tiles", default="f16,bf16")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    m = k = args.m
    reps = args.replicas
    for density in (float(d) for d in args.densities.split(",")):
        ri, ro, ci, nnz = random_csr(m, k, density, dev, seed=9)
        vals = uniform((nnz,), dev, 10) - 0.5
        for seq in (int(s) for s in args.seqs.split(",")):
            x = uniform((reps, k, seq), dev, 21) - 0.5
            out = torch.empty(reps, m, seq, device=dev)

            def line(name, t):
                print(json.dumps(dict(m=m, density=density, seq=seq, replicas=reps, route=name,
                                      us=round(1000 * t, 1),
                                      sparse_tflops=round(2.0 * nnz * seq * reps / t / 1e9, 1))), flush=True)

            ws = torch.empty(capi.spmm_workspace_bytes(m, k, seq, nnz) + 16, dtype=torch.uint8, device=dev)
            line("float32 vector kernels", timeit(
                lambda: capi.spmm_batched(m, k, seq, reps, ri, vals, 0, ro, ci, x, out, ws), iters=20, warmup=5))
            for tname in args.tiles.split(","):
                tile = {"f16": torch.float16, "bf16": torch.bfloat16}[tname]
                for vk, v in (("half", vals.to(tile)), ("float32", vals)):
                    for dk, d in (("half", x.to(tile)), ("float32", x)):
                        need = capi.left_spmm_half_tiles_workspace_bytes(m, k, seq, nnz, reps, v, d, tile)
                        if need == 0:
                            continue
                        wst = torch.empty(need, dtype=torch.uint8, device=dev)
                        line(f"tiles {tname}: values {vk}, dense {dk}", timeit(
                            lambda: capi.left_spmm_half_tiles(m, k, seq, reps, ro, ci, v, d, tile, out, wst),
                            iters=20, warmup=5))
                # the typed call on half operands with the matrix cores switched off
                os.environ["SPUTNIK_HIP_SPMM_KERNEL"] = "wide"
                capi.reload_options()
                v, d = vals, x.to(tile)
                wsv = torch.empty(capi.spmm_typed_workspace_bytes(m, k, seq, nnz, reps, v, 0, d) + 256,
                                  dtype=torch.uint8, device=dev)
                try:
                    line(f"vector kernels {tname}: values float32, dense half (knob: wide)", timeit(
                        lambda: capi.spmm_typed(m, k, seq, reps, ri, v, 0, ro, ci, d, out, wsv), iters=20, warmup=5))
                finally:
                    os.environ.pop("SPUTNIK_HIP_SPMM_KERNEL", None)
                    capi.reload_options()


if __name__ == "__main__":
    main()
