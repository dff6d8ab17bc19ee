#!/usr/bin/env python3
"""The three products of a sparse layer on half-stored activations, one by one (config 5:
2048 x 2048 weight at density 0.2, batch 8 x seq 512 / 2048): csrc/sparse_linear_half.hip on
csrc/mfma_gemm.h, by tile rows (SPUTNIK_HIP_MFMA_TILE) and with parts of the loop switched
off (SPUTNIK_HIP_MFMA_DEBUG: 1 no MFMAs, 2 no copies, 4 no fragment reads).

    python tools/half_linear_bench.py [--seq 512] [--tiles 0,128,256] [--debug 0,1,2,4,6,7]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from torch_sputnik_amd import capi, ops  # noqa: E402
from torch_sputnik_amd.synthetic import random_csr, uniform  # noqa: E402
from tools.flat_bench import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--tiles", default="0")
    ap.add_argument("--debug", default="0")
    ap.add_argument("--values", default="half,float32")
    ap.add_argument("--grads", default="float32")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    m = k = 2048
    batch, seq = 8, args.seq
    ri, ro, ci, nnz = random_csr(m, k, 0.2, dev, seed=9)
    vals = uniform((nnz,), dev, 10) - 0.5
    x = (uniform((batch, seq, k), dev, 21) - 0.5).half()
    gy = uniform((batch, m, seq), dev, 22) - 0.5
    dense = 2.0 * m * k * seq * batch
    for tile in args.tiles.split(","):
        for dbg in args.debug.split(","):
            os.environ["SPUTNIK_HIP_MFMA_TILE"] = tile
            os.environ["SPUTNIK_HIP_MFMA_DEBUG"] = dbg
            capi.reload_options()
            for vk in args.values.split(","):
                v = vals.half() if vk == "half" else vals
                image = ops.half_linear_image(m, k, v, ro, ci, torch.float16)
                plan = ops.half_linear_plan(m, k, ro, ci)
                row = dict(seq=seq, tile=tile, debug=dbg, values=vk)
                t = timeit(lambda: ops.half_linear_image(m, k, v, ro, ci, torch.float16), iters=20, warmup=3)
                row["image_us"] = round(1000 * t, 1)
                t = timeit(lambda: ops.half_linear_forward(m, image, v.dtype, x), iters=20, warmup=3)
                row["forward_us"] = round(1000 * t, 1)
                row["forward_dense_tflops"] = round(dense * (2 if vk == "float32" else 1) / t / 1e9)
                for gk in args.grads.split(","):
                    planes = gk == "float32"
                    g = ops.half_planes(gy, torch.float16) if planes else gy.half()
                    if planes:
                        t = timeit(lambda: ops.half_planes(gy, torch.float16), iters=20, warmup=3)
                        row["planes_us"] = round(1000 * t, 1)
                    t = timeit(lambda: ops.half_linear_weight_gradient(m, ro, ci, g, planes, x, plan), iters=20, warmup=3)
                    row[f"wgrad_{gk}_us"] = round(1000 * t, 1)
                    t = timeit(lambda: ops.half_linear_input_gradient(m, k, g, planes, image, v.dtype, x, batch, seq),
                               iters=20, warmup=3)
                    row[f"dx_{gk}_us"] = round(1000 * t, 1)
                print(json.dumps(row), flush=True)
    os.environ.pop("SPUTNIK_HIP_MFMA_TILE", None)
    os.environ.pop("SPUTNIK_HIP_MFMA_DEBUG", None)
    capi.reload_options()


if __name__ == "__main__":
    main()
