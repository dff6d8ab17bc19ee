#!/usr/bin/env python3
"""Config 3's SDDMM (1024^2 mask at density 0.1, k = 64, 64 replicas) on a planned
workspace: the pair-flat kernel (csrc/sddmm_flat.hip) against the rhs-stationary quad
kernel (SPUTNIK_HIP_SDDMM_FLAT=0), float32 / float16 / bfloat16 operands, and the
pair-flat kernel with parts switched off (SPUTNIK_HIP_SDDMM_DEBUG bits 8.. : 0x100 no
arithmetic, 0x200 no slab copies after the first two, 0x400 no stores, 0x800 no work loop
at all, 0x1000 no prologue copies).

    python tools/sddmm_flat_bench.py [--replicas 64] [--seq 1024]
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
    ap.add_argument("--replicas", type=int, default=64)
    ap.add_argument("--seq", type=int, default=1024)
    ap.add_argument("--density", type=float, default=0.1)
    ap.add_argument("--variants", default="1:0,0:0,1:0x100,1:0x200,1:0x400,1:0x700,1:0x800,1:0x1800,1:0x1a00",
                    help="flat:debug pairs (host timing is launch-bound below ~16 us: run ONE variant "
                         "under rocprofv3 --kernel-trace --stats for the device-side time)")
    ap.add_argument("--types", default="f32,f16,bf16")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    s, d, reps = args.seq, 64, args.replicas
    ri, ro, ci, nnz = random_csr(s, s, args.density, dev, seed=3)
    q = uniform((reps, s, d), dev, 4)
    k = uniform((reps, s, d), dev, 5)
    variants = [(int(a), int(b, 0)) for a, b in (v.split(":") for v in args.variants.split(","))]
    for name, dt in (("f32", torch.float32), ("f16", torch.float16), ("bf16", torch.bfloat16)):
        if name not in args.types.split(","):
            continue
        qq, kk = q.to(dt), k.to(dt)
        out = torch.empty(reps, nnz, device=dev)
        for flat, dbg in variants:
            os.environ["SPUTNIK_HIP_SDDMM_FLAT"] = str(flat)
            os.environ["SPUTNIK_HIP_SDDMM_DEBUG"] = str(dbg)
            capi.reload_options()
            # (the plan depends on the geometry the knob selects)
            ws = torch.empty(capi.sddmm_workspace_bytes(s, d, s, nnz) + 16, dtype=torch.uint8, device=dev)
            capi.sddmm_plan(s, d, s, ri, ro, ci, ws)
            t = timeit(lambda: capi.sddmm_typed(s, d, s, reps, ri, ro, ci, qq, kk, out, ws, planned=True),
                       iters=50, warmup=10)
            print(json.dumps(dict(type=name, kernel=capi.sddmm_kernel_name(s, d, s, nnz, reps, qq.element_size(), True),
                                  debug=hex(dbg), us=round(1000 * t, 2),
                                  tflops=round(2.0 * nnz * d * reps / t / 1e9, 2))), flush=True)
    os.environ.pop("SPUTNIK_HIP_SDDMM_FLAT", None)
    os.environ.pop("SPUTNIK_HIP_SDDMM_DEBUG", None)


if __name__ == "__main__":
    main()
