#!/usr/bin/env python3
"""Where the LDS-tiled SDDMM (pre-pass + stationary kernel) overtakes the row-wave kernel
(one launch): both forced through SPUTNIK_HIP_SDDMM_KERNEL, per-call form (the pre-pass
inside the call), and what the automatic rule picks (csrc/sddmm.hip, takes_tiled).

    python tools/small_sddmm.py [--summed] [--half]      (--half: float16 operands; --summed: the form summed over the replicas,
                                                 the weight gradient of a layer: sddmm_sum_batched)
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from torch_sputnik_amd import capi  # noqa: E402
from torch_sputnik_amd.synthetic import random_csr, uniform  # noqa: E402
from tools.flat_bench import timeit  # noqa: E402

SHAPES = ((256, 64, 1), (1024, 64, 1), (1024, 64, 8), (1024, 64, 16), (1024, 64, 32), (1024, 64, 64),
          (512, 64, 64), (2048, 64, 8), (2048, 64, 16), (1024, 128, 16), (1024, 128, 64), (512, 512, 1),
          (2048, 512, 1), (2048, 512, 8), (1024, 256, 8))


SUMMED_SHAPES = ((256, 256, 8), (512, 64, 8), (512, 256, 8), (512, 256, 32), (512, 1024, 1), (512, 1024, 8),
                 (1024, 64, 8), (1024, 256, 8), (1024, 512, 4), (1024, 512, 16), (2048, 128, 8), (2048, 512, 2),
                 (2048, 512, 8), (2048, 1024, 1))


def main():
    dev = torch.device("cuda:0")
    summed = "--summed" in sys.argv
    half = "--half" in sys.argv     # float16 operands, float32 output: sputnik_hip_sddmm_typed
    for (sz, k, reps) in (SUMMED_SHAPES if summed else SHAPES):
        for d in (0.5, 0.1, 0.05, 0.02):
            ri, ro, ci, nnz = random_csr(sz, sz, d, dev, seed=3)
            if nnz < 4 * sz:
                continue
            lhs = uniform((reps, sz, k), dev, 4)
            rhs = uniform((reps, sz, k), dev, 5)
            if half:
                lhs, rhs = lhs.half(), rhs.half()
            out = torch.empty(nnz if summed else (reps, nnz), device=dev)
            ws = torch.empty((capi.sddmm_sum_workspace_bytes if summed else capi.sddmm_workspace_bytes)(
                sz, k, sz, nnz) + 16, dtype=torch.uint8, device=dev)
            row = {"m": sz, "k": k, "replicas": reps, "density": d, "nnz": nnz,
                   "nnz_k2_r_log2": round(float(torch.log2(torch.tensor(float(nnz) * k * k * reps))), 2)}
            for name in ("tiled", "wave", ""):
                if name:
                    os.environ["SPUTNIK_HIP_SDDMM_KERNEL"] = name
                else:
                    os.environ.pop("SPUTNIK_HIP_SDDMM_KERNEL", None)
                capi.reload_options()
                if summed:
                    scr = torch.empty(capi.sddmm_sum_scratch_bytes(sz, k, sz, nnz, reps) + 16, dtype=torch.uint8,
                                      device=dev)
                    t = timeit(lambda: capi.sddmm_sum_batched(sz, k, sz, reps, ri, ro, ci, lhs, rhs, out, ws,
                                                              scr), iters=40)
                elif half:
                    t = timeit(lambda: capi.sddmm_typed(sz, k, sz, reps, ri, ro, ci, lhs, rhs, out, ws), iters=40)
                else:
                    t = timeit(lambda: capi.sddmm_batched(sz, k, sz, reps, ri, ro, ci, lhs, rhs, out, ws),
                               iters=40)
                row[(name or "auto") + "_us"] = round(1000 * t, 1)
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
