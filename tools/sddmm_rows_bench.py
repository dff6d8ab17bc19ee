#!/usr/bin/env python3
"""The rhs-stationary SDDMM kernel with a workgroup per (slab, row block) against workgroups that
walk several row blocks of one staged slab (round 5, csrc/sddmm_tiled.hip launch_rows_y):
SPUTNIK_HIP_SDDMM_DEBUG bits 20.. = the target workgroup count in hundreds (1023: every row
block its own workgroup, the form of rounds 1-4).  Summed (weight gradient) and plain products.

    python tools/sddmm_rows_bench.py
"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_sputnik_amd import capi
from torch_sputnik_amd.synthetic import random_csr, uniform
from tools.flat_bench import timeit
dev = torch.device("cuda:0")
for (m, k, reps, d, summed) in ((2048, 512, 8, 0.2, True), (2048, 512, 8, 0.05, True), (4096, 512, 4, 0.1, True),
                                (1024, 1024, 8, 0.3, True), (512, 1024, 8, 0.1, True), (2048, 2048, 8, 0.2, True),
                                (2048, 512, 8, 0.2, False), (4096, 256, 4, 0.1, False), (2048, 128, 16, 0.2, False),
                                # the quad kernel (k = 64)
                                (2048, 64, 64, 0.1, False), (4096, 64, 16, 0.05, False), (2048, 64, 32, 0.3, False)):
    ri, ro, ci, nnz = random_csr(m, m, d, dev, seed=3)
    lhs = uniform((reps, m, k), dev, 4); rhs = uniform((reps, m, k), dev, 5)
    row = dict(m=m, k=k, replicas=reps, density=d, summed=summed)
    for name, dbg in (("all_row_blocks", 1023 << 20), ("default_768", 0), ("t512", 5 << 20), ("t1536", 15 << 20)):
        os.environ["SPUTNIK_HIP_SDDMM_DEBUG"] = str(dbg); capi.reload_options()
        if summed:
            out = torch.empty(nnz, device=dev)
            ws = torch.empty(capi.sddmm_sum_workspace_bytes(m, k, m, nnz) + 16, dtype=torch.uint8, device=dev)
            scr = torch.empty(capi.sddmm_sum_scratch_bytes(m, k, m, nnz, reps) + 16, dtype=torch.uint8, device=dev)
            capi.sddmm_sum_plan(m, k, m, ri, ro, ci, ws)
            t = timeit(lambda: capi.sddmm_sum_batched(m, k, m, reps, ri, ro, ci, lhs, rhs, out, ws, scr, planned=True), iters=30)
        else:
            out = torch.empty(reps, nnz, device=dev)
            ws = torch.empty(capi.sddmm_workspace_bytes(m, k, m, nnz) + 16, dtype=torch.uint8, device=dev)
            os.environ["SPUTNIK_HIP_SDDMM_KERNEL"] = "tiled"; capi.reload_options()
            t = timeit(lambda: capi.sddmm_batched(m, k, m, reps, ri, ro, ci, lhs, rhs, out, ws), iters=30)
            os.environ.pop("SPUTNIK_HIP_SDDMM_KERNEL")
        row[name + "_us"] = round(1000 * t, 1)
    print(json.dumps(row), flush=True)
