#!/usr/bin/env python3
"""The chunk-table pre-pass alone (sputnik_hip_spmm_plan) on 4096^2 at density 0.1 against 72
columns, for rocprofv3 --kernel-trace --stats (tools/profile_kernels.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_sputnik_amd import capi  # noqa: E402
from torch_sputnik_amd.synthetic import random_csr  # noqa: E402

dev = torch.device("cuda:0")
for (m, n, d) in ((4096, 72, 0.1), (1024, 64, 0.1), (2048, 512, 0.2)):
    ri, ro, ci, nnz = random_csr(m, m, d, dev, seed=21)
    ws = torch.empty(capi.spmm_workspace_bytes(m, m, n, nnz) + 16, dtype=torch.uint8, device=dev)
    for _ in range(50):
        capi.spmm_plan(m, m, n, ri, ro, ci, ws)
    torch.cuda.synchronize()
