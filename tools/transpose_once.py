#!/usr/bin/env python3
"""Profiling target: 50 csr_transpose calls at config 5's weight (for rocprofv3 --kernel-trace)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_sputnik_amd import capi  # noqa: E402
from torch_sputnik_amd.synthetic import random_csr, uniform  # noqa: E402

dev = torch.device("cuda:0")
m = n = 2048
ri, ro, ci, nnz = random_csr(m, n, 0.2, dev, seed=9)
vals = uniform((nnz,), dev, 10)
ov, oro = torch.empty_like(vals), torch.empty(n + 1, dtype=torch.int32, device=dev)
oci = torch.empty(nnz, dtype=torch.int32, device=dev)
ws = torch.empty(capi.csr_transpose_workspace_bytes(m, n, nnz), dtype=torch.uint8, device=dev)
for _ in range(50):
    capi.csr_transpose(m, n, 1, vals, ro, ci, ov, oro, oci, None, ws)
torch.cuda.synchronize()
