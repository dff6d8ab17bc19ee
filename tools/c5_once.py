#!/usr/bin/env python3
"""Profiling target: 30 SparseLinear forward + backward steps at config 5
(2048^2 weight, density 0.2, batch 8 x seq 512) for rocprofv3 --kernel-trace."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_sputnik_amd import SparseLinear  # noqa: E402

if os.environ.get("C5_PER_CALL") == "1":
    from torch_sputnik_amd import functional
    functional.enable_transpose_cache(False)
    functional.enable_plan_cache(False)
dev = torch.device("cuda:0")
torch.manual_seed(0)
n = m = 2048
batch, seq = 8, 512
layer = SparseLinear(n, m).to(dev)
layer.weight = torch.nn.Parameter(torch.randn(m, n, device=dev) * (torch.rand(m, n, device=dev) < 0.2))
layer.setup_sparse_tensors()
dt = torch.float16 if len(sys.argv) > 1 and sys.argv[1] == "fp16" else torch.float32
if dt == torch.float16:
    layer.values = torch.nn.Parameter(layer.values.detach().half())
x = torch.randn(batch, seq, n, device=dev).to(dt).requires_grad_(True)
gout = torch.randn(batch, m, seq, device=dev)
for _ in range(30):
    layer.values.grad = None
    x.grad = None
    layer(x).backward(gout)
torch.cuda.synchronize()
