#!/usr/bin/env python3
"""Profiling target: 20 SparseAttention forward + backward steps at config 3
(S 1024, 8 heads x batch 8, head_dim 64, mask and projections at density 0.1)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_sputnik_amd import SparseAttention  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
s, emb, heads, batch = 1024, 512, 8, 8
mode = sys.argv[1] if len(sys.argv) > 1 else "separate"
attn = SparseAttention(heads, emb, max_sequence_length=s, device=dev, sparsity=0.9,
                       mask_generator=np.random.default_rng(0),
                       differentiable_softmax=(mode == "separate"), fused_training=(mode == "fused"))
for lin in attn.linears:
    w = torch.randn(emb, emb, device=dev) / 7.0 * (torch.rand(emb, emb, device=dev) < 0.1)
    lin.weight = torch.nn.Parameter(w)
    lin.setup_sparse_tensors()
x = torch.randn(batch, s, emb, device=dev, requires_grad=True)
gout = torch.randn(batch, s, emb, device=dev)
steps = 20
for _ in range(steps):
    x.grad = None
    for lin in attn.linears:
        lin.values.grad = None
    if mode == "forward":
        with torch.no_grad():
            attn(x, x, x)
    else:
        attn(x, x, x).backward(gout)
torch.cuda.synchronize()
