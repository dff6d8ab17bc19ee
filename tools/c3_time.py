#!/usr/bin/env python3
"""Developer timing: SparseAttention at config 3, forward (no grad) and
forward + backward through the separate operators, bursts of back-to-back steps."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from torch_sputnik_amd import SparseAttention
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    s, emb, heads, batch = 1024, 512, 8, 8
    attn = SparseAttention(heads, emb, max_sequence_length=s, device=dev, sparsity=0.9,
                           mask_generator=np.random.default_rng(0), differentiable_softmax=True)
    for lin in attn.linears:
        w = torch.randn(emb, emb, device=dev) / 7.0 * (torch.rand(emb, emb, device=dev) < 0.1)
        lin.weight = torch.nn.Parameter(w)
        lin.setup_sparse_tensors()
    x = torch.randn(batch, s, emb, device=dev, requires_grad=True)
    gout = torch.randn(batch, s, emb, device=dev)

    def fwd():
        with torch.no_grad():
            attn(x, x, x)

    def fwd_bwd():
        x.grad = None
        for lin in attn.linears:
            lin.values.grad = None
        attn(x, x, x).backward(gout)

    def timeit(fn, bursts=7, steps=10):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(bursts):
            a = torch.cuda.Event(enable_timing=True)
            b = torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(steps):
                fn()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / steps)
        return sorted(ts)[len(ts) // 2]

    print("forward %.1f us   forward+backward %.1f us" % (timeit(fwd) * 1e3, timeit(fwd_bwd) * 1e3), flush=True)


if __name__ == "__main__":
    main()
