#!/usr/bin/env python3
"""Config 5's training step alone (SparseLinear 2048 x 2048 at density 0.2, batch 8 x seq 512 /
2048; forward + backward through the autograd Function, plans and transposed topology cached):
one timing per invocation, for A/B runs under the SPUTNIK_HIP_* knobs and for
tools/profile_kernels.sh / prof_variants.sh.

    python tools/c5_step.py [--seq 512] [--dtype float32|float16] [--iters 30]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from bench import event_time_ms  # noqa: E402
from torch_sputnik_amd import SparseLinear  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--iters", type=int, default=30)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    m = n = 2048
    batch = 8
    torch.manual_seed(0)
    layer = SparseLinear(n, m).to(dev)
    w = torch.randn(m, n, device=dev) * (torch.rand(m, n, device=dev) < 0.2)
    layer.weight = torch.nn.Parameter(w)
    layer.setup_sparse_tensors()
    xin = torch.randn(batch, args.seq, n, device=dev).to(getattr(torch, args.dtype)).requires_grad_(True)
    gout = torch.randn(batch, m, args.seq, device=dev)

    def fwd_bwd():
        layer.values.grad = None
        xin.grad = None
        layer(xin).backward(gout)

    print(f"c5 seq {args.seq} {args.dtype} SLAB={os.environ.get('SPUTNIK_HIP_SDDMM_SLAB', 'auto')}: "
          f"{event_time_ms(fwd_bwd, args.iters):.4f} ms")


if __name__ == "__main__":
    main()
