#!/usr/bin/env python3
"""BASELINE config 5 end to end: SparseLinear (2048 x 2048 weight at density 0.2) forward +
backward through the torch ops and the autograd Functions, batch 8 x seq 512 / 2048, for
float32 and half-stored activations (and half-stored weights) -- the keys
`sparse_linear_fwd_bwd_c5_*` of bench.py, on their own.

    python tools/c5_step_bench.py [--seqs 512,2048] [--iters 20]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from tools.flat_bench import timeit  # noqa: E402
from torch_sputnik_amd import SparseLinear  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seqs", default="512,2048")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--cases", default="fp32,fp16_storage,fp16_storage_and_weights,bf16_storage")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    m = n = 2048
    batch = 8
    torch.manual_seed(0)
    w = torch.randn(m, n, device=dev) * (torch.rand(m, n, device=dev) < 0.2)
    for seq in (int(s) for s in args.seqs.split(",")):
        for case in args.cases.split(","):
            xdt = {"fp32": torch.float32, "fp16_storage": torch.float16,
                   "fp16_storage_and_weights": torch.float16, "bf16_storage": torch.bfloat16}[case]
            layer = SparseLinear(n, m).to(dev)
            layer.weight = torch.nn.Parameter(w.clone())
            layer.setup_sparse_tensors()
            if case.endswith("and_weights"):
                layer.values = torch.nn.Parameter(layer.values.detach().to(xdt))
            xin = torch.randn(batch, seq, n, device=dev).to(xdt).requires_grad_(True)
            gout = torch.randn(batch, m, seq, device=dev)

            def fwd_bwd():
                layer.values.grad = None
                xin.grad = None
                layer(xin).backward(gout)

            def fwd():
                with torch.no_grad():
                    layer(xin)

            print(json.dumps(dict(case=case, seq=seq, batch=batch,
                                  fwd_bwd_ms=round(timeit(fwd_bwd, iters=args.iters, warmup=5), 4),
                                  fwd_ms=round(timeit(fwd, iters=args.iters, warmup=3), 4))), flush=True)


if __name__ == "__main__":
    main()
