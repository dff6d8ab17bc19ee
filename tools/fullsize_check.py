#!/usr/bin/env python3
"""Whole-matrix check of the SpMM kernels at BASELINE.json's full size against a
dense fp32 product on the GPU (the parity tests compare full outputs only at
sizes the CPU oracle finishes in seconds), plus a look at the chunk table the
pre-pass left in the workspace.  Developer tool.

    python tools/fullsize_check.py [--size 4096] [--densities 0.5,0.1,0.05]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from torch_sputnik_amd import capi  # noqa: E402
from torch_sputnik_amd.synthetic import random_csr, uniform  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--densities", default="0.5,0.1,0.05")
    ap.add_argument("--bk", type=int, default=0, help="chunk rows of the table to inspect (0: skip)")
    ap.add_argument("--seed", type=int, default=-1, help="topology seed (default: derived from the density)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    m = k = n = args.size
    bad = 0
    for d in [float(x) for x in args.densities.split(",")]:
        ri, ro, ci, nnz = random_csr(m, k, d, dev, seed=args.seed if args.seed >= 0 else int(d * 1000) + 11)
        vals = uniform((nnz,), dev, 1) - 0.5
        b = uniform((k, n), dev, 2) - 0.5
        rows = torch.repeat_interleave(torch.arange(m, device=dev), (ro[1:] - ro[:-1]).long())
        a = torch.zeros(m, k, device=dev)
        a[rows, ci.long()] = vals
        ref = a @ b
        ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, nnz) + 16, dtype=torch.uint8, device=dev)
        out = torch.full((m, n), float("nan"), device=dev)
        capi.spmm_batched(m, k, n, 1, ri, vals, 0, ro, ci, b, out, ws)
        torch.cuda.synchronize()
        err = (out - ref).abs().amax(dim=1) / ref.abs().amax()
        worst = int(err.argmax())
        nbad = int((err > 1e-4).sum()) + int(torch.isnan(out).any())
        bad += nbad
        print(f"density {d}: max rel err {float(err.max()):.2e} (row {worst}, "
              f"{int(ro[worst + 1] - ro[worst])} nonzeros), rows off by > 1e-4: {nbad}", flush=True)
        if args.bk:
            slots = (m + 255) // 256 * 256
            nchunks = (k + args.bk - 1) // args.bk
            skip = (slots * 4 + 255) // 256 * 256
            t = ws[skip:skip + 4 * (nchunks + 1) * slots].view(torch.int32).view(nchunks + 1, slots)
            dec = (t[1:] < t[:-1])
            print(f"  table ({args.bk}-row chunks): decreasing steps: {int(dec.sum())}", flush=True)
            if dec.any():
                c, s = [int(x[0]) for x in torch.nonzero(dec, as_tuple=True)]
                print(f"  first: chunk {c} slot {s}: {int(t[c, s])} -> {int(t[c + 1, s])}", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
