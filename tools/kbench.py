#!/usr/bin/env python3
"""Kernel-level timing of the five ops at BASELINE.json's configs (dev tool;
bench.py is the contract benchmark).  Calls the C ABI directly.

    python tools/kbench.py [--ops spmm,sddmm,softmax,transpose] [--iters 50]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from torch_sputnik_amd import capi  # noqa: E402


def gpu_csr(m, n, density, dev, seed=0, round_to=4):
    from torch_sputnik_amd.synthetic import random_csr
    return random_csr(m, n, density, dev, seed=seed, round_to=round_to)


def timeit(fn, iters, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    for i in range(iters):
        starts[i].record()
        fn()
        ends[i].record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in zip(starts, ends))
    return ts[len(ts) // 2] * 1e-3, ts[0] * 1e-3


def bench_spmm(dev, iters, densities, m=4096, k=4096, n=4096, replicas=1):
    out_rows = []
    for d in densities:
        ri, ro, ci, nnz = gpu_csr(m, k, d, dev, seed=int(d * 1000))
        vals = torch.rand(replicas, nnz, device=dev) if replicas > 1 else torch.rand(nnz, device=dev)
        b = torch.rand((replicas, k, n) if replicas > 1 else (k, n), device=dev)
        out = torch.empty((replicas, m, n) if replicas > 1 else (m, n), device=dev)
        ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, nnz) + 16, dtype=torch.uint8, device=dev)
        fn = lambda: capi.spmm_batched(m, k, n, replicas, ri, vals, nnz if replicas > 1 else 0,
                                       ro, ci, b, out, ws)
        med, best = timeit(fn, iters)
        flops = 2.0 * nnz * n * replicas
        bytes_ = (8.0 * nnz + 4 * k * n + 4 * m * n) * replicas + 4 * (2 * m + 1)
        out_rows.append(dict(op="spmm", m=m, k=k, n=n, density=d, replicas=replicas, nnz=nnz,
                             ms=med * 1e3, ms_min=best * 1e3, gflops=flops / med / 1e9,
                             alg_gbs=bytes_ / med / 1e9, hbm_frac=bytes_ / med / 8e12,
                             fma_frac=flops / med / 157.3e12))
        print(json.dumps(out_rows[-1]), flush=True)
    return out_rows


def bench_attention_ops(dev, iters, s=1024, d=64, replicas=64, density=0.1):
    ri, ro, ci, nnz = gpu_csr(s, s, density, dev, seed=7)
    q = torch.rand(replicas, s, d, device=dev)
    kk = torch.rand(replicas, s, d, device=dev)
    v = torch.rand(replicas, s, d, device=dev)
    scores = torch.empty(replicas, nnz, device=dev)
    sd_ws = torch.empty(capi.sddmm_workspace_bytes(s, d, s, nnz) + 16, dtype=torch.uint8, device=dev)
    probs = torch.empty(replicas, nnz, device=dev)
    ctx = torch.empty(replicas, s, d, device=dev)
    rows = []
    med, best = timeit(lambda: capi.sddmm_batched(s, d, s, replicas, ri, ro, ci, q, kk, scores, sd_ws), iters)
    by = replicas * (4.0 * (2 * s * d) + 4 * nnz) + 4 * nnz + 4 * (2 * s + 1)
    rows.append(dict(op="sddmm", s=s, d=d, replicas=replicas, nnz=nnz, ms=med * 1e3, ms_min=best * 1e3,
                     gflops=2.0 * nnz * d * replicas / med / 1e9, alg_gbs=by / med / 1e9,
                     hbm_frac=by / med / 8e12))
    print(json.dumps(rows[-1]), flush=True)
    med, best = timeit(lambda: capi.sparse_softmax_batched(s, replicas, scores, ri, ro, ci, probs), iters)
    by = replicas * 8.0 * nnz + 4 * (2 * s + 1)
    rows.append(dict(op="softmax", s=s, replicas=replicas, nnz=nnz, ms=med * 1e3, ms_min=best * 1e3,
                     alg_gbs=by / med / 1e9, hbm_frac=by / med / 8e12))
    print(json.dumps(rows[-1]), flush=True)
    ws = torch.empty(capi.spmm_workspace_bytes(s, s, d, nnz) + 16, dtype=torch.uint8, device=dev)
    med, best = timeit(lambda: capi.spmm_batched(s, s, d, replicas, ri, probs, nnz, ro, ci, v, ctx, ws), iters)
    by = replicas * (4.0 * nnz + 4 * 2 * s * d) + 4 * nnz + 4 * (2 * s + 1)
    rows.append(dict(op="spmm_attn", s=s, d=d, replicas=replicas, nnz=nnz, ms=med * 1e3, ms_min=best * 1e3,
                     gflops=2.0 * nnz * d * replicas / med / 1e9, alg_gbs=by / med / 1e9,
                     hbm_frac=by / med / 8e12))
    print(json.dumps(rows[-1]), flush=True)
    if capi.sparse_attention_supported(s, s, d, nnz):
        aws = torch.empty(capi.sparse_attention_workspace_bytes(s, s, d, nnz), dtype=torch.uint8,
                          device=dev)
        scale = 1.0 / d ** 0.5
        med, best = timeit(lambda: capi.sparse_attention_forward(s, s, d, replicas, ri, ro, ci, q, kk,
                                                                 v, scale, ctx, None, aws), iters)
        by = replicas * 4.0 * 4 * s * d + 4 * nnz + 4 * (2 * s + 1)  # Q, K, V in; O out
        rows.append(dict(op="attention_fused", s=s, d=d, replicas=replicas, nnz=nnz, ms=med * 1e3,
                         ms_min=best * 1e3, gflops=4.0 * nnz * d * replicas / med / 1e9,
                         alg_gbs=by / med / 1e9, hbm_frac=by / med / 8e12))
        print(json.dumps(rows[-1]), flush=True)
    return rows


def bench_transpose(dev, iters, m=2048, n=2048, density=0.2):
    ri, ro, ci, nnz = gpu_csr(m, n, density, dev, seed=9)
    vals = torch.rand(nnz, device=dev)
    ov = torch.empty_like(vals)
    oro = torch.empty(n + 1, dtype=torch.int32, device=dev)
    oci = torch.empty(nnz, dtype=torch.int32, device=dev)
    ws = torch.empty(capi.csr_transpose_workspace_bytes(m, n, nnz), dtype=torch.uint8, device=dev)
    med, best = timeit(lambda: capi.csr_transpose(m, n, 1, vals, ro, ci, ov, oro, oci, None, ws), iters)
    by = 16.0 * nnz + 4 * (m + n + 2)
    row = dict(op="transpose", m=m, n=n, nnz=nnz, ms=med * 1e3, ms_min=best * 1e3,
               alg_gbs=by / med / 1e9, hbm_frac=by / med / 8e12)
    print(json.dumps(row), flush=True)
    return [row]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ops", default="spmm,attn,transpose")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--densities", default="0.5,0.25,0.2,0.15,0.1,0.05")
    ap.add_argument("--replicas", type=int, default=1)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    print("lib:", capi.version(), "| device:", torch.cuda.get_device_name(0), flush=True)
    rows = []
    ops = args.ops.split(",")
    t0 = time.time()
    if "spmm" in ops:
        rows += bench_spmm(dev, args.iters, [float(x) for x in args.densities.split(",")],
                           m=args.size, k=args.size, n=args.size, replicas=args.replicas)
    if "attn" in ops:
        rows += bench_attention_ops(dev, args.iters)
    if "transpose" in ops:
        rows += bench_transpose(dev, args.iters)
    print(f"done in {time.time() - t0:.1f}s", flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
