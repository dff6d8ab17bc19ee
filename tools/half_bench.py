"""Timing of the natively typed operators at config 3 (float32 against float16 /
bfloat16 storage).  Usage: python tools/half_bench.py [softmax] [sddmm] [spmm]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_sputnik_amd import capi  # noqa: E402
from torch_sputnik_amd.synthetic import random_csr, uniform  # noqa: E402


def ev(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # us


def main():
    what = set(sys.argv[1:]) or {"softmax", "sddmm", "spmm"}
    dev = torch.device("cuda:0")
    s, d = 1024, 64
    ri, ro, ci, nnz = random_csr(s, s, 0.1, dev, seed=7)
    for reps in (64, 512):
        x = uniform((reps, nnz), dev, 3) * 8 - 4
        if "softmax" in what:
            for dt in (torch.float32, torch.float16, torch.bfloat16):
                a, y, g = x.to(dt), torch.empty(reps, nnz, device=dev, dtype=dt), torch.empty(reps, nnz, device=dev, dtype=dt)
                tf = ev(lambda: capi.sparse_softmax_typed(s, reps, a, ri, ro, ci, 1.0, y))
                tb = ev(lambda: capi.sparse_softmax_backward_typed(s, reps, y, a, ro, 1.0, g))
                eb = a.element_size()
                print(f"softmax R={reps} {str(dt):16s} fwd {tf:7.2f} us ({2 * eb * reps * nnz / tf / 8e6:.3f} of 8 TB/s)  "
                      f"bwd {tb:7.2f} us ({3 * eb * reps * nnz / tb / 8e6:.3f})", flush=True)
    if "sddmm" in what:
        for (m, k, n, dens, reps, tag) in ((1024, 64, 1024, 0.1, 64, "c3 attention scores"),
                                           (1024, 128, 1024, 0.1, 64, "k=128"),
                                           (2048, 512, 2048, 0.2, 8, "c5 weight gradient (per replica)")):
            ri, ro, ci, nnz = random_csr(m, n, dens, dev, seed=7)
            ws = torch.empty(capi.sddmm_workspace_bytes(m, k, n, nnz) + 16, dtype=torch.uint8, device=dev)
            for dt in (torch.float32, torch.float16, torch.bfloat16):
                q = (uniform((reps, m, k), dev, 11) - 0.5).to(dt)
                kk = (uniform((reps, n, k), dev, 12) - 0.5).to(dt)
                out = torch.empty(reps, nnz, device=dev)
                t = ev(lambda: capi.sddmm_typed(m, k, n, reps, ri, ro, ci, q, kk, out, ws), 20)
                capi.sddmm_plan(m, k, n, ri, ro, ci, ws)
                tp = ev(lambda: capi.sddmm_typed(m, k, n, reps, ri, ro, ci, q, kk, out, ws, planned=True), 20)
                line = f"sddmm {tag:34s} {str(dt):15s} {t:8.2f} us  planned {tp:8.2f} us ({2.0 * nnz * k * reps / tp / 1e6:7.1f} TFLOP/s)"
                if dt != torch.float32 and k <= 256:
                    oh = torch.empty(reps, nnz, device=dev, dtype=dt)
                    th = ev(lambda: capi.sddmm_typed(m, k, n, reps, ri, ro, ci, q, kk, oh, ws, planned=True), 20)
                    line += f"  half out {th:8.2f} us"
                print(line, flush=True)
    if "spmm" in what:
        for (m, k, n, dens, reps, shared, tag) in ((1024, 1024, 64, 0.1, 64, False, "c3 attention P.V"),
                                                   (512, 512, 1024, 0.1, 8, True, "c3 projection"),
                                                   (2048, 2048, 512, 0.2, 8, True, "c5 left_spmm")):
            ri, ro, ci, nnz = random_csr(m, k, dens, dev, seed=7)
            out = torch.empty(reps, m, n, device=dev)
            for dt in (torch.float32, torch.float16, torch.bfloat16):
                v = (uniform((nnz,) if shared else (reps, nnz), dev, 11) - 0.5).to(dt)
                b = (uniform((reps, k, n), dev, 12) - 0.5).to(dt)
                need = max(capi.spmm_typed_workspace_bytes(m, k, n, nnz, reps, v, 0 if shared else nnz, b),
                           capi.spmm_workspace_bytes(m, k, n, nnz))
                ws = torch.empty(need + 256, dtype=torch.uint8, device=dev)
                t = ev(lambda: capi.spmm_typed(m, k, n, reps, ri, v, 0 if shared else nnz, ro, ci, b, out, ws), 20)
                line = f"spmm  {tag:34s} {str(dt):15s} {t:8.2f} us ({2.0 * nnz * n * reps / t / 1e6:7.1f} TFLOP/s)"
                if dt != torch.float32:   # what the widening path costs: cast both, then the float kernels
                    tw = ev(lambda: capi.spmm_typed(m, k, n, reps, ri, v.float(), 0 if shared else nnz, ro, ci,
                                                    b.float(), out, ws), 20)
                    line += f"   widen + float kernels {tw:8.2f} us"
                print(line, flush=True)


if __name__ == "__main__":
    main()
