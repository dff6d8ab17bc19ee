"""Small-problem latency of SDDMM: tiled (pre-pass + stationary kernel) vs row-wave kernel."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_sputnik_amd import capi
from torch_sputnik_amd.synthetic import random_csr, uniform
dev = torch.device("cuda:0")
def timeit(fn, iters=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts=[]
    for _ in range(iters):
        s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return sorted(ts)[len(ts)//2] * 1e3
for (sz, k, R) in ((64, 64, 1), (256, 64, 1), (1024, 64, 1), (1024, 64, 8), (1024, 64, 64), (512, 512, 1), (2048, 512, 1), (2048, 512, 8), (1024, 128, 16)):
    for d in (0.5, 0.1):
        ri, ro, ci, nnz = random_csr(sz, sz, d, dev, seed=3)
        lhs = uniform((R, sz, k), dev, 4); rhs = uniform((R, sz, k), dev, 5); o = torch.empty(R, nnz, device=dev)
        ws = torch.empty(capi.sddmm_workspace_bytes(sz, k, sz, nnz) + 16, dtype=torch.uint8, device=dev)
        t1 = timeit(lambda: capi.sddmm_batched(sz, k, sz, R, ri, ro, ci, lhs, rhs, o, ws))
        t2 = timeit(lambda: capi.sddmm_batched(sz, k, sz, R, ri, ro, ci, lhs, rhs, o, None))
        print(f"m=n={sz} k={k} R={R} d={d} W={nnz*k*R/1e6:.0f}M: tiled {t1:.1f} us, row wave {t2:.1f} us", flush=True)
