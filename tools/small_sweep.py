"""Small-problem latency: tiled path (pre-pass + kernel) vs the workspace-free row-gather kernel."""
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
for sz in (64, 128, 256, 512, 1024):
    for d in (0.5, 0.1):
        ri, ro, ci, nnz = random_csr(sz, sz, d, dev, seed=3)
        vals = uniform((nnz,), dev, 4); b = uniform((sz, sz), dev, 5); o = torch.empty(sz, sz, device=dev)
        ws = torch.empty(capi.spmm_workspace_bytes(sz, sz, sz, nnz) + 16, dtype=torch.uint8, device=dev)
        t1 = timeit(lambda: capi.spmm_batched(sz, sz, sz, 1, ri, vals, 0, ro, ci, b, o, ws))
        t2 = timeit(lambda: capi.spmm_batched(sz, sz, sz, 1, ri, vals, 0, ro, ci, b, o, None))
        print(f"{sz}^3 d={d}: with workspace {t1:.1f} us, row gather {t2:.1f} us", flush=True)
