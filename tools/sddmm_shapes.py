import sys, torch
sys.path.insert(0, "/root/repo")
from torch_sputnik_amd import capi
from torch_sputnik_amd.synthetic import random_csr, uniform
dev = torch.device("cuda:0")
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts=[]
    for _ in range(iters):
        s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return sorted(ts)[len(ts)//2]
for (m, n, k, d, R) in [(2048, 2048, 2048, 0.2, 1), (2048, 2048, 512, 0.2, 4), (2048, 2048, 256, 0.2, 4), (2048,2048,2048,0.2,4)]:
    ri, ro, ci, nnz = random_csr(m, n, d, dev, seed=3)
    lhs = uniform((R, m, k), dev, 1); rhs = uniform((R, n, k), dev, 2)
    out = torch.empty(R, nnz, device=dev)
    ws = torch.empty(capi.sddmm_workspace_bytes(m, k, n, nnz) + 16, dtype=torch.uint8, device=dev)
    t = timeit(lambda: capi.sddmm_batched(m, k, n, R, ri, ro, ci, lhs, rhs, out, ws))
    print(f"sddmm m={m} n={n} k={k} d={d} R={R}: {t:.3f} ms  {2.0*nnz*k*R/t/1e9:.2f} TFLOP/s eff")
    # left_spmm of the same layer (forward), n = k here
    vals = uniform((nnz,), dev, 4); b = uniform((R, n, k), dev, 5); o = torch.empty(R, m, k, device=dev)
    ws2 = torch.empty(capi.spmm_workspace_bytes(m, n, k, nnz) + 16, dtype=torch.uint8, device=dev)
    t2 = timeit(lambda: capi.spmm_batched(m, n, k, R, ri, vals, 0, ro, ci, b, o, ws2))
    print(f"  left_spmm same layer (n={k}): {t2:.3f} ms  {2.0*nnz*k*R/t2/1e9:.2f} TFLOP/s eff")
