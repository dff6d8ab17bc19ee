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
for (m, k, n, d, R) in [(2048,2048,256,0.2,4),(2048,2048,256,0.2,16),(2048,2048,512,0.2,4),(2048,2048,512,0.2,16),(512,512,1024,0.1,8),(2048,2048,1024,0.2,8),(4096,4096,256,0.1,1)]:
    ri, ro, ci, nnz = random_csr(m, k, d, dev, seed=3)
    vals = uniform((nnz,), dev, 4); b = uniform((R, k, n), dev, 5); o = torch.empty(R, m, n, device=dev)
    ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, nnz) + (1 << 20), dtype=torch.uint8, device=dev)
    t = timeit(lambda: capi.spmm_batched(m, k, n, R, ri, vals, 0, ro, ci, b, o, ws))
    print(f"left_spmm m={m} k={k} n={n} d={d} R={R}: {t:.3f} ms  {2.0*nnz*n*R/t/1e9:.2f} TF")
