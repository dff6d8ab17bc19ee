"""left_spmm at SparseLinear-like shapes under each kernel choice (SPUTNIK_HIP_SPMM_KERNEL)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_sputnik_amd import capi
from torch_sputnik_amd.synthetic import random_csr, uniform
dev = torch.device("cuda:0")
def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts=[]
    for _ in range(iters):
        s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return sorted(ts)[len(ts)//2]
shapes = [(512,512,1024,0.1,8),(512,512,1024,0.1,1),(512,512,256,0.1,8),(1024,1024,256,0.1,4),(1024,1024,1024,0.1,1),(1024,1024,1024,0.5,1),
          (256,256,64,0.5,64),(1024,1024,64,0.1,16),(2048,2048,256,0.2,4),(2048,2048,512,0.2,4),(4096,4096,256,0.1,1)]
for (m, k, n, d, R) in shapes:
    ri, ro, ci, nnz = random_csr(m, k, d, dev, seed=3)
    vals = uniform((nnz,), dev, 4); b = uniform((R, k, n), dev, 5); o = torch.empty(R, m, n, device=dev)
    line = f"m={m} k={k} n={n} d={d} R={R} W={nnz*n*R/1e6:.0f}M:"
    for kern in ("auto", "gather", "narrow", "wide"):
        if kern == "auto": os.environ.pop("SPUTNIK_HIP_SPMM_KERNEL", None)
        else: os.environ["SPUTNIK_HIP_SPMM_KERNEL"] = kern
        capi.reload_options()  # the library reads its knobs once
        ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, nnz) + (1 << 20), dtype=torch.uint8, device=dev)
        t = timeit(lambda: capi.spmm_batched(m, k, n, R, ri, vals, 0, ro, ci, b, o, ws))
        line += f"  {kern} {t*1e3:.1f}us"
    print(line, flush=True)
