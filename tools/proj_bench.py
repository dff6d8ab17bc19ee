"""Developer timing script (GPU only); run directly, never imported."""


def main():
    import sys, os, torch
    sys.path.insert(0, "/root/repo")
    from torch_sputnik_amd import capi
    from torch_sputnik_amd.synthetic import random_csr, uniform
    dev = torch.device("cuda:0")
    def timeit(fn, iters=50):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        ts=[]
        for _ in range(iters):
            s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
            s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
        return sorted(ts)[len(ts)//2]*1e3
    m=k=512; n=1024; R=8
    ri, ro, ci, nnz = random_csr(m, k, 0.1, dev, seed=3)
    vals = uniform((nnz,), dev, 4); b = uniform((R, k, n), dev, 5); o = torch.empty(R, m, n, device=dev)
    ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, nnz) + (1<<20), dtype=torch.uint8, device=dev)
    capi.spmm_plan(m, k, n, ri, ro, ci, ws)
    print("per call", timeit(lambda: capi.spmm_batched(m, k, n, R, ri, vals, 0, ro, ci, b, o, ws)))
    print("planned ", timeit(lambda: capi.spmm_batched_planned(m, k, n, R, ri, vals, 0, ro, ci, b, o, ws)))



if __name__ == "__main__":
    main()
