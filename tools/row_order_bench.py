"""Developer timing script (GPU only); run directly, never imported."""


def main():
    import sys, os, torch
    sys.path.insert(0, "/root/repo")
    from torch_sputnik_amd import capi
    from torch_sputnik_amd.synthetic import random_csr, uniform
    dev = torch.device("cuda:0")
    def timeit(fn, iters=30):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        ts=[]
        for _ in range(iters):
            s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
            s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
        return sorted(ts)[len(ts)//2]
    m=k=n=4096
    for d in (0.5, 0.2, 0.1, 0.05):
        ri, ro, ci, nnz = random_csr(m, k, d, dev, seed=3)
        vals = uniform((nnz,), dev, 4); b = uniform((k, n), dev, 5); o = torch.empty(m, n, device=dev)
        ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, nnz) + 16, dtype=torch.uint8, device=dev)
        lens = (ro[1:] - ro[:-1])
        ident = torch.arange(m, dtype=torch.int32, device=dev)
        # interleaved: slot j*16+b  <- sorted position ... emulate by permuting the sorted order
        nb = m // 256
        inter = ri.reshape(256, nb).t().contiguous().reshape(-1)   # block b gets sorted rows b, b+nb, ...
        res = []
        for name, order in (("sorted", ri), ("identity", ident), ("interleaved", inter)):
            t = timeit(lambda: capi.spmm_batched(m, k, n, 1, order, vals, 0, ro, ci, b, o, ws))
            res.append(f"{name} {t*1e3:.0f}us")
        print(f"d={d} rowlen min/max {int(lens.min())}/{int(lens.max())}: " + "  ".join(res), flush=True)



if __name__ == "__main__":
    main()
