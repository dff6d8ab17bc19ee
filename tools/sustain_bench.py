"""Developer timing script (GPU only); run directly, never imported."""


def main():
    import sys, time, torch
    sys.path.insert(0, "/root/repo")
    from torch_sputnik_amd import capi
    from torch_sputnik_amd.synthetic import random_csr, uniform
    dev = torch.device("cuda:0")
    m=k=n=4096
    ri, ro, ci, nnz = random_csr(m, k, 0.1, dev, seed=5234)
    vals = uniform((nnz,), dev, 4); b = uniform((k, n), dev, 5); o = torch.empty(m, n, device=dev)
    ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, nnz) + 16, dtype=torch.uint8, device=dev)
    capi.spmm_plan(m, k, n, ri, ro, ci, ws)
    def loop(fn, steps=50):
        for _ in range(10): fn()
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(steps): fn()
        torch.cuda.synchronize(); return (time.perf_counter()-t0)/steps*1e3
    full = lambda: capi.spmm_batched(m, k, n, 1, ri, vals, 0, ro, ci, b, o, ws)
    kern = lambda: capi.spmm_batched_planned(m, k, n, 1, ri, vals, 0, ro, ci, b, o, ws)
    for name, fn in (("planned (kernel only)", kern), ("per-call (pre-pass + kernel)", full), ("planned again", kern)):
        print(name, "sustained ms/step:", round(loop(fn), 4), " 200 steps:", round(loop(fn, 200), 4), flush=True)



if __name__ == "__main__":
    main()
