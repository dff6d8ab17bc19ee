"""Developer timing script (GPU only): attention-projection shapes per SpMM kernel choice."""


def main():
    import os, sys, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from torch_sputnik_amd import capi
    from torch_sputnik_amd.synthetic import random_csr, uniform
    dev = torch.device("cuda:0")

    def timeit(fn, iters=10, burst=40):
        # bursts of back-to-back launches: a lone launch between host syncs runs
        # on a clock that has dropped (4096^3: 0.88 ms alone, 0.35 ms in a burst)
        for _ in range(burst):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(iters):
            s = torch.cuda.Event(enable_timing=True)
            e = torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(burst):
                fn()
            e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / burst)
        return sorted(ts)[len(ts) // 2] * 1e3

    shapes = [(512, 512, 1024, 8, 0.1), (512, 512, 1024, 8, 0.5), (512, 512, 512, 16, 0.1),
              (2048, 512, 1024, 1, 0.1), (4096, 512, 4096, 1, 0.1), (512, 64, 512, 64, 0.1),
              (1024, 256, 256, 8, 0.25), (1024, 1024, 64, 64, 0.1), (1024, 1024, 64, 64, 0.3),
              (2048, 2048, 64, 16, 0.1), (4096, 4096, 64, 8, 0.1), (4096, 4096, 72, 1, 0.1),
              (2048, 1024, 1024, 1, 0.1), (1024, 1024, 128, 32, 0.1)]
    if len(sys.argv) > 1:   # e.g. "4096,4096,4096,1,0.1"
        f = sys.argv[1].split(",")
        shapes = [(int(f[0]), int(f[1]), int(f[2]), int(f[3]), float(f[4]))]
    for (m, k, n, R, d) in shapes:
        ri, ro, ci, nnz = random_csr(m, k, d, dev, seed=3)
        vals = uniform((nnz,), dev, 4)
        b = uniform((R, k, n), dev, 5)
        o = torch.empty(R, m, n, device=dev)
        os.environ.pop("SPUTNIK_HIP_SPMM_KERNEL", None)   # the size query follows the knob too
        capi.reload_options()
        ws = torch.empty(capi.spmm_workspace_bytes(m, k, n, nnz) + (1 << 20), dtype=torch.uint8, device=dev)
        line = [f"{m}x{k}x{n} R={R} d={d}:"]
        ref = None
        for kern in ("auto", "panel", "narrow", "wide", "wide512", "gather"):
            if kern == "auto":
                os.environ.pop("SPUTNIK_HIP_SPMM_KERNEL", None)
            else:
                os.environ["SPUTNIK_HIP_SPMM_KERNEL"] = kern
            capi.reload_options()
            capi.spmm_plan(m, k, n, ri, ro, ci, ws)
            t = timeit(lambda: capi.spmm_batched_planned(m, k, n, R, ri, vals, 0, ro, ci, b, o, ws))
            if ref is None:
                ref = o.clone()
            err = (o - ref).abs().max().item()
            line.append(f"{kern} {t:.1f}us (d {err:.1e})")
        print("  ".join(line), flush=True)


if __name__ == "__main__":
    main()
