"""Developer timing (GPU only): tiled SDDMM against the k-panel width, plain and summed."""


def main():
    import os, sys, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from torch_sputnik_amd import capi
    from torch_sputnik_amd.synthetic import random_csr, uniform
    dev = torch.device("cuda:0")

    def timeit(fn, iters=10, burst=30):
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

    # (m, k, n, replicas, density, summed)
    shapes = [(2048, 512, 2048, 8, 0.2, True), (512, 1024, 512, 8, 0.1, True), (512, 512, 512, 8, 0.1, True),
              (4096, 512, 4096, 8, 0.05, True), (1024, 1024, 1024, 8, 0.3, True), (4096, 512, 4096, 4, 0.1, True),
              (2048, 256, 2048, 8, 0.1, True), (2048, 512, 2048, 8, 0.2, False), (512, 1024, 512, 8, 0.5, True)]
    for (m, k, n, R, d, summed) in shapes:
        ri, ro, ci, nnz = random_csr(m, n, d, dev, seed=3)
        lhs = uniform((R, m, k), dev, 4)
        rhs = uniform((R, n, k), dev, 5)
        line = [f"{m}x{k}x{n} R={R} d={d} {'sum' if summed else 'each'}:"]
        ref = None
        for width in (0, 512, 256, 128, 64):
            if width and k % width:
                continue
            os.environ["SPUTNIK_HIP_SDDMM_PANEL"] = str(width)
            os.environ["SPUTNIK_HIP_SDDMM_KERNEL"] = "tiled"
            capi.reload_options()
            size = capi.sddmm_sum_workspace_bytes if summed else capi.sddmm_workspace_bytes
            ws = torch.empty(size(m, k, n, nnz) + 16, dtype=torch.uint8, device=dev)
            if ws.numel() <= 16:
                continue
            (capi.sddmm_sum_plan if summed else capi.sddmm_plan)(m, k, n, ri, ro, ci, ws)
            if summed:
                out = torch.empty(nnz, device=dev)
                scratch = torch.empty(capi.sddmm_sum_scratch_bytes(m, k, n, nnz, R) + 16, dtype=torch.uint8, device=dev)
                fn = lambda: capi.sddmm_sum_batched(m, k, n, R, ri, ro, ci, lhs, rhs, out, ws, scratch, planned=True)
            else:
                out = torch.empty(R, nnz, device=dev)
                fn = lambda: capi.sddmm_batched_planned(m, k, n, R, ri, ro, ci, lhs, rhs, out, ws)
            t = timeit(fn)
            if ref is None:
                ref = out.clone()
            line.append(f"w{width} {t:.1f}us (d {(out - ref).abs().max().item():.1e})")
        print("  ".join(line), flush=True)


if __name__ == "__main__":
    main()
