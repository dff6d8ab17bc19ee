"""Config 3's whole SparseAttention training step (forward + backward through the separate
operators) a few times -- the workload of tools/profile_kernels.sh for
profiles/r*_c3_fwd_bwd_kernel_stats.csv.   usage: python tools/c3_step.py [iters] [fused]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_sputnik_amd import SparseAttention  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    fused = len(sys.argv) > 2
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    s, emb, heads, batch = 1024, 512, 8, 8
    attn = SparseAttention(heads, emb, max_sequence_length=s, device=dev, sparsity=0.9,
                           mask_generator=np.random.default_rng(0))
    for lin in attn.linears:
        w = torch.randn(emb, emb, device=dev) * (torch.rand(emb, emb, device=dev) < 0.1)
        lin.weight = torch.nn.Parameter(w)
        lin.setup_sparse_tensors()
    attn.low_memory_training = fused
    attn.differentiable_softmax = not fused
    x = torch.randn(batch, s, emb, device=dev).requires_grad_(True)
    gout = torch.randn(batch, s, emb, device=dev)
    def step():
        x.grad = None
        for lin in attn.linears:
            lin.values.grad = None
        attn(x, x, x, None).backward(gout)

    for it in range(iters + 3):
        if it == 3:
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
        step()
    b.record()
    torch.cuda.synchronize()
    print(f"c3 forward + backward: {a.elapsed_time(b) / iters:.4f} ms per step", flush=True)
    if os.environ.get("C3_STEP_HOST_PROFILE"):
        # where the HOST spends its time per step (the eager step is ~27 launches in ~0.58 ms:
        # a host that needs more than ~20 us per launch leaves the GPU waiting): the device
        # queue is drained after every step so that only host work is on the clock
        import cProfile
        import pstats
        import time
        prof = cProfile.Profile()
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        print(f"host wall per step, unprofiled: {(time.perf_counter() - t0) / n * 1e3:.4f} ms", flush=True)
        prof.enable()
        for _ in range(n):
            step()
        prof.disable()
        torch.cuda.synchronize()
        st = pstats.Stats(prof)
        st.sort_stats("cumulative").print_stats(45)
        st.sort_stats("tottime").print_stats(30)


if __name__ == "__main__":
    main()
