import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kbench import timeit
from torch_sputnik_amd import capi
dev = torch.device("cuda:0")
s = 1024
for per in (104, 102, 100, 128):
    nnz = s * per
    ro = (torch.arange(s + 1, device=dev) * per).to(torch.int32)
    ci = torch.arange(per, device=dev, dtype=torch.int32).repeat(s)
    ri = torch.arange(s, device=dev, dtype=torch.int32)
    for reps in (64, 512):
        x = torch.rand(reps, nnz, device=dev) * 8 - 4
        y = torch.empty_like(x)
        t, tmin = timeit(lambda: capi.sparse_softmax_batched(s, reps, x, ri, ro, ci, y), 100, 20)
        cp, _ = timeit(lambda: y.copy_(x), 100, 20)
        print(per, reps, "fwd us %.1f frac %.3f copy %.1f" % (t * 1e6, reps * 8.0 * nnz / t / 8e12, cp * 1e6))
