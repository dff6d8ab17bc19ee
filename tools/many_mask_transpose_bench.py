#!/usr/bin/env python3
"""csr_transpose_many_mask at the attention size (8 masks of 1024^2 at densities 0.1 / 0.2 /
0.05 / 0.5, 8 heads each): all masks in the same three launches (a region of tables per mask
in the workspace) against mask after mask (the single-mask workspace).

    python tools/many_mask_transpose_bench.py
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from bench import event_time_ms  # noqa: E402
from torch_sputnik_amd import capi  # noqa: E402
from torch_sputnik_amd.synthetic import random_csr  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--which', default='regions,single')
args = ap.parse_args()
dev = torch.device('cuda:0')
s = 1024; b_mm, h_mm = 8, 8
dens = (0.1, 0.2, 0.05, 0.5)
topo = [random_csr(s, s, dens[i % len(dens)], dev, seed=70 + i) for i in range(b_mm)]
nn = torch.tensor([t4[3] for t4 in topo], dtype=torch.int32)
width = int(nn.max())
mro = torch.cat([t4[1] for t4 in topo]); mci = torch.cat([t4[2] for t4 in topo])
r = b_mm * h_mm
vals = torch.rand(r, width, device=dev)
vt = torch.zeros_like(vals); rot = torch.empty(b_mm, s + 1, dtype=torch.int32, device=dev); cit = torch.empty_like(mci)
for name, nb in [x for x in (("regions", capi.csr_transpose_many_mask_workspace_bytes(b_mm, s, s, width)), ("single", capi.csr_transpose_workspace_bytes(s, s, width))) if x[0] in args.which.split(",")]:
    w = torch.empty(nb, dtype=torch.uint8, device=dev)
    t = event_time_ms(lambda: capi.csr_transpose_many_mask(b_mm, s, s, nn, r, vals, mro, mci, vt, rot, cit, None, w), 50)
    print(name, nb, "ms", t)
