#!/bin/bash
# Developer tool: softmax forward / backward time against prefetch depth and rows per group.
for depth in ${DEPTHS:-1 2}; do for rpg in ${RPGS:-0 1 2 4}; do echo "depth=$depth rpg=$rpg"; SPUTNIK_HIP_SOFTMAX_DEPTH=$depth SPUTNIK_HIP_SOFTMAX_RPG=$rpg python tools/softmax_bench.py 2>&1 | tail -2 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('  R=%d fwd %.1f us (%.2f) bwd %.1f us (%.2f) copy %.1f'%(d['replicas'],d['fwd_us'],d['fwd_hbm_frac'],d['bwd_us'],d['bwd_hbm_frac'],d['copy_us']))"; done; done
