#!/bin/bash
# Parts of the tiled SDDMM switched off (SPUTNIK_HIP_SDDMM_DEBUG bits: 1 no compute,
# 2 no slab staging, 8 the 16-lanes-per-entry kernel), config 3 shapes.
for d in 0 1 2 3 8 9 10; do
  echo "== SPUTNIK_HIP_SDDMM_DEBUG=$d"
  SPUTNIK_HIP_SDDMM_DEBUG=$d python tools/half_bench.py sddmm 2>&1 | grep -E "c3|k=128"
done
