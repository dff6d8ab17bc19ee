#!/bin/bash
# Developer tool: SpMM kernel time with parts of the main loop switched off
# (SPUTNIK_HIP_SPMM_DEBUG: 1 no arithmetic, 2 staging re-reads chunk 0, 4 no
# barrier; results are wrong, only the time means something).
# usage: tools/dbg_sweep.sh [wide512|wide|narrow]
kern=${1:-auto}
for d in 0 1 2 4 5; do
  echo "kernel=$kern DEBUG=$d"
  SPUTNIK_HIP_SPMM_KERNEL=$kern SPUTNIK_HIP_SPMM_DEBUG=$d timeout -k 10 100 python tools/kbench.py --ops spmm --densities 0.5,0.1,0.05 2>&1 | grep '"op"' | cut -c60-140
done
