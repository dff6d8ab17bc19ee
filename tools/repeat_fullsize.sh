#!/bin/bash
# Developer tool: the full-size SpMM test N times in fresh processes (a kernel
# whose waits are counted by hand is checked for timing-dependent failures this way).
n=${1:-6}
for i in $(seq 1 $n); do
  timeout -k 10 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "spmm_full_size" > gpurun_out/t_rep_$i.log 2>&1 || { echo "run $i FAILED"; grep -v "^  File\|Extension modules" gpurun_out/t_rep_$i.log | tail -4 | cut -c1-200; exit 1; }
  echo "run $i ok"
done
AMD_LOG_LEVEL=2 timeout -k 10 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "spmm_full_size" > gpurun_out/t_rep_log.log 2>&1 || { echo "logged run FAILED"; exit 1; }
echo "logged run ok"
