#!/bin/bash
# Run ON THE GPU BOX (through gpurun): one rocprofv3 --pmc pass counting the
# instructions each kernel issues by class (the tiled SpMM loop is bound by
# scalar instruction issue).  usage: tools/collect_pmc_insts.sh <tag> [kbench args...]
set -u
TAG=${1:-run}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU \
  --output-format csv -d $R/gpurun_out/pmci_${TAG} -- python3 $R/tools/kbench.py "$@" > $R/gpurun_out/pmci_${TAG}.log 2>&1 || exit 1
echo "instruction counters collected for $TAG"
