#!/bin/bash
# Run ON THE GPU BOX (through gpurun): LDS bank-conflict counters over any python script of tools/.
# usage: tools/collect_pmc_lds.sh <tag> <script.py> [args...]
set -u
TAG=${1:-run}; shift
SCRIPT=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES \
  --output-format csv -d $R/gpurun_out/pmc_${TAG} -- python3 $R/$SCRIPT "$@" > $R/gpurun_out/pmc_${TAG}.log 2>&1 || exit 1
echo "pmc collected for $TAG"
