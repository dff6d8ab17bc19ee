#!/bin/bash
# Run ON THE GPU BOX (through gpurun): one rocprofv3 --pmc pass (8 SQ counters) over
# any python script of tools/.  usage: tools/collect_pmc_script.sh <tag> <script.py> [args...]
set -u
TAG=${1:-run}; shift
SCRIPT=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE \
  --output-format csv -d $R/gpurun_out/pmc_${TAG} -- python3 $R/$SCRIPT "$@" > $R/gpurun_out/pmc_${TAG}.log 2>&1 || exit 1
echo "pmc collected for $TAG"
