#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 passes of `bench.py --no-extras`.
#   kernel-trace/stats and each PMC counter in its own run (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2).
# usage: tools/collect_profiles.sh <tag>      -> gpurun_out/prof_<tag>_{stats,fetch,write}
# then HERE: python tools/summarize_profile.py --stats gpurun_out/prof_<tag>_stats --fetch ... --write ...
#            --bench-log gpurun_out/prof_<tag>_stats.log --tag <tag> --traffic-name spmm_c2_d010_traffic.json
set -u
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_stats -- python3 $R/bench.py --no-extras --steps 20 --warmup 5 > $R/gpurun_out/prof_${TAG}_stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_fetch -- python3 $R/bench.py --no-extras --steps 5 --warmup 2 > $R/gpurun_out/prof_${TAG}_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_write -- python3 $R/bench.py --no-extras --steps 5 --warmup 2 > $R/gpurun_out/prof_${TAG}_write.log 2>&1 || exit 1
echo "profiles collected for $TAG"
