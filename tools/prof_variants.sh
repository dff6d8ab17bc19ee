#!/bin/bash
# Run ON THE GPU BOX: device-side kernel times (rocprofv3 --kernel-trace --stats) of one
# tool invocation per variant.  usage: tools/prof_variants.sh <tag> <tool.py> <fixed args> -- <variant arg name> v1 v2 ...
# -> gpurun_out/pv_<tag>.txt (kernel name, calls, average ns per variant)
set -u
TAG=$1; TOOL=$2; shift 2
FIXED=()
while [ "$1" != "--" ]; do FIXED+=("$1"); shift; done
shift
ARG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/pv_${TAG}.txt
: > $OUT
for v in "$@"; do
  D=$R/gpurun_out/pv_${TAG}_$(echo $v | tr ':,' '__')
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/$TOOL "${FIXED[@]}" $ARG $v > $D.log 2>&1 || { echo "variant $v failed" >> $OUT; exit 1; }
  echo "== $v" >> $OUT
  python3 - "$D" >> $OUT <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        name = r["Name"].split("(")[0][-60:]
        print(f"  {name:60s} calls {r['Calls']:>5s} avg_ns {float(r['AverageNs']):10.0f} min_ns {r['MinNs']}")
PY
done
cat $OUT
