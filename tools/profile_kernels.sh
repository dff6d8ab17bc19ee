#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel trace of one python script, prints the
# per-kernel averages of our kernels.   usage: tools/profile_kernels.sh <script.py> <tag> [name filter]
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$2 -o $2 -- python3 $R/$1 > $R/gpurun_out/prof_$2.log 2>&1
cd $R
python3 - "$2" "${3:-sputnik_hip}" <<'P'
import csv, glob, sys
f = glob.glob('gpurun_out/prof_%s/**/*kernel_stats.csv' % sys.argv[1], recursive=True)[0]
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r['Name']:
        print('%-90s calls %5s avg %9.1f ns min %8s' % (r['Name'][:90], r['Calls'], float(r['AverageNs']), r['MinNs']))
P
