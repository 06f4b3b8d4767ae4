#!/bin/bash
# usage: kstat.sh "<bench args>" <pattern>  -- single-stream rocprofv3 kernel summary of the current library, rows matching pattern
O=gpurun_out/kstat; mkdir -p $O; rm -rf $O/prof
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o k -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-events --no-side-stream --steps 5 --warmup 2 $1 > $GRAFT_REPO_ROOT/$O/log.txt 2>&1 )
python - "$2" <<'P'
import csv, sys, re
rows = list(csv.DictReader(open('gpurun_out/kstat/prof/k_kernel_stats.csv')))
pat = re.compile(sys.argv[1])
for r in rows:
    if pat.search(r['Name']): print('%-80s %5s %8.1f us' % (r['Name'][:80], r['Calls'], float(r['AverageNs']) / 1e3))
P
