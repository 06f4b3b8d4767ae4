#!/bin/bash
# same-box check of the timing events: step time with sampled / all / no events, and the sampled average next to rocprofv3's view of the same run
O=gpurun_out/evchk; mkdir -p $O
for i in 1 2; do
for m in "" "--all-kernel-events" "--no-kernel-events"; do
  python bench.py --no-cpu-baseline --steps 20 $m 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d.get('roofline') or {}
print('$m'.ljust(22), d['ms_per_step'], r.get('category'), r.get('avg_launch_us'), r.get('frac'))"
done; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o ev -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 $O/prof.log | cut -c1-300
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -4 $f | cut -c1-200
t=$(find $O/prof -name "*kernel_trace.csv" | head -1); python tools/timeline.py $t > $O/timeline.txt 2>&1; grep "attn_bwd_res" $O/timeline.txt | cut -c1-120 | head -24
