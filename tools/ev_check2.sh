#!/bin/bash
O=gpurun_out/evchk2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o ev -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-events > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 $O/prof.log | cut -c1-200
t=$O/prof/ev_kernel_trace.csv; python tools/timeline.py <(head -1 $t) $t 0 10 > $O/timeline.txt 2>&1; grep "attn_bwd_res" $O/timeline.txt | cut -c1-120 | head -24
head -3 $O/prof/ev_kernel_stats.csv | cut -c1-200
