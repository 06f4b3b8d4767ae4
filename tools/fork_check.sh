#!/bin/bash
# same-box A/B of the fork mechanism + main-stream gaps from a kernel trace of each
O=gpurun_out/forkchk; mkdir -p $O
bash tools/ab_opts.sh "" "--option fork_on_kernel=1" 3
for v in 0 1; do
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof$v -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-events --option fork_on_kernel=$v > $GRAFT_REPO_ROOT/$O/prof$v.log 2>&1 )
  t=$O/prof$v/t_kernel_trace.csv; python tools/timeline.py <(head -1 $t) $t 0 10 > $O/timeline$v.txt 2>&1; head -3 $O/timeline$v.txt
done
