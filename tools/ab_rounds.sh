# usage: ab_rounds.sh "<bench args>" [rounds]   -- alternates the ROUND-4 tree (scratch/r4tree: bench.py + m2trans_amd/ + libm2t.so of commit
# f313e5b, unpacked by hand with `git archive`; git-ignored) and this tree on the same box.  (ab_libs.sh cannot cross a round that adds C-ABI
# symbols: the new Python binding refuses the old library.)
cd $GRAFT_REPO_ROOT
ARGS="$1"; N=${2:-3}
for r in $(seq 1 $N); do
  for v in A B; do
    if [ $v = A ]; then B=scratch/r4tree/bench.py; X=""; else B=bench.py; X="--no-also"; fi
    python $B --no-cpu-baseline --no-kernel-events $X $ARGS 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"
  done
done
