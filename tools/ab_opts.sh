# usage: ab_opts.sh "<bench args A>" "<bench args B>" [rounds]   -- alternates two bench.py configurations on the same box
cd $GRAFT_REPO_ROOT
A="$1"; B="$2"; N=${3:-3}
for r in $(seq 1 $N); do
  for v in A B; do
    if [ $v = A ]; then ARGS="$A"; else ARGS="$B"; fi
    python bench.py --no-cpu-baseline --no-kernel-events --no-also --steps 20 $ARGS 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"
  done
done
