cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-kernel-events --steps 20 $@ 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
run --option side_priority=0
run --option side_priority=1
run --option side_priority=-1
done
for rep in 1 2; do
run --config 3 --option side_priority=0
run --config 3 --option side_priority=1
done
