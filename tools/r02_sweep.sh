cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-kernel-events --steps 20 $@ 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
run --option gate_branch=2
run --option gate_branch=1
run --option gate_branch=1 --option side_conv_pos=0
run --config 3 --option gate_branch=2
run --config 3 --option gate_branch=1
run --config 3 --option gate_branch=1 --option side_conv_pos=0
run --config 4 --option gate_branch=2
run --config 4 --option gate_branch=1
done
