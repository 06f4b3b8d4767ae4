cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-kernel-events --steps 20 $@ 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
run --option fused_c16_dgrad=0
run --option fused_c16_dgrad=1
run --option fused_c16_dgrad=1 --option gate_branch=0
done
run --config 3 --option fused_c16_dgrad=0
run --config 3 --option fused_c16_dgrad=1
run --config 3 --option fused_c16_dgrad=0
run --config 3 --option fused_c16_dgrad=1
