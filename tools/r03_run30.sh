cd $GRAFT_REPO_ROOT
ab() { python bench.py --no-cpu-baseline --no-kernel-events --steps 20 $2 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3 4; do
ab ungated ""
ab three_forks "--option gate_branch=-2"
done
ab c3_ungated "--config 3"
ab c3_three_forks "--config 3 --option gate_branch=-2"
