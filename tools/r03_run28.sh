cd $GRAFT_REPO_ROOT
ab() { python bench.py --no-cpu-baseline --no-kernel-events --steps 20 $2 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
ab gate2 ""
ab gate0_one_fork "--option gate_branch=0"
ab gate1 "--option gate_branch=1"
ab gate3 "--option gate_branch=3"
ab ungated "--option gate_branch=-1"
ab single "--no-side-stream"
done
ab c3_gate2 "--config 3"
ab c3_gate0 "--config 3 --option gate_branch=0"
ab c3_gate1 "--config 3 --option gate_branch=1"
