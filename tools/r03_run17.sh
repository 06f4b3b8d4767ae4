cd $GRAFT_REPO_ROOT
ab() { python bench.py --no-cpu-baseline --no-kernel-events --steps 20 $2 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
for t in 512 768 1024 256; do
export M2T_WGRAD_TARGET=$t
ab target$t ""
done
done
for t in 512 768 1024; do
export M2T_WGRAD_TARGET=$t
ab c3_target$t "--config 3"
done
