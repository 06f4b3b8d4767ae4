cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu -x -k "not drift_at and not baseline_configs" 2>&1 | tail -4 | tee $O/pytest.log
for i in 1 2; do
python bench.py --no-cpu-baseline --no-kernel-events --steps 20 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('default', d['value'], d['ms_per_step'])"
done
python bench.py --no-cpu-baseline --all-kernel-events 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print(d['value'], d['ms_per_step'], r['category'], r['avg_launch_us'], r['frac']); print([(o['category'], o['avg_launch_us'], o['total_ms']) for o in r['others']])"
