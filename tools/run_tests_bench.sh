cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print(d['value'], d['ms_per_step'], r['category'], r['avg_launch_us'], r['frac'], [(o['category'], o['avg_launch_us']) for o in r['others']])"
python bench.py --no-cpu-baseline --all-kernel-events 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print(d['value'], d['ms_per_step'], r['category'], r['avg_launch_us'], r['frac']); print([(o['category'], o['avg_launch_us']) for o in r['others']])"
