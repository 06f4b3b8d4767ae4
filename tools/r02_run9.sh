cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for o in 1 0; do
python bench.py --no-cpu-baseline --all-kernel-events --no-side-stream --option fused_qkv_dgrad=$o 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('single-stream', d['value'], d['ms_per_step'], r['category'], r['avg_launch_us']); print([(o['category'], o['avg_launch_us']) for o in r['others']])"
done
