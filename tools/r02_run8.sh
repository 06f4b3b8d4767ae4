cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -x -k "fused_projection or bitwise" 2>&1 | tail -2
python bench.py --no-cpu-baseline --all-kernel-events --no-side-stream 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('single-stream', d['value'], d['ms_per_step'], r['category'], r['avg_launch_us']); print([(o['category'], o['avg_launch_us']) for o in r['others']])"
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-kernel-events --steps 20 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
