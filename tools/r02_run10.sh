cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_swin.py -q -m gpu -x 2>&1 | tail -6
for i in 1 2 3; do python bench.py --config 2 --no-cpu-baseline --no-kernel-events --steps 10 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
python bench.py --config 2 --no-cpu-baseline --no-kernel-events --steps 10 --no-overlap-semantic 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('serial', d['value'], d['ms_per_step'])"
