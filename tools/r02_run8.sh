cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_baseline_configs.py -q -m gpu -x 2>&1 | tail -2
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-kernel-events --steps 20 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
