cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_swin.py tests/test_gpu_baseline_configs.py -q -m gpu -x -k "swin or semantic or config2" 2>&1 | tail -3
for i in 1 2; do
python bench.py --config 2 --no-cpu-baseline --no-kernel-events --steps 20 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('config2', d['value'], d['ms_per_step'], 'host', d['config']['host_enqueue_ms_per_step'])"
done
python bench.py --config 3 --no-cpu-baseline --no-kernel-events --steps 20 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('config3', d['value'], d['ms_per_step'], 'host', d['config']['host_enqueue_ms_per_step'])"
