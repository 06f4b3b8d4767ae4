cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -x 2>&1 | tail -2
for r in 1 2 3; do
  for v in 0 1 512 1024; do
    M2T_CONV_WREG=$v python bench.py --no-cpu-baseline --steps 20 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; print('wreg=$v', d['value'], d['ms_per_step'], r['category'], r['avg_launch_us'])"
  done
done
