cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu 2>&1 | grep -E "^FAILED|^E  |passed|failed" | head -20
for cfg in "0 1024 0" "1 2048 0" "1 2048 1" "1 1024 1" "1 512 1"; do
  set -- $cfg
  echo "PIPE=$1 BLOCKS=$2 XCD=$3"
  M2T_CONV_PIPE=$1 M2T_CONV_PIPE_BLOCKS=$2 M2T_CONV_XCD=$3 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print(d['value'], d['ms_per_step'], r['category'], r['avg_launch_us'], [(o['category'], o['avg_launch_us']) for o in r['others']])"
done
