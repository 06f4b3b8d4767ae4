cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_baseline_configs.py -q -m gpu -x 2>&1 | tail -3
ab() { python bench.py --no-cpu-baseline --no-kernel-events --steps 20 $2 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do ab default ""; done
ab c3 "--config 3"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03g -- python3 bench.py --steps 5 --warmup 2 --no-kernel-events --no-side-stream --no-cpu-baseline > /dev/null 2>&1
find gpurun_out/r03g -name "*kernel_stats.csv" | head -1 | xargs grep -h "instnorm\|conv3x3_c64_rows" | cut -d, -f1-4 | cut -c1-120
find gpurun_out/r03g -type f -delete
