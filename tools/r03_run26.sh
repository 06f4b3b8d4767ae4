cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_baseline_configs.py -q -m gpu -x -k "tail or fast_kernels or small_model or config1" 2>&1 | tail -3
ab() { python bench.py --no-cpu-baseline --no-kernel-events --steps 20 $2 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do ab two_wg_tail ""; done
ab c3 "--config 3"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r03h
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03h -- python3 bench.py --steps 5 --warmup 2 --no-kernel-events --no-side-stream --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv, glob
f=glob.glob('gpurun_out/r03h/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'tail_' in r['Name']: print(r['Name'][:60], r['Calls'], '%.1f us avg'%(float(r['AverageNs'])/1e3))
PY
rm -rf gpurun_out/r03h
