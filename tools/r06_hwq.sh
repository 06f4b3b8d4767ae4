# round 6: the default line (headline + `also`) with the runtime's 4 hardware queues against GPU_MAX_HW_QUEUES=8, alternated on one box
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d['also_summary']
print('queues $q', d['value'], d['ms_per_step'], [s[k][0] for k in ('config3', 'config4', 'config2', 'config1_fp32')])"
done; done 2>&1 | tee gpurun_out/r06_hw_queues.txt
