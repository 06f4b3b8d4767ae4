#!/bin/bash
# fp32 parity mode after the tail kernels of round 5: fp32 tests, A/B, kernel summary
O=gpurun_out/r05f32b; mkdir -p $O
pick() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
timeout 1500 python -m pytest tests -m gpu -x -q -k "fp32 or oracle or golden or config1_x2 or psnr or adam" 2>&1 | tail -5 | tee $O/pytest.txt
for o in 0 1 0 1; do python bench.py --dtype fp32 --no-also --no-cpu-baseline --no-kernel-events --steps 8 --warmup 2 --option fp32_fast=$o 2>/dev/null | pick | sed "s/^/fp32_fast=$o /" | tee -a $O/ab.txt; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $O/prof; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --config 1 --dtype fp32 --steps 5 --warmup 2 --no-kernel-events --no-side-stream --no-cpu-baseline --no-also > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_fp32.csv; find $O/prof -type f -delete
python3 - <<'P'
import csv
rows=list(csv.DictReader(open('gpurun_out/r05f32b/kernel_stats_fp32.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows); print('kernel sum per step (7 steps):', tot/7e6, 'ms')
for r in rows[:26]: print(r['Name'][:70].ljust(70), r['Calls'], round(float(r['TotalDurationNs'])/7e3), round(float(r['AverageNs'])/1e3,1))
P
