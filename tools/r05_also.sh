#!/bin/bash
# why does configs[4] read ~9 % lower inside the default line's `also` list than as its own process?  (+ fp32 tests / A/B of the VALU tail kernels)
O=gpurun_out/r05also; mkdir -p $O
pick() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step'])
for a in d.get('also',[]): print('  also', a.get('workload'), a.get('value'), a.get('ms_per_step'), a.get('steps'), a.get('warmup'))
"; }
timeout 1200 python -m pytest tests -m gpu -x -q -k "fp32 or oracle or golden or config1_x2 or psnr" 2>&1 | tail -4 | tee $O/pytest.txt
python bench.py --no-cpu-baseline 2>/dev/null | pick | tee $O/a.txt
python bench.py --no-cpu-baseline --also-steps 20 2>/dev/null | pick | tee -a $O/a.txt
python bench.py --no-cpu-baseline --config 4 --no-also --steps 6 --warmup 2 2>/dev/null | pick | tee -a $O/a.txt
python bench.py --no-cpu-baseline --config 4 --no-also --steps 20 --warmup 5 2>/dev/null | pick | tee -a $O/a.txt
for o in 0 1 0 1; do python bench.py --dtype fp32 --no-also --no-cpu-baseline --no-kernel-events --steps 8 --warmup 2 --option fp32_fast=$o 2>/dev/null | pick | sed "s/^/fp32_fast=$o /" | tee -a $O/a.txt; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $O/prof; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --config 1 --dtype fp32 --steps 5 --warmup 2 --no-kernel-events --no-side-stream --no-cpu-baseline --no-also > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_fp32.csv; find $O/prof -type f -delete
head -25 $O/kernel_stats_fp32.csv | cut -c1-110
