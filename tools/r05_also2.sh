#!/bin/bash
O=gpurun_out/r05also; mkdir -p $O
pick() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step'])
for a in d.get('also',[]): print('  also', a.get('workload'), a.get('value'), a.get('ms_per_step'), a.get('steps'), a.get('warmup'))
"; }
for l in config2,config1_fp32 config3,config4,config2,config1_fp32 config2,config4,config3; do echo "== also-list $l"; python bench.py --no-cpu-baseline --also-list $l 2>/dev/null | pick; done
echo "== AMD_LOG_LEVEL probe skipped"
