#!/bin/bash
# the REAL N-rank control flow of bench.py on a 1-GPU box: 2 (and 4) ranks share device 0 and exchange over gloo
O=gpurun_out/r05self; mkdir -p $O
for n in 2 4; do
  timeout 600 python bench.py --gpus $n --selftest-shared-device --steps 5 --warmup 2 > $O/n$n.json 2> $O/n$n.err; echo "rc=$?"
  tail -3 $O/n$n.err | cut -c1-300
  python3 - <<P
import json
try:
    d=json.loads(open('$O/n$n.json').read().strip().splitlines()[-1])
    print('n_gpus', d['n_gpus'], 'invalid', d.get('invalid'), 'ms/step', d['ms_per_step'], 'pg_ranks', d['config']['world_size'], 'devices', d['rank_devices'], 'backend', d['config']['backend'], 'exposed', d.get('exposed_comm_ms_per_step'), 'global_batch', d['config']['global_batch'])
    print('also', [(a.get('workload'), a.get('n_gpus'), a.get('global_batch'), a.get('ms_per_step'), a.get('error')) for a in d.get('also',[])])
    print('others', [o['category'] for o in (d.get('roofline') or {}).get('others',[])])
except Exception as e: print('parse failed', e); print(open('$O/n$n.json').read()[-500:])
P
done
