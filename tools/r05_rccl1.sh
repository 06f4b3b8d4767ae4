#!/bin/bash
# one-rank RCCL line (the real collectives, comm stream) with the runtime's default of 4 hardware queues and with 8
pick() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('hw_queues'), d.get('exposed_comm_ms_per_step'))"; }
for r in 1 2 3; do for q in 4 8; do
GPU_MAX_HW_QUEUES=$q python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2954$r bench.py --gpus 1 --force-comm-path --no-cpu-baseline --no-also --no-kernel-events --steps 20 2>/dev/null | pick
done; done
