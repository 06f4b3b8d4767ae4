#!/bin/bash
# fp32 parity mode after the 32x32x2 GEMM kernels: the fp32 tests, then config 1 fp32 with the option on / off on the same box
set -x
OUT=gpurun_out/r05f32; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q -k "fp32 or oracle or psnr or adam or golden" 2>&1 | tail -8 | tee $OUT/pytest.txt
for r in 1 2; do
  for o in 1 0; do
    timeout 600 python bench.py --dtype fp32 --no-also --no-cpu-baseline --no-kernel-events --steps 8 --warmup 2 --option fp32_fast=$o 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fp32_fast=$o', d['value'], d['ms_per_step'])" | tee -a $OUT/ab.txt
  done
done
