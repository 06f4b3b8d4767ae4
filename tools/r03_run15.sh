cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -x -k "c16 or fast_kernels or projection or bitwise" 2>&1 | tail -5
ab() { python bench.py --no-cpu-baseline --no-kernel-events --steps 20 $2 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
ab c16prep ""
ab three_kernels "--option attn_bwd=2"
done
ab c3_c16prep "--config 3"
ab c3_three "--config 3 --option attn_bwd=2"
