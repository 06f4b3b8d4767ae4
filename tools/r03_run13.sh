cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -x -k "conv or bitwise or fast_kernels" 2>&1 | tail -5
ab() { python bench.py --no-cpu-baseline --no-kernel-events --steps 20 $2 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
ab fusedconv ""
ab separate "--option fused_conv_bwd=0"
ab fused_gate2 "--option gate_branch=2"
ab fused_ungated "--option gate_branch=-1"
done
ab c3_fused "--config 3"
ab c3_separate "--config 3 --option fused_conv_bwd=0"
python bench.py --no-cpu-baseline --steps 10 2>&1 | tail -1 | cut -c1-3000
