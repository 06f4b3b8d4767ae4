cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -x 2>&1 | tail -5
ab() { python bench.py --no-cpu-baseline --no-kernel-events --steps 20 $2 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
ab c64_recompute ""
ab c64_stored "--option fused_attn_fwd=1"
done
ab c3_recompute "--config 3"
ab c3_stored "--config 3 --option fused_attn_fwd=1"
