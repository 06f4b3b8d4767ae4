cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x -k "not drift_at" 2>&1 | tail -6 | tee $O/pytest.log
ab() { python bench.py --no-cpu-baseline --no-kernel-events --steps 20 $2 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
ab default ""
ab fused_tail_fwd "--option fused_tail_fwd=1"
ab conv_d2 "--option conv_rows=2"
ab nosidestream "--no-side-stream"
done
ab c3_default "--config 3"
ab c3_fused_tail "--config 3 --option fused_tail_fwd=1"
