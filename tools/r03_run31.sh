cd $GRAFT_REPO_ROOT
ab() { python bench.py --no-cpu-baseline --no-kernel-events --steps 20 $2 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
ab c1_rows_default ""
ab c1_rows_pipe "--option conv_rows=4"
ab c4_rows_default "--config 4"
ab c4_rows_pipe "--config 4 --option conv_rows=4"
ab c3_rows_default "--config 3"
ab c3_rows_pipe "--config 3 --option conv_rows=4"
done
