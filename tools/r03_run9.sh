cd $GRAFT_REPO_ROOT
O=gpurun_out/r03i; mkdir -p $O
timeout 300 ./scratch/bench_conv_rows 2>&1 | tee $O/bench_conv_rows.log | grep -v "^B=1 \|^B=2 \|^B=3 "
ab() { env $3 python bench.py --no-cpu-baseline --no-kernel-events --steps 20 $2 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
ab default "" "A=1"
ab slabs10 "" "M2T_WGRAD_SLABS=10"
ab slabs16 "" "M2T_WGRAD_SLABS=16"
ab conv_pipe3 "--option conv_rows=3" "A=1"
ab conv_d2 "--option conv_rows=2" "A=1"
done
python bench.py --no-cpu-baseline --all-kernel-events 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print(d['value'], d['ms_per_step'], r['category'], r['avg_launch_us'], r['frac']); print([(o['category'], o['avg_launch_us'], o['total_ms']) for o in r['others']])"
