cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 | tee $O/pytest.log
for i in 1 2; do
for v in 1 0; do
python bench.py --no-cpu-baseline --no-kernel-events --steps 20 --option conv_rows=$v 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('conv_rows=$v', d['value'], d['ms_per_step'])"
done; done
python bench.py --no-cpu-baseline --config 3 --no-kernel-events --steps 20 --option conv_rows=1 2>&1 | tail -1 | cut -c1-150
python bench.py --no-cpu-baseline --config 3 --no-kernel-events --steps 20 --option conv_rows=0 2>&1 | tail -1 | cut -c1-150
python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print(d['value'], d['ms_per_step'], r['category'], r['avg_launch_us'], r['frac'], [(o['category'], o['avg_launch_us']) for o in r['others']])"
