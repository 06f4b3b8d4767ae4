cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -x -k "tail or fused or bf16_fast or bitwise or golden" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_baseline_configs.py -q -m gpu -x -k "small or config1" 2>&1 | tail -3
bash tools/ab_opts.sh "--option fused_tail_fwd=1" "--option fused_tail_fwd=0" 3 2>&1 | tee $O/ab_tail.log
python bench.py --no-cpu-baseline --all-kernel-events --no-side-stream 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('single-stream', d['value'], d['ms_per_step'], r['category'], r['avg_launch_us'], r['frac']); print([(o['category'], o['avg_launch_us']) for o in r['others']])"
