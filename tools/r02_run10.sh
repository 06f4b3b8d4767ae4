cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_swin.py -q -m gpu -x 2>&1 | tail -3
for i in 1 2 3; do python bench.py --config 2 --no-cpu-baseline --no-kernel-events --steps 10 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
python bench.py --config 2 --no-cpu-baseline --no-kernel-events --steps 10 --no-overlap-semantic 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('serial', d['value'], d['ms_per_step'])"
bash tools/swin_trace.sh > /dev/null; grep "gemm_nt" gpurun_out/swin_trace.txt | awk '{s+=$2} END {print "gemm total us", s}'; grep -c gemm_nt_wide gpurun_out/swin_trace.txt
