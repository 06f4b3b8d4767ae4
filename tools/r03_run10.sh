cd $GRAFT_REPO_ROOT
O=gpurun_out/r03j; mkdir -p $O
ab() { python bench.py --no-cpu-baseline --no-kernel-events --steps 20 $2 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
ab gate1 ""
ab gate2 "--option gate_branch=2"
ab gate0 "--option gate_branch=0"
ab gate3 "--option gate_branch=3"
ab ungated "--option gate_branch=-1"
ab conv4 "--option conv_rows=4"
done
ab c3_gate1 "--config 3"
ab c3_gate2 "--config 3 --option gate_branch=2"
ab c3_ungated "--config 3 --option gate_branch=-1"
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -x -k "row_streaming or fast_kernels" 2>&1 | tail -3
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --force-comm-path --no-cpu-baseline --steps 10 2>&1 | tail -1 | cut -c1-700
