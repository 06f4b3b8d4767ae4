cd $GRAFT_REPO_ROOT
python - <<'PY'
# gradient parity of the LDS-DMA weight-gradient kernel against the default kernels on a whole step (bf16, batch 4, 128x128, 2 blocks)
import torch, sys
sys.path.insert(0, '.')
from tests.gpu_util import build_model
from oracle import m2trans_oracle as O
from m2trans_amd import _lib
outs = []
for opt in (-1, -2):
    x = O.closed_form_image(4, 3, 128, 128).cuda(); hr = O.closed_form_image(4, 3, 512, 512, phase=0.7).cuda()
    model, _ = build_model(4, 2, "bf16")
    plan = model._plan_for(x)
    _lib.check(_lib.load().m2t_set_option(plan.handle, b"wgrad_big_tiles", opt), "opt")
    sr = model(x); torch.nn.L1Loss()(sr, hr).backward()
    outs.append({n: q.grad.double().cpu() for n, q in model.named_parameters() if q.requires_grad})
worst = 0
for n in outs[0]:
    d = float((outs[0][n] - outs[1][n]).norm() / (outs[0][n].norm() + 1e-30))
    if 'qkv' in n: print(n, '%.2e' % d)
    worst = max(worst, d)
print('worst rel diff over all tensors: %.2e' % worst)
PY
ab() { python bench.py --no-cpu-baseline --no-kernel-events --steps 20 $2 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
ab default ""
ab dma "--option wgrad_big_tiles=-2"
done
ab c3_default "--config 3"
ab c3_dma "--config 3 --option wgrad_big_tiles=-2"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r03h
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03h -- python3 bench.py --steps 5 --warmup 2 --no-kernel-events --no-side-stream --no-cpu-baseline --option wgrad_big_tiles=-2 > /dev/null 2>&1
python3 - <<PY
import csv, glob
f=glob.glob('gpurun_out/r03h/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'wgrad_tn' in r['Name']: print(r['Name'][:44], r['Calls'], '%.1f us avg  min %.1f max %.1f'%(float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
rm -rf gpurun_out/r03h
