cd $GRAFT_REPO_ROOT
cat > /tmp/grads.py <<'PY'
import sys, torch
sys.path.insert(0, '.')
from tests.gpu_util import build_model
from oracle import m2trans_oracle as O
out = {}
for (B, H, W) in ((4, 128, 128), (2, 60, 90), (3, 96, 160)):
    x = O.closed_form_image(B, 3, H, W).cuda(); hr = O.closed_form_image(B, 3, 4 * H, 4 * W, phase=0.7).cuda()
    model, _ = build_model(4, 2, "bf16")
    sr = model(x); torch.nn.L1Loss()(sr, hr).backward()
    out[(B, H, W)] = (sr.detach().cpu(), torch.cat([q.grad.reshape(-1) for _, q in model.named_parameters() if q.requires_grad]).cpu())
torch.save(out, sys.argv[1])
PY
cp m2trans_amd/libm2t.so /tmp/lib_keep.so
cp scratch/libA.so m2trans_amd/libm2t.so; python /tmp/grads.py /tmp/a.pt
cp scratch/libB.so m2trans_amd/libm2t.so; python /tmp/grads.py /tmp/b.pt
cp /tmp/lib_keep.so m2trans_amd/libm2t.so
cmp scratch/libA.so scratch/libB.so > /dev/null && echo "WARNING: the two builds are the same file"
python - <<'PY'
import torch
a, b = torch.load('/tmp/a.pt'), torch.load('/tmp/b.pt')
for k in a:
    print(k, 'sr equal', torch.equal(a[k][0], b[k][0]), 'grads equal', torch.equal(a[k][1], b[k][1]))
PY
bash tools/ab_libs.sh "--steps 20" 3 2>&1 | tail -6
bash tools/ab_libs.sh "--steps 20 --config 3" 1 2>&1 | tail -2
