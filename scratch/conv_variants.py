import sys, torch
sys.path.insert(0, ".")
from oracle import m2trans_oracle as O
from tests.gpu_util import build_model
from m2trans_amd import _lib
tag, persistent = sys.argv[1], int(sys.argv[2])
scale, nb, B, H, W = 4, 2, 4, 128, 128
x = O.closed_form_image(B, 3, H, W).cuda()
model, _ = build_model(scale, nb, "bf16")
plan = model._plan_for(x)
_lib.check(_lib.load().m2t_set_option(plan.handle, b"persistent_conv", persistent), "opt")
with torch.no_grad():
    sr = model(x)
    sr2 = model(x)
print(tag, "deterministic", torch.equal(sr, sr2), float(sr.double().sum()))
torch.save(sr.cpu(), f"gpurun_out/sr_{tag}.pt")
