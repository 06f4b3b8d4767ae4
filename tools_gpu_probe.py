# scratch: print bf16 error table
import torch, sys
sys.path.insert(0,'.')
from oracle import m2trans_oracle as O
from tests.gpu_util import build_model, rms_rel, rel
scale, nb, B, H, W = 4, 2, 2, 32, 32
model, p = build_model(scale, nb, "bf16")
x = O.closed_form_image(B, 3, H, W); hr = O.closed_form_image(B, 3, H*scale, W*scale, phase=0.7)
loss_o, sr_o, g_o = O.l1_loss_and_grads(x, hr, p, scale, nb)
sr = model(x.cuda()); loss = torch.nn.L1Loss()(sr, hr.cuda()); loss.backward()
print("sr rms_rel", rms_rel(sr, sr_o), "max", rel(sr, sr_o), "sr_o absmax", float(sr_o.abs().max()), "frac zero", float((sr_o==0).float().mean()))
print("loss", float(loss), float(loss_o))
for n,q in model.named_parameters():
    if q.requires_grad: print(f"{n:40s} rms {rms_rel(q.grad,g_o[n]):.3e} max {rel(q.grad,g_o[n]):.3e}")
