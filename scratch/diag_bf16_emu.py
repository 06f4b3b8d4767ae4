"""Diagnostic (GPU box): where does the bf16 HIP forward leave the bf16-rounding oracle?  Prints per-intermediate
max-rel / rms-rel of HIP vs oracle(emulate_bf16=True) and vs the fp32 oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import m2trans_oracle as O
from tests.gpu_util import build_model, rel, rms_rel, ws_nchw
from tests.test_gpu_model import _oracle_forward_padded, BR_C, BR_L

def table(scale, nb, B, H, W):
    model, p = build_model(scale, nb, "bf16")
    x = O.closed_form_image(B, 3, H, W)
    cap_e, cap_f = {}, {}
    with torch.no_grad():
        O.forward(x, p, scale, nb, return_preclamp=True, cap=cap_e, emulate_bf16=True)
        O.forward(x, p, scale, nb, return_preclamp=True, cap=cap_f)
        sr = model(x.cuda())
    torch.cuda.synchronize()
    plan = model._plan_for(x.cuda())
    names = [("X0", 64, 0)]
    for b in range(nb):
        for i in range(4):
            names += [(f"b{b}.d{i+1}", BR_C[i], BR_L[i]), (f"b{b}.qkv{i+1}", 3 * BR_C[i], BR_L[i])]
        names += [(f"b{b}.xc", 64, 0), (f"X{b+1}", 64, 0)]
    r0 = 2 if scale == 4 else scale
    print(f"--- x{scale} nb={nb} B={B} {H}x{W}: name, vs emulated (max-rel, rms-rel), vs fp32 (max-rel, rms-rel)")
    for nm, C, L in names:
        got = ws_nchw(plan, nm, B, H >> L, W >> L, C)
        print(f"{nm:10s} {rel(got, cap_e[nm]):.3e} {rms_rel(got, cap_e[nm]):.3e}   {rel(got, cap_f[nm]):.3e} {rms_rel(got, cap_f[nm]):.3e}")
    for nm, f in (("t1act", r0), ("t1der", r0)) + ((("t2act", 4), ("t2der", 4)) if scale == 4 else ()):
        got = ws_nchw(plan, nm, B, H * f, W * f, 64)
        print(f"{nm:10s} {rel(got, cap_e[nm]):.3e} {rms_rel(got, cap_e[nm]):.3e}   {rel(got, cap_f[nm]):.3e} {rms_rel(got, cap_f[nm]):.3e}")
    pre = plan.ws_tensor("srpre", dtype=torch.float32).view(B, 3, H * scale, W * scale).cpu()
    print(f"{'srpre':10s} {rel(pre, cap_e['srpre']):.3e} {rms_rel(pre, cap_e['srpre']):.3e}   {rel(pre, cap_f['srpre']):.3e} {rms_rel(pre, cap_f['srpre']):.3e}")
    print("srpre rms", float(cap_f['srpre'].pow(2).mean().sqrt()), "sr rms", float(sr.pow(2).mean().sqrt()))

pass
pass

from tests.test_gpu_baseline_configs import fwd_bwd, grad_table, fmt

# ---- teacher-forced comparison: stage errors and gradient table ----
from tests.gpu_util import hip_forward_trace
for (scale, nb, B, lr) in ((4, 2, 2, 32), (3, 2, 2, 32), (4, 8, 2, 128), (3, 8, 1, 256)):
    model, p = build_model(scale, nb, "bf16")
    x = O.closed_form_image(B, 3, lr, lr); hr = O.closed_form_image(B, 3, lr * scale, lr * scale, phase=0.7)
    sr, loss, grads = fwd_bwd(model, x.cuda(), hr.cuda(), hr.numel())
    plan = model._plan_for(x.cuda())
    trace = hip_forward_trace(plan, scale, nb, B, plan.query("padded_h"), plan.query("padded_w"))
    rep = {}
    loss_o, sr_o, g_o = O.l1_loss_and_grads(x, hr, p, scale, nb, emulate_bf16=True, force=trace, stage_report=rep)
    print(f"=== forced x{scale} nb={nb} B={B} {lr}: sr rel {rel(sr, sr_o):.3e} loss {loss} {float(loss_o)}")
    items = sorted(rep.items(), key=lambda kv: -kv[1][1])
    print("worst stages by rms:", [(k, f"{v[0]:.2e}", f"{v[1]:.2e}") for k, v in items[:8]])
    items = sorted(rep.items(), key=lambda kv: -kv[1][0])
    print("worst stages by max:", [(k, f"{v[0]:.2e}", f"{v[1]:.2e}") for k, v in items[:8]])
    rows = grad_table(model, grads, g_o)
    rows.sort(key=lambda r: -min(r[1] / 2e-2, r[2] / 1e-3))
    print(fmt(rows[:14]))
    rows.sort(key=lambda r: -r[1])
    print("-- by rel:"); print(fmt(rows[:8]))
