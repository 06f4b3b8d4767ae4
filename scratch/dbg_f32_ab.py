"""fp32_fast = 1 against 0: per-parameter relative gradient difference (debug helper for the A/B test)."""
import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from gpu_util import build_model
import oracle.m2trans_oracle as O
from m2trans_amd import _lib
scale, B, H, W = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (4, 2, 64, 64)))
x = O.closed_form_image(B, 3, H, W).cuda()
hr = O.closed_form_image(B, 3, H * scale, W * scale, phase=0.7).cuda()
outs = []
for fast in (1, 0):
    model, _ = build_model(scale, 2, "fp32")
    plan = model._plan_for(x)
    _lib.check(_lib.load().m2t_set_option(plan.handle, b"fp32_fast", fast), "m2t_set_option")
    sr = model(x)
    ((sr - hr) ** 2).mean().backward()
    outs.append((sr.detach().clone(), {n: q.grad.clone() for n, q in model.named_parameters() if q.requires_grad}))
(sa, ga), (sb, gb) = outs
print("sr", float((sa - sb).abs().max()))
for n in ga:
    d = float((ga[n].double() - gb[n].double()).norm()) / max(float(gb[n].double().norm()), 1e-30)
    if d > 5e-6 or n.startswith("tail"):
        print(n, d, float(gb[n].abs().max()))
