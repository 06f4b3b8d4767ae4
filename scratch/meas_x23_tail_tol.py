"""Measure what the x2 / x3 row-streaming tail backward really differs by from the plain kernels (to set the test's tolerances)."""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import m2trans_oracle as O
from tests.gpu_util import build_model
from m2trans_amd import _lib
for (scale, B, H, W) in [(3, 2, 40, 56), (3, 1, 96, 128), (2, 2, 60, 90), (2, 1, 128, 64), (3, 2, 32, 32), (2, 1, 32, 64), (3, 1, 160, 64), (2, 1, 192, 32)]:
    x = O.closed_form_image(B, 3, H, W).cuda()
    hr = O.closed_form_image(B, 3, H * scale, W * scale, phase=0.7).cuda()
    outs = []
    for fused in (3, 0):
        model, _ = build_model(scale, 1, "bf16")
        plan = model._plan_for(x)
        _lib.check(_lib.load().m2t_set_option(plan.handle, b"fused_tail", fused), "opt")
        sr = model(x)
        torch.nn.L1Loss()(sr, hr).backward()
        torch.cuda.synchronize()
        gT = plan.ws_tensor("gT").float().clone()
        outs.append((sr.detach().clone(), {n: q.grad.detach().double().cpu() for n, q in model.named_parameters() if q.requires_grad}, gT))
    (sa, ga, ta), (sb, gb, tb) = outs
    total = math.sqrt(sum(float(v.pow(2).sum()) for v in gb.values()))
    diff = math.sqrt(sum(float((ga[n] - gb[n]).pow(2).sum()) for n in gb))
    worst = max((float((ga[n] - gb[n]).norm()) / max(float(gb[n].norm()), 1e-30), n) for n in gb if float(gb[n].norm()) > 1e-5 * total)
    tail = {n: float((ga[n] - gb[n]).norm()) / max(float(gb[n].norm()), 1e-30) for n in gb if n.startswith("tail")}
    print(scale, B, H, W, "sr equal", torch.equal(sa, sb), "gT rel L2 %.2e max %.2e frac!=0 %.4f" % (float((ta - tb).norm() / tb.norm()), float((ta - tb).abs().max() / tb.abs().max()), float(((ta - tb) != 0).float().mean())),
          "whole %.2e worst-param %.2e %s" % (diff / total, worst[0], worst[1]), "tail", {k: "%.1e" % v for k, v in tail.items()})
