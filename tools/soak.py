#!/usr/bin/env python3
"""Soak / race detector (GPU box): R runs of S identical bf16 training steps at the bench configuration on a
non-default stream; every run must end with bit-identical parameters (no atomics, every cross-stream hand-over an event).
    python tools/soak.py [runs=8] [steps=12]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.gpu_util import build_model
from oracle import m2trans_oracle as O          # closed-form data only (this is a test tool, not the product path)
from m2trans_amd.train_step import TrainStep

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
scale, nb, B, H, W = 4, 8, 16, 128, 128
data = [(O.closed_form_image(B, 3, H, W, phase=0.2 * s).cuda(), O.closed_form_image(B, 3, H * scale, W * scale, phase=0.5 + 0.2 * s).cuda())
        for s in range(4)]
ref = None
t0 = time.time()
for r in range(runs):
    model, _ = build_model(scale, nb, "bf16")
    ts = TrainStep(model, lr=1e-4, lambda_l1=1.0, world_size=1)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for s in range(steps):
            loss = ts.step(*data[s % 4])
        st.synchronize()
    cur = (float(loss), model.flat_params.detach().clone())
    if ref is None:
        ref = cur
    same = cur[0] == ref[0] and torch.equal(cur[1], ref[1])
    print(f"run {r}: loss {cur[0]:.9f} {'identical' if same else 'DIFFERENT'}", flush=True)
    if not same:
        d = (cur[1] - ref[1]).abs()
        print("  max abs diff", float(d.max()), "at", int(d.argmax()))
        sys.exit(1)
print(f"{runs} runs x {steps} steps identical ({time.time() - t0:.1f} s)")
