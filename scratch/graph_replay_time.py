"""Eager steps vs replays of ONE captured step (bench configuration): is the launch / fork overhead worth a graph?
   python scratch/graph_replay_time.py [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.gpu_util import build_model
from oracle import m2trans_oracle as O          # closed-form data only (a probe, not the product path)
from m2trans_amd.train_step import TrainStep

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
scale, nb, H, W = 4, 8, 128, 128
x = O.closed_form_image(B, 3, H, W).cuda()
hr = O.closed_form_image(B, 3, H * scale, W * scale, phase=0.7).cuda()
model, _ = build_model(scale, nb, "bf16")
ts = TrainStep(model, lr=1e-4, world_size=1)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        ts.step(x, hr)
    s.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ts.step(x, hr)
    s.synchronize()
    eager = (time.perf_counter() - t0) / 20
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        ts.step(x, hr)
    for _ in range(3):
        g.replay()
    s.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    s.synchronize()
    graph = (time.perf_counter() - t0) / 20
print(f"batch {B}: eager {eager * 1e3:.3f} ms / step, graph replay {graph * 1e3:.3f} ms / step")
