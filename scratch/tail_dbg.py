"""Read the per-phase timestamps of the fused tail backward (debug build with M2T_TAIL_DBG=1)."""
import os, sys, torch
os.environ["M2T_TAIL_DBG"] = sys.argv[1] if len(sys.argv) > 1 else "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gpu_util import build_model
import oracle.m2trans_oracle as O
B, H = 16, 128
model, _ = build_model(4, 1, "bf16")
x = torch.rand(B, 3, H, H, device="cuda"); hr = torch.rand(B, 3, 4 * H, 4 * H, device="cuda")
for _ in range(2):
    sr = model(x); torch.nn.L1Loss()(sr, hr).backward()
torch.cuda.synchronize()
plan = model._plan_for(x)
arena = plan.ws_tensor("arena", dtype=torch.float32)
# slab_b3 is the third arena allocation of the backward: head_cols? no -- find by scanning for plausible timestamps
a = arena.view(torch.int64).cpu()
import numpy as np
v = a.numpy()
idx = np.where((v > 1e11) & (v < 1e16))[0]
print("candidates", len(idx))
groups = []
for i in idx:
    if not groups or i - groups[-1][-1] > 1: groups.append([i])
    else: groups[-1].append(i)
for gr in groups:
    for base in range(gr[0], gr[-1] + 1, 8):
        t = v[base:base + 8]
        print("at", base, [int(t[i + 1] - t[i]) for i in range(7)], "total", int(t[7] - t[0]))
