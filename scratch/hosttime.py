import sys, time, types, torch
sys.path.insert(0, '.')
from m2trans_amd import _lib
from m2trans_amd.M2Trans_network import create_model
from m2trans_amd.train_step import TrainStep
torch.manual_seed(33)
model = create_model(types.SimpleNamespace(n_feats=64, scale=4, rgb_range=1.0, n_blocks=8, colors=3, compute_dtype="bf16")).cuda()
for B in (16, 4):
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    lr = torch.rand(B, 3, 128, 128, generator=g, device="cuda"); hr = torch.rand(B, 3, 512, 512, generator=g, device="cuda")
    ts = TrainStep(model, world_size=1)
    for _ in range(3): ts.step(lr, hr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): ts.step(lr, hr)
    t_issue = (time.perf_counter() - t0) / 10
    torch.cuda.synchronize()
    t_total = (time.perf_counter() - t0) / 10
    print(f"B={B}: host issue {t_issue*1e3:.3f} ms/step, wall {t_total*1e3:.3f} ms/step")
    # one step at a time (sync between): pure latency
    t0 = time.perf_counter()
    for _ in range(5):
        ts.step(lr, hr); torch.cuda.synchronize()
    print(f"   synced per step: {(time.perf_counter()-t0)/5*1e3:.3f} ms")
