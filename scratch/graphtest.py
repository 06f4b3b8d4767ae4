import sys, time, types, torch
sys.path.insert(0, '.')
from m2trans_amd import _lib
from m2trans_amd.M2Trans_network import create_model
from m2trans_amd.train_step import TrainStep
torch.manual_seed(33)
model = create_model(types.SimpleNamespace(n_feats=64, scale=4, rgb_range=1.0, n_blocks=8, colors=3, compute_dtype="bf16")).cuda()
B = 16
g = torch.Generator(device="cuda"); g.manual_seed(1)
lr = torch.rand(B, 3, 128, 128, generator=g, device="cuda"); hr = torch.rand(B, 3, 512, 512, generator=g, device="cuda")
for side in (1, 0):
    ts = TrainStep(model, world_size=1)
    plan = model._plan_for(lr)
    _lib.check(_lib.load().m2t_set_option(plan.handle, b"side_stream", side))
    for _ in range(3): ts.step(lr, hr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): ts.step(lr, hr)
    torch.cuda.synchronize()
    print(f"side={side} eager: {(time.perf_counter()-t0)/10*1e3:.3f} ms/step", flush=True)
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            ts.step(lr, hr)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            ts.step(lr, hr)
        torch.cuda.synchronize()
        for _ in range(3): gr.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): gr.replay()
        torch.cuda.synchronize()
        print(f"side={side} graph: {(time.perf_counter()-t0)/10*1e3:.3f} ms/step", flush=True)
    except Exception as e:
        print("graph capture failed:", repr(e)[:300], flush=True)
