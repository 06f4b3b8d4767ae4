# experiment: half-batches driven by 2 host threads on 2 streams (kernel-granularity interleave)
import sys, time, types, torch, threading
sys.path.insert(0, '.')
from m2trans_amd import _lib
from m2trans_amd.M2Trans_network import create_model, Plan
lib = _lib.load()
torch.manual_seed(33)
model = create_model(types.SimpleNamespace(n_feats=64, scale=4, rgb_range=1.0, n_blocks=8, colors=3, compute_dtype="bf16")).cuda()
B = 16
g = torch.Generator(device="cuda"); g.manual_seed(1)
lr = torch.rand(B, 3, 128, 128, generator=g, device="cuda"); hr = torch.rand(B, 3, 512, 512, generator=g, device="cuda")
dev = lr.device

def run(nsplit, steps=10):
    h = B // nsplit
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    parts = [(lr[i*h:(i+1)*h].contiguous(), hr[i*h:(i+1)*h].contiguous()) for i in range(nsplit)]
    grads = [torch.zeros_like(model.flat_params) for _ in range(nsplit)]
    losses = [torch.zeros(1, device="cuda") for _ in range(nsplit)]
    plans = [Plan(h, 128, 128, 4, 8, _lib.BF16, dev) for _ in range(nsplit)]
    m = torch.zeros_like(model.flat_params); v = torch.zeros_like(model.flat_params)
    div = float(hr.numel())
    go = [threading.Event() for _ in range(nsplit)]; done = [threading.Event() for _ in range(nsplit)]
    stop = [False]
    def worker(i):
        torch.cuda.set_device(dev)
        while True:
            go[i].wait(); go[i].clear()
            if stop[0]: return
            with torch.cuda.stream(streams[i]):
                st = _lib.stream_ptr(); ws = _lib.ptr(plans[i].workspace)
                _lib.check(lib.m2t_forward(plans[i].handle, _lib.ptr(model.flat_params), _lib.ptr(parts[i][0]), None, 1.0, 1, ws, st))
                _lib.check(lib.m2t_l1_loss(plans[i].handle, _lib.ptr(parts[i][1]), 1.0, div, 1.0, _lib.ptr(losses[i]), ws, st))
                _lib.check(lib.m2t_backward(plans[i].handle, _lib.ptr(model.flat_params), _lib.ptr(parts[i][0]), _lib.ptr(grads[i]), ws, st))
            done[i].set()
    ths = [threading.Thread(target=worker, args=(i,), daemon=True) for i in range(nsplit)]
    for t in ths: t.start()
    def step(k):
        cur = torch.cuda.current_stream()
        for s in streams: s.wait_stream(cur)
        for i in range(nsplit): go[i].set()
        for i in range(nsplit): done[i].wait(); done[i].clear()
        for s in streams: cur.wait_stream(s)
        gsum = grads[0]
        for i in range(1, nsplit): gsum = gsum + grads[i]
        _lib.check(lib.m2t_adam_step(_lib.ptr(model.flat_params), _lib.ptr(gsum), _lib.ptr(m), _lib.ptr(v), gsum.numel(), 1e-4, 0.9, 0.999, 1e-8, k, 1.0, _lib.stream_ptr()))
    for k in range(3): step(k+1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps): step(k+4)
    torch.cuda.synchronize(); dt = (time.perf_counter()-t0)/steps
    print(f"threads/streams {nsplit}: {dt*1e3:.3f} ms/step  {B/dt:.0f} patches/s  loss {sum(float(l) for l in losses):.5f}", flush=True)
    stop[0] = True
    for i in range(nsplit): go[i].set()
for n in (1, 2, 4):
    run(n)
