#!/usr/bin/env python3
"""N data-parallel ranks through the real TrainStep:
  * overlapped (bucketed, communication stream) and plain gradient exchange must give bit-identical parameters;
  * both must match a single-process step on the full global batch (fp32 compute; differences = summation order).
Backend: RCCL ("nccl"), one rank per GPU, whenever the node has at least WORLD_SIZE devices; otherwise every rank
shares GPU 0 and the exchange runs over gloo on device tensors (the 1-GPU development boxes).  M2T_DP_BACKEND overrides.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/dp2_check.py
"""
import os, sys, types
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    backend = os.environ.get("M2T_DP_BACKEND") or ("nccl" if torch.cuda.device_count() >= world else "gloo")
    if backend == "nccl":                                   # "nccl" is RCCL on ROCm; one rank per GPU over xGMI
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from m2trans_amd.M2Trans_network import create_model
    from m2trans_amd.train_step import TrainStep
    scale, nb, Bl, H = 4, 2, 2, 64
    args = types.SimpleNamespace(n_feats=64, scale=scale, rgb_range=1.0, n_blocks=nb, colors=3, compute_dtype="fp32")
    g = torch.Generator().manual_seed(7)
    xs = [torch.rand(world * Bl, 3, H, H, generator=g) for _ in range(2)]
    hs = [torch.rand(world * Bl, 3, H * scale, H * scale, generator=g) for _ in range(2)]
    finals = {}
    for mode in ("overlap", "plain", "single"):
        torch.manual_seed(33)
        model = create_model(args).to("cuda")
        w = 1 if mode == "single" else world
        ts = TrainStep(model, lr=1e-3, world_size=w, overlap_comm=(mode == "overlap"))
        assert ts.overlap_comm == (mode == "overlap")
        torch.cuda.synchronize()
        torch.cuda.set_stream(torch.cuda.Stream())
        for i in range(2):
            if mode == "single":
                ts.step(xs[i].cuda(), hs[i].cuda())
            else:
                sl = slice(rank * Bl, (rank + 1) * Bl)
                ts.step(xs[i][sl].cuda(), hs[i][sl].cuda())
        torch.cuda.synchronize()
        finals[mode] = model.flat_params.detach().float().cpu().clone()
    same = torch.equal(finals["overlap"], finals["plain"])
    d = float((finals["overlap"] - finals["single"]).abs().max())
    moved = float((finals["single"] - create_model(args).flat_params.cpu()).abs().max()) if False else None
    # every rank must hold the same parameters
    mine = finals["overlap"].cuda() if backend == "nccl" else finals["overlap"]
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    replicas_equal = all(torch.equal(gathered[0], t) for t in gathered)
    if rank == 0:
        print(f"dp2_check[{backend}, {world} ranks]: overlap == plain: {same}; replicas identical: {replicas_equal}; max |DP - single process| = {d:.3e}")
    ok = same and replicas_equal and d < 5e-6
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
