#!/usr/bin/env python3
"""bench.py -- M2Trans x4 train-step throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--dtype bf16|fp32]

One "step" = one pass of the hot path (forward, L1 loss, backward, gradient all-reduce when
N > 1, fused Adam) over one batch of synthetic 128x128 LR / 512x512 HR patches that are already
resident in HBM.  N > 1: one rank per GPU over RCCL, either launched by the caller under
torch.distributed.run, or -- `python bench.py --gpus N` alone -- by this script, which then starts that
launcher as a CHILD process before touching the GPU and exits with its status.  Per-GPU batch is
fixed (weak scaling); value = patches of ALL ranks / max-over-ranks time.

Rank 0 prints ONE JSON line (metric/value/... + "roofline" for the dominant kernel measured
with HIP events inside the timed region + "cpu_baseline": the CPU oracle timed on the host).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402


PRESETS = {1: dict(batch=16, lr_size=128, scale=4, sem=False), 2: dict(batch=32, lr_size=128, scale=4, sem=True),
           3: dict(batch=32, lr_size=128, scale=4, sem=False), 4: dict(batch=8, lr_size=256, scale=3, sem=False)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=None, choices=[1, 2, 3, 4],
                    help="BASELINE.json configs[i] preset: 1 = x4 128x128 batch 16/GPU L1 only (the headline; default at every --gpus N); "
                         "2 = x4 batch 32 + MedCLIP regulariser; 3 = x4 batch 32/GPU (256 on 8 GPUs; runs behind the default line in `also`); "
                         "4 = x3 256x256 LR, batch 8/GPU.  --batch / --lr-size / --scale override the preset")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (configs[1]: 16)")
    ap.add_argument("--lr-size", type=int, default=None)
    ap.add_argument("--scale", type=int, default=None)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not record per-kernel HIP events")
    ap.add_argument("--all-kernel-events", action="store_true", help="time every kernel category in the timed region (adds ~8%% overhead)")
    ap.add_argument("--kernel-event-stride", type=int, default=3,
                    help="HIP events ride on every n-th launch of the dominant kernel (an event-carrying dispatch costs ~10 us: "
                         "16 per step = +2.3%% on the step; 1 = every launch)")
    ap.add_argument("--null-stream", action="store_true", help="run on the legacy default stream instead of a torch stream")
    ap.add_argument("--no-overlap-comm", action="store_true", help="one all-reduce after the backward instead of the bucketed overlap")
    ap.add_argument("--force-comm-path", action="store_true", help="issue the gradient collectives even with one rank (needs an initialised process group)")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=INT", help="m2t_set_option on the plan (experiments)")
    ap.add_argument("--no-side-stream", action="store_true", help="run the whole backward on one stream")
    ap.add_argument("--main-priority", type=int, default=None, choices=[0, -1],
                    help="priority of the stream the step runs on (the plan's side stream inherits it).  Default 0; -1 (high) was tried for "
                         "configs[2] so that the MedCLIP encoder's normal-priority stream runs BEHIND the step's kernels: no gain")
    ap.add_argument("--no-overlap-semantic", action="store_true", help="SemanticLoss forward after the backward pass instead of beside it")
    ap.add_argument("--debug-skip-side", action="store_true", help="TIMING EXPERIMENT: skip all parameter-gradient kernels (wrong results)")
    ap.add_argument("--semantic-loss", action="store_true",
                    help="BASELINE configs[2]: add the MedCLIP(Swin-T) image-text regulariser (random-init tower, hash text features)")
    ap.add_argument("--cpu-baseline-batch", type=int, default=None)
    ap.add_argument("--no-also", action="store_true",
                    help="default run only: skip the short runs of the other BASELINE workloads that fill the line's `also` list")
    ap.add_argument("--also-steps", type=int, default=6, help="timed steps of each `also` run (>= 5)")
    ap.add_argument("--also-list", default=None, metavar="NAME,NAME", help="which `also` workloads to run and in which order (default: all four; diagnosis)")
    ap.add_argument("--selftest-shared-device", action="store_true",
                    help="HARNESS SELF-TEST on a 1-GPU box: the N ranks of --gpus N all compute on device 0 and exchange over gloo, so that the "
                         "REAL N-rank control flow (headline, extra timing steps, `also` workload, collectives in every step) runs end to end; "
                         "the line says `invalid` (ranks share a device)")
    ap.add_argument("--stub-also-fail-rank", type=int, default=None,
                    help="with --stub-step: behind the headline run a stand-in `also` workload in which this rank raises before its collective "
                         "(exercises AlsoWatch: rank 0 still prints the headline, every rank exits 0)")
    ap.add_argument("--stub-step", action="store_true",
                    help="HARNESS SELF-TEST on CPU: run this script's N-rank control flow with gloo and a sleeping stand-in for the step; "
                         "prints an `invalid` line with no throughput")
    args = ap.parse_args()
    # presets = BASELINE.json configs[i]; configs[0] (x2 64x64 CPU forward) is a parity case, not a bench line
    if args.config is None:
        # the SAME per-GPU workload at every N (weak scaling: 16 patches per GPU, BASELINE configs[1]), so that the driver's 1 -> N efficiency
        # compares like with like; configs[3] (32 per GPU = batch 256 on 8 GPUs) rides along in `also` (all ranks run it behind the headline)
        args.config = 1
    preset = PRESETS[args.config]
    args.preset_overridden = any(v is not None for v in (args.batch, args.lr_size, args.scale))
    if args.batch is None:
        args.batch = preset["batch"]
    if args.lr_size is None:
        args.lr_size = preset["lr_size"]
    if args.scale is None:
        args.scale = preset["scale"]
    args.semantic_loss = bool(args.semantic_loss or preset["sem"])
    if args.cpu_baseline_batch is None:
        args.cpu_baseline_batch = 2 if args.lr_size <= 128 else 1
    return args


def source_stamp() -> str:
    """sha256 over the kernel sources the library is built from: profiles/pmc_traffic.json carries the stamp of the
    build its counters were collected on, and roofline.traffic is reported only when it matches this build."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "m2trans_amd", "csrc", "*"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def synthetic_batch(B, h, scale, rank, step, device):
    """U[0,1) LR and HR patches keyed by (seed 33, rank, step) -- train.py:40 uses seed 33."""
    g = torch.Generator(device=device)
    g.manual_seed(33 + 1000003 * rank + 7919 * step)
    lr = torch.rand(B, 3, h, h, generator=g, device=device)
    hr = torch.rand(B, 3, h * scale, h * scale, generator=g, device=device)
    return lr, hr


def usable_cores() -> int:
    """Cores this process may really use: min(affinity mask, cgroup cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(args):
    """The CPU oracle (plain PyTorch restatement pinned to the reference, oracle/) timed on the
    host cores on a bounded sample: the same x4 128x128 train step at a small batch."""
    from oracle import m2trans_oracle as O
    ncores = usable_cores()
    torch.set_num_threads(ncores)
    B = args.cpu_baseline_batch
    p = O.closed_form_params(64, args.scale, 8)
    names = O.trainable_names(p)
    m = {k: torch.zeros_like(p[k]) for k in names}
    v = {k: torch.zeros_like(p[k]) for k in names}
    g = torch.Generator().manual_seed(33)
    x = torch.rand(B, 3, args.lr_size, args.lr_size, generator=g)
    hr = torch.rand(B, 3, args.lr_size * args.scale, args.lr_size * args.scale, generator=g)

    def one(step):
        _, _, gr = O.l1_loss_and_grads(x, hr, p, args.scale, 8)
        for k in names:
            p[k], m[k], v[k] = O.adam_update(p[k], gr[k], m[k], v[k], step, 1e-4)

    tw = time.perf_counter()
    one(1)                          # warm-up
    tw = time.perf_counter() - tw
    t0 = time.perf_counter()
    n = 0
    while n < 1 or (n < 3 and tw < 15.0) or (time.perf_counter() - t0 < 10.0 and n < 8 and tw < 4.0):
        n += 1
        one(n + 1)
    dt = (time.perf_counter() - t0) / n
    return {"value": round(B / dt, 4), "unit": "HR patches/s", "cores": ncores, "kind": "port",
            "sample": f"x{args.scale} {args.lr_size}x{args.lr_size} LR train step (fwd+bwd+Adam, fp32), batch {B}, "
                      f"{n} timed steps after 1 warm-up, torch CPU {torch.get_num_threads()} threads"}


def child_command(argv, n_gpus: int, port=None):
    """The launcher line a plain `python bench.py --gpus N` turns into: one rank per GPU under torch.distributed.run.
    port = None: `--standalone --local-addr 127.0.0.1` (a c10d store on a port the LAUNCHER binds itself -- no probe-then-reuse
    race); an explicit port gives the driver's own line (--master-addr 127.0.0.1 --master-port P)."""
    rdzv = ["--standalone", "--local-addr", "127.0.0.1"] if port is None else ["--master-addr", "127.0.0.1", "--master-port", str(port)]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}"] + rdzv + [os.path.abspath(__file__)] + list(argv)


def visible_gpu_count(sysfs_root=None):
    """GPUs this process could open, counted WITHOUT the HIP runtime (torch.cuda.device_count() falls back to
    hipGetDeviceCount when amdsmi is absent, which initialises HIP in the parent): KFD topology nodes with SIMDs are
    GPUs (CPU nodes have simd_count 0), narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES.
    Returns None when the topology cannot be read (sysfs not mounted, restricted container): UNKNOWN is not zero, and the
    caller then lets the ranks fail with the real error instead of refusing."""
    if sysfs_root is None:
        sysfs_root = os.environ.get("M2T_KFD_TOPOLOGY", "/sys/class/kfd/kfd/topology/nodes")
    n = 0
    try:
        nodes = sorted(os.listdir(sysfs_root))
    except OSError:
        return None
    readable = 0
    for node in nodes:
        try:
            props = dict(l.split(None, 1) for l in open(os.path.join(sysfs_root, node, "properties")).read().splitlines() if " " in l)
        except OSError:
            continue
        readable += 1
        try:
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        except ValueError:
            continue
    if readable == 0:
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            toks = [t.strip() for t in v.split(",") if t.strip() != ""]
            if "-1" in toks:                         # the runtime stops reading the list at -1
                toks = toks[:toks.index("-1")]
            n = min(n, len(toks))
    return n


def rank_environment(base=None, n_gpus: int = 1):
    """Environment of the rank processes.  HSA_ENABLE_IPC_MODE_LEGACY: this pool's host driver only supports dmabuf IPC, and
    without `0` RCCL between processes fails with `hipIpcGetMemHandle: invalid argument`; a value the caller exported wins
    (other drivers may need the legacy mode) and whatever is in force is logged."""
    env = dict(os.environ if base is None else base)
    src = "caller" if "HSA_ENABLE_IPC_MODE_LEGACY" in env else "bench.py default"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // max(1, n_gpus))))
    return env, f"HSA_ENABLE_IPC_MODE_LEGACY={env['HSA_ENABLE_IPC_MODE_LEGACY']} ({src})"


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD process (never an exec), relay rank 0's
    JSON line, return the child's status.  The parent stays strictly GPU-free: devices are counted from sysfs, and it refuses
    only when sysfs POSITIVELY reports fewer GPUs than requested."""
    import subprocess
    if not args.stub_step and not args.selftest_shared_device:
        visible = visible_gpu_count()
        if visible is None:
            print("bench.py: KFD topology not readable, device count unknown: launching the ranks anyway", file=sys.stderr)
        elif visible < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but only {visible} device(s) are visible", file=sys.stderr)
            return 2
    env, ipc = rank_environment(n_gpus=args.gpus)
    cmd = child_command(sys.argv[1:], args.gpus)
    print("bench.py: " + ipc + "; launching " + " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def group_report(device_index: int) -> dict:
    """What a SCALE record needs to be audited: the number of ranks READ BACK from the process group (not the flag), its
    backend, and the device index every rank actually computes on (all-gathered; two ranks on one device = a broken launch)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return {"pg_ranks": 1, "rccl_ranks": None, "backend": None, "rank_devices": [device_index]}
    backend = dist.get_backend()
    on_gpu = backend == "nccl"
    mine = torch.tensor([dist.get_rank(), device_index], dtype=torch.int64, device=torch.device("cuda", device_index) if on_gpu else "cpu")
    alls = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(alls, mine)
    devs = [int(t[1]) for t in sorted(alls, key=lambda t: int(t[0]))]
    n = dist.get_world_size()
    return {"pg_ranks": n, "rccl_ranks": n if on_gpu else None, "backend": backend, "rank_devices": devs,
            "ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}


def stub_main(args, world: int, rank: int) -> None:
    """--stub-step: the N-rank CONTROL FLOW of this script (launcher relay, WORLD_SIZE guard, barrier-bracketed timed region,
    MAX-over-ranks time, rank-0-only JSON line, process-group read-back, exposed-communication accounting) on CPU with gloo and a
    sleeping stand-in for the step.  A self-test of the harness for boxes without GPUs (tests/test_host_cpu.py); its line says
    `invalid` and carries no throughput."""
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus says {args.gpus}")
    grp = group_report(int(os.environ.get("LOCAL_RANK", "0")))
    per_step = 0.01 * (1 + rank)                 # rank r is (r + 1) x slower: the MAX reduction must report the slowest rank
    grad = torch.ones(1024)
    exposed = 0.0

    def step():
        nonlocal exposed
        time.sleep(per_step)
        if world > 1:                            # the stand-in for the gradient exchange: the wait for it is the exposed time
            t = time.perf_counter()
            dist.all_reduce(grad)
            exposed += time.perf_counter() - t

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    exposed = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    own = dt
    if world > 1:
        t = torch.tensor([dt, exposed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, exposed = float(t[0]), float(t[1])
    line = {}
    if rank == 0:
        line = ({"metric": "bench.py control-flow self-test (no GPU work)", "value": None, "invalid": True, "stub": True,
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000.0 * dt / args.steps, 3),
                          "rank0_ms_per_step": round(1000.0 * own / args.steps, 3), "scaling": "weak",
                          "rccl_ranks": grp["rccl_ranks"], "pg_ranks": grp["pg_ranks"], "rank_devices": grp["rank_devices"],
                          "exposed_comm_ms_per_step": round(1000.0 * exposed / args.steps, 3),
                          "config": {"per_gpu_batch": args.batch, "global_batch": world * args.batch, "parallelism": f"dp{world}",
                                     "backend": "gloo", "world_size": dist.get_world_size() if dist.is_initialized() else 1,
                                     "ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}})
    if args.stub_also_fail_rank is not None and world > 1:
        # stand-in for the multi-rank `also` workload: one rank raises before the collective the others are already waiting in
        watch = AlsoWatch(rank, world, line, interval=0.2)
        try:
            watch.start()
            if rank == args.stub_also_fail_rank:
                raise RuntimeError("stub failure in the extra workload")
            dist.all_reduce(grad)
            watch.stop()
        except Exception as e:
            watch.fail(f"{type(e).__name__}: {e}")
    if rank == 0:
        print(json.dumps(finalize_line(line)), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


WORKLOAD_NOTE = {1: "L1 loss only (BASELINE configs[1])", 2: "L1 + MedCLIP(Swin-T) regulariser (BASELINE configs[2])",
                 3: "L1 loss only (BASELINE configs[3]: 32 patches per GPU, 256 on 8 GPUs)",
                 4: "L1 loss only (BASELINE configs[4])"}


def run_workload(args, device, rank: int, world: int, backend, grp: dict, cpu_base: bool) -> dict:
    """Build the model + step driver for `args`, warm up, time EXACTLY args.steps steps between barriers, and return the JSON
    object of the line (rank 0; other ranks return {})."""
    from m2trans_amd import _lib
    from m2trans_amd.M2Trans_network import create_model
    from m2trans_amd.train_step import TrainStep
    from m2trans_amd import profile as m2t_profile
    import types
    _lib.load()
    torch.manual_seed(33)
    margs = types.SimpleNamespace(n_feats=64, scale=args.scale, rgb_range=1.0, n_blocks=8, colors=3,
                                  compute_dtype=args.dtype)
    model = create_model(margs).to(device)
    if world > 1:
        # persistent replicas start from rank 0's weights (the reference re-broadcasts every forward, train.py:73)
        from m2trans_amd.dist import broadcast_params
        broadcast_params(model.flat_params, src=0)
    sem, captions = None, None
    B = args.batch
    if args.semantic_loss:
        from m2trans_amd.losses import SemanticLoss
        sem = SemanticLoss(criterion="l1", N_patches=3, device=device, compute_dtype=args.dtype, max_batch=B, synthetic_text=True)
        sem._enc = None
        from m2trans_amd.losses import SwinEncoder
        e = SwinEncoder(2 * B, _lib.F32 if args.dtype == "fp32" else _lib.BF16, device)
        g = torch.Generator().manual_seed(33)
        state = {n: (torch.randn(k, generator=g) * (0.02 if "weight" in n and "norm" not in n else 0.0)
                     + (1.0 if n.endswith("norm.weight") or ("layernorm" in n and n.endswith("weight")) else 0.0))
                 for n, (o, k) in e.slots.items()}
        del e
        sem.load_image_encoder(state)
        captions = [f"synthetic ultrasound caption {i}" for i in range(B)]
    ts = TrainStep(model, lr=1e-4, lambda_l1=1.0, process_group=None, world_size=world,
                   semantic_loss=sem, lambda_clip=0.01 if sem is not None else 0.0,
                   overlap_comm=not args.no_overlap_comm, force_comm_path=args.force_comm_path,
                   overlap_semantic=not args.no_overlap_semantic)
    batches = [synthetic_batch(B, args.lr_size, args.scale, rank, s, device) for s in range(2)]
    plan = model._plan_for(batches[0][0])
    if args.debug_skip_side:
        _lib.check(_lib.load().m2t_set_option(plan.handle, b"debug_skip_side", 1), "m2t_set_option")
    for kv in args.option:
        key, val = kv.split("=")
        _lib.check(_lib.load().m2t_set_option(plan.handle, key.encode(), int(val)), "m2t_set_option")
    if args.no_side_stream:
        _lib.check(_lib.load().m2t_set_option(plan.handle, b"side_stream", 0), "m2t_set_option")

    dominant_mask = 0
    # a non-default stream: the plan's side stream is a blocking stream and would serialise against the legacy default stream
    # normal priority everywhere: a high-priority launch stream beside the MedCLIP encoder's stream measured no gain (queue priority does not
    # arbitrate CUs between resident kernels), and a SECOND launch stream in the process left the next workload's side stream on a shared
    # hardware queue (the fp32 `also` run behind configs[2]: 872 -> 660 patches/s)
    main_priority = args.main_priority if args.main_priority is not None else 0
    if not args.null_stream:
        torch.cuda.synchronize()
        # ONE launch stream per priority for the whole process: every new stream takes a hardware queue (4 by default), and once the `also`
        # workloads had made a handful of them a plan's main and side stream could land on the SAME queue -- configs[4] then ran 10-40 % slower
        # inside the default line than as its own process (profiles/README.md, round 5)
        key = (str(device), main_priority)
        if key not in _MAIN_STREAMS:
            _MAIN_STREAMS[key] = torch.cuda.Stream(device=device, priority=main_priority)
        torch.cuda.set_stream(_MAIN_STREAMS[key])
    events_on = rank == 0 and not args.no_kernel_events
    for s in range(args.warmup):
        if events_on and s == args.warmup - 1:
            m2t_profile.enable()      # last warm-up step: time every category to find the dominant kernel
        ts.step(*batches[s % 2], captions)
    torch.cuda.synchronize()
    if events_on and args.warmup > 0:
        tt = m2t_profile.read_all()
        dominant_mask = m2t_profile.dominant_mask(tt)                    # dominant single-shape kernel
        m2t_profile.enable(0)
    ts.measure_exposed_comm = ts.bucket is not None                      # events around the compute stream's wait for the exchange
    ts.exposed_comm_events = []
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    stride = 1
    if events_on:
        # HIP events on the launch stream around the dominant kernel only (keeps the timed region honest)
        stride = 1 if (args.all_kernel_events or not dominant_mask) else max(1, args.kernel_event_stride)
        m2t_profile.enable(m2t_profile.ALL_MASK if args.all_kernel_events else (dominant_mask or m2t_profile.ALL_MASK), sample_every=stride)
    t0 = time.perf_counter()
    for s in range(args.steps):
        ts.step(*batches[s % 2], captions)
    host_enqueue_s = time.perf_counter() - t0     # the host's share: how long the launches took to queue (no sync inside)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    exposed_ms = None
    if ts.bucket is not None:
        exposed_ms = sum(a.elapsed_time(b) for a, b in ts.exposed_comm_events) / max(1, args.steps)
    ts.measure_exposed_comm = False
    if world > 1:
        t = torch.tensor([dt, exposed_ms or 0.0], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt, exposed_ms = float(t[0]), float(t[1])
    loss = float(ts.loss)
    roofline = None
    if events_on:
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json" if args.config == 1 else f"pmc_traffic_config{args.config}.json")
        wl = f"config{args.config}" if not args.preset_overridden else None
        roofline = m2t_profile.roofline_report(B, args.lr_size, args.scale, args.dtype, args.steps, pmc, source_stamp(), workload=wl, plan=plan)
        if roofline is not None:
            roofline["timing"] = ("HIP events (hipEventDisableSystemFence: timing only, no L2 write-back behind the measured kernel) on the kernel's own "
                                  "dispatches, launch stream, inside the timed region; "
                                  + ("every launch" if stride == 1 else f"1 launch in {stride} of the category (uniform sample; an event-carrying dispatch costs ~10 us of launch path)")
                                  + ".  avg_launch_us is the IN-STEP duration (two streams) and agrees with rocprofv3's kernel trace of this command run with "
                                    "--no-kernel-events; inside a rocprofv3 session the event-carrying launches themselves are recorded ~17 us longer "
                                    "(profiles/README.md, round 4), which lifts the average of the summary taken WITH events by ~3 us")
    # The other categories (side-stream weight-gradient GEMMs included) are timed AFTER the timed region, on every launch of two extra
    # steps, so that the step the value is quoted on carries events on one kernel only.  EVERY rank runs these two steps -- a step contains
    # the gradient all-reduce, and a collective that only rank 0 enters never returns -- whatever rank 0's report came out as.
    if not args.no_kernel_events and not args.all_kernel_events:
        if events_on:
            m2t_profile.enable(m2t_profile.ALL_MASK, sample_every=1)
        for s in range(2):
            ts.step(*batches[s % 2], captions)
        torch.cuda.synchronize()
        if events_on and roofline is not None:
            full = m2t_profile.roofline_report(B, args.lr_size, args.scale, args.dtype, 2, pmc, source_stamp(), workload=wl, plan=plan)
            if full is not None:
                rows = [{k: full[k] for k in ("category", "bound", "frac", "avg_launch_us", "launches_per_step", "est_ms_per_step", "hbm_GBs", "mfma_TFs")}] + full["others"]
                rows = [r for r in rows if r["category"] != roofline["category"]]
                rows.sort(key=lambda r: -r["est_ms_per_step"])
                roofline["others"] = rows[:5]
                roofline["others_timing"] = ("top five other categories by est_ms_per_step; timed on two extra steps behind the timed region with events on "
                                             "EVERY dispatch of every category (the all-events mode costs ~8 % of the step and lengthens what it times by a few us)")
    if events_on:
        m2t_profile.enable(0)
    out = {}
    if rank == 0:
        what = WORKLOAD_NOTE[args.config]
        if args.preset_overridden:
            what = ("L1 + MedCLIP(Swin-T) regulariser" if args.semantic_loss else "L1 loss only") + " (preset overridden on the command line)"
        # every switch that changes the measured work or schedule is echoed, so an experiment cannot pass for a headline
        experiment = {k: v for k, v in {
            "option": args.option or None,
            "no_side_stream": args.no_side_stream or None, "null_stream": args.null_stream or None, "main_priority": args.main_priority,
            "no_overlap_comm": args.no_overlap_comm or None, "force_comm_path": args.force_comm_path or None,
            "all_kernel_events": args.all_kernel_events or None, "no_overlap_semantic": args.no_overlap_semantic or None, "debug_skip_side": args.debug_skip_side or None}.items() if v is not None}
        out = {
            "metric": f"train-step HR patches/sec at {args.lr_size}x{args.lr_size} LR x{args.scale}",
            "value": None if args.debug_skip_side else round(world * B * args.steps / dt, 3),
            "unit": "HR patches/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1000.0 * dt / args.steps, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic U[0,1) LR/HR patches resident in HBM, seed-33 reference init" + (
                "; MedCLIP image tower = random-init Swin-T, text features = hash stand-ins (weights not vendored)" if args.semantic_loss else ""),
            # audit fields of a multi-GPU record: ranks read back from the process group, the device of every rank, and the time the
            # compute stream spent WAITING for the gradient exchange (by events around its wait on the communication stream; MAX over ranks)
            "rccl_ranks": grp["rccl_ranks"],
            "rank_devices": grp["rank_devices"],
            "exposed_comm_ms_per_step": None if exposed_ms is None else round(exposed_ms, 4),
            "config": {"workload": f"x{args.scale} SR train step (fwd + L1 + bwd + Adam), {args.lr_size}x{args.lr_size} LR "
                                   f"patches, batch {B}/GPU, " + what,
                       "global_batch": world * B, "per_gpu_batch": B, "parallelism": f"dp{world}",
                       "backend": ("rccl (torch.distributed 'nccl')" if backend == "nccl" else backend) if backend else "none (single process)",
                       "world_size": grp["pg_ranks"],
                       "ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                       "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)"),
                       "grad_exchange": ("none" if ts.bucket is None else
                                         ("bucketed all-reduce overlapped with backward" if ts.overlap_comm else "one all-reduce after backward")),
                       "stream_priority": main_priority,
                       "final_loss": round(loss, 6),
                       "host_enqueue_ms_per_step": round(1000.0 * host_enqueue_s / args.steps, 3)},
            "roofline": roofline,
        }
        if experiment:
            out["config"]["experiment_flags"] = experiment
        if args.debug_skip_side:
            out["invalid"] = True          # parameter-gradient kernels skipped: not a train step
    # release this workload's HBM (plans, workspaces, moments) before the next one
    del ts, model, plan, batches, sem
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    if rank == 0 and cpu_base:
        out["cpu_baseline"] = cpu_baseline(args)
    return out


_MAIN_STREAMS: dict = {}
ALSO_RUNS = [("config3", dict(config=3, dtype="bf16")), ("config4", dict(config=4, dtype="bf16")),
             ("config2", dict(config=2, dtype="bf16")), ("config1_fp32", dict(config=1, dtype="fp32"))]


def finalize_line(out: dict) -> dict:
    """Key order of the printed line.  The driver's record keeps the parsed contract keys and the LAST 2 000 characters of the line:
    the long per-workload tables (`also`, with their `others` lists) go in front, and the line ends with `also_summary` -- every extra
    workload's value / ms_per_step / dominant category / frac in < 600 characters -- followed by `cpu_baseline`."""
    also = out.get("also")
    tail_keys = ("also_summary", "cpu_baseline")
    res = {k: v for k, v in out.items() if k not in tail_keys and k != "also"}
    if also is not None:
        res["also"] = also
        summ = {}
        for e in also:
            if "value" in e:
                dk = e.get("dominant_kernel") or {}
                summ[e["workload"]] = [e["value"], e["ms_per_step"], dk.get("category"), dk.get("frac")]
            else:
                summ[e["workload"]] = (e.get("error") or e.get("skipped") or "no result")[:60]
        res["also_summary"] = {"columns": ["value (HR patches/s)", "ms_per_step", "dominant category", "frac"], **summ}
    if "cpu_baseline" in out:
        res["cpu_baseline"] = out["cpu_baseline"]
    return res


class AlsoWatch:
    """Failure agreement for the multi-rank `also` workload.  While it runs, a thread on every rank polls the process group's store
    for a failure flag; `fail()` sets the flag.  Whoever sees it (or sets it): rank 0 prints the headline line with the error,
    everyone leaves through os._exit(0) -- a rank blocked inside a collective cannot be unblocked any other way."""
    KEY = "m2t_also_failed"

    def __init__(self, rank: int, world: int, headline: dict, interval: float = 0.5):
        import threading
        self.rank, self.world, self.headline, self.interval = rank, world, headline, interval
        self.store = None
        self._stop = threading.Event()
        self._thread = None
        self._once = threading.Lock()
        if world > 1 and torch.distributed.is_initialized():
            try:
                from torch.distributed.distributed_c10d import _get_default_store
                self.store = _get_default_store()
            except Exception:
                self.store = None

    def _leave(self, msg: str):
        if not self._once.acquire(blocking=False):
            return
        if self.rank == 0:
            line = dict(self.headline)
            line["also"] = [{"workload": "config3", "error": msg[:300]}]
            print(json.dumps(finalize_line(line)), flush=True)
        sys.stderr.write(f"bench.py rank {self.rank}: extra workload abandoned ({msg[:200]})\n")
        sys.stderr.flush()
        os._exit(0)

    def _poll(self):
        while not self._stop.wait(self.interval):
            try:
                if self.store.check([self.KEY]):
                    self._leave(self.store.get(self.KEY).decode(errors="replace"))
            except Exception as e:                      # the store went away with the rank that hosted it: nobody is left to wait for
                if not self._stop.is_set():
                    self._leave(f"rendezvous store unreachable ({type(e).__name__})")
                return

    def start(self):
        import threading
        if self.store is None:
            return
        self._thread = threading.Thread(target=self._poll, daemon=True)
        self._thread.start()

    def stop(self):
        self._stop.set()

    def fail(self, msg: str):
        self._stop.set()
        if self.store is None:
            return
        try:
            self.store.set(self.KEY, f"rank {self.rank}: {msg}")
        except Exception:
            return
        time.sleep(2 * self.interval)       # let the other ranks read the flag before the store's owner may go away
        self._leave(f"rank {self.rank}: {msg}")


def also_args(args, config: int, dtype: str):
    """The argument set of one `also` run: the preset of BASELINE configs[config] at `dtype`, few steps, default options."""
    import copy
    a = copy.copy(args)
    pre = PRESETS[config]
    a.config, a.dtype = config, dtype
    a.batch, a.lr_size, a.scale, a.semantic_loss = pre["batch"], pre["lr_size"], pre["scale"], pre["sem"]
    a.preset_overridden = False
    a.steps, a.warmup = max(5, args.also_steps), 2
    a.cpu_baseline_batch = 2 if a.lr_size <= 128 else 1
    return a


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}: launch one rank per GPU")
    if args.stub_step:
        return stub_main(args, world, rank)
    if world > 1 or args.force_comm_path:
        # before the first HIP call: a rank keeps the null stream, the launch stream, the plan's side stream, the communication stream and
        # RCCL's own streams alive -- more than the runtime's default of four hardware queues, and a launch / side stream pair that shares a
        # queue loses the overlap of the backward pass (DESIGN.md "Hardware queues are a resource").  A value the caller exported wins.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    dev_index = 0 if args.selftest_shared_device else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    backend = None
    if world > 1 or (args.force_comm_path and "RANK" in os.environ):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rank == 0:
            print("bench.py: " + rank_environment(os.environ, world)[1], file=sys.stderr, flush=True)
        if args.selftest_shared_device:
            torch.distributed.init_process_group("gloo", rank=rank, world_size=world)     # (RCCL refuses two ranks on one device)
        else:
            torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        backend = torch.distributed.get_backend()
        if torch.distributed.get_world_size() != args.gpus and world > 1:
            raise SystemExit(f"process group has {torch.distributed.get_world_size()} ranks, --gpus says {args.gpus}")
    grp = group_report(dev_index)
    if world > 1 and len(set(grp["rank_devices"])) != world and not args.selftest_shared_device:
        raise SystemExit(f"bench.py: ranks share a device: {grp['rank_devices']} (one rank per GPU is the contract)")

    t_start = time.perf_counter()
    out = run_workload(args, device, rank, world, backend, grp, cpu_base=False)
    # the default line also carries short runs of the other BASELINE workloads, so that the driver's record shows them
    plain_default = (world == 1 and args.config == 1 and not args.preset_overridden and args.dtype == "bf16" and not args.option
                     and not any((args.no_side_stream, args.null_stream, args.no_overlap_comm, args.force_comm_path, args.all_kernel_events,
                                  args.no_overlap_semantic, args.debug_skip_side, args.no_kernel_events)))
    plain_multi = (world > 1 and args.config == 1 and not args.preset_overridden and args.dtype == "bf16" and not args.option
                   and not any((args.no_side_stream, args.null_stream, args.no_overlap_comm, args.all_kernel_events, args.no_overlap_semantic,
                                args.debug_skip_side, args.no_kernel_events)))
    if plain_multi and not args.no_also:
        # N > 1: EVERY rank runs configs[3]'s per-GPU share behind the headline (the collectives need all of them); rank 0 reports it.
        # A rank that raises inside it (out of memory, a plan error) leaves the others blocked in a collective it never enters: every
        # rank therefore watches the rendezvous store while the extra workload runs (AlsoWatch), and when ANY rank has failed rank 0
        # prints the headline line -- complete, with the failure under `also` -- and every rank leaves at once with status 0.
        a = also_args(args, **dict(ALSO_RUNS)["config3"])
        watch = AlsoWatch(rank, world, out)
        try:
            watch.start()
            r = run_workload(a, device, rank, world, backend, grp, cpu_base=False)
            watch.stop()
            if rank == 0:
                roof = r.get("roofline") or {}
                out["also"] = [{"workload": "config3", "what": r["config"]["workload"], "dtype": r["dtype"], "n_gpus": world,
                                "global_batch": r["config"]["global_batch"], "per_gpu_batch": r["config"]["per_gpu_batch"], "value": r["value"],
                                "unit": r["unit"], "ms_per_step": r["ms_per_step"], "steps": r["steps"], "warmup": r["warmup"],
                                "exposed_comm_ms_per_step": r.get("exposed_comm_ms_per_step"),
                                "dominant_kernel": {k: roof.get(k) for k in ("category", "kernel", "bound", "frac", "avg_launch_us")} if roof else None}]
        except Exception as e:            # the headline line must not be lost to a failure of the extra workload
            watch.fail(f"{type(e).__name__}: {e}"[:300])          # (does not return when other ranks may be inside a collective)
            if rank == 0:
                out["also"] = [{"workload": "config3", "error": f"{type(e).__name__}: {e}"[:300]}]
    if (plain_default or args.also_list) and not args.no_also and rank == 0 and world == 1:
        also = []
        runs = ALSO_RUNS if not args.also_list else [(n, dict(ALSO_RUNS)[n]) for n in args.also_list.split(",")]
        for name, kw in runs:
            if time.perf_counter() - t_start > 75.0:          # keep the whole default run within ~2 minutes
                also.append({"workload": name, "skipped": "time box"})
                continue
            a = also_args(args, **kw)
            try:
                r = run_workload(a, device, rank, world, backend, grp, cpu_base=False)
            except Exception as e:                              # an `also` run never takes the headline down with it
                also.append({"workload": name, "error": f"{type(e).__name__}: {e}"[:300]})
                continue
            rf = r.get("roofline") or {}
            also.append({"workload": name, "what": r["config"]["workload"], "dtype": r["dtype"], "value": r["value"], "unit": r["unit"],
                         "ms_per_step": r["ms_per_step"], "steps": r["steps"], "warmup": r["warmup"], "per_gpu_batch": r["config"]["per_gpu_batch"],
                         "final_loss": r["config"]["final_loss"],
                         "dominant_kernel": {k: rf.get(k) for k in ("category", "kernel", "bound", "frac", "avg_launch_us", "launches_per_step", "achieved", "peak", "unit", "traffic")},
                         "others": rf.get("others")})
        out["also"] = also
    if rank == 0:
        if args.selftest_shared_device:
            out["invalid"] = True           # harness self-test: the ranks shared one device
            out["value"] = None
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(finalize_line(out)), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
