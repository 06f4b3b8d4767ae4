"""CPU tests (no GPU): the C-ABI library loads and exports every declared symbol, the host-side
mirror of the reference interface (state_dict, init, loaders, errors), and the data-parallel
exchange over gloo (world_size 2)."""
import ctypes as C
import os
import re
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import m2trans_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _args(scale=4, nb=8):
    return types.SimpleNamespace(n_feats=64, scale=scale, rgb_range=1.0, n_blocks=nb, colors=3)


def test_library_exports_every_declared_symbol():
    from m2trans_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "m2t.h")).read()
    declared = set(re.findall(r"\b(m2t_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/m2t.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.m2t_version() >= 100


def test_plan_layout_matches_module_and_oracle_inventory():
    from m2trans_amd import _lib
    from m2trans_amd.M2Trans_network import create_model
    lib = _lib.load()
    for scale in (2, 3, 4):
        m = create_model(_args(scale))
        shapes = O.param_shapes(64, scale, 8)
        sd = m.state_dict()
        assert list(sd.keys()) == list(shapes.keys())
        assert all(tuple(sd[k].shape) == shapes[k] for k in sd)
        h = C.c_void_p()
        _lib.check(lib.m2t_plan_create(C.byref(h), 2, 40, 56, scale, 8, _lib.F32))
        assert lib.m2t_plan_query(h, b"num_params") == m.flat_params.numel()
        assert lib.m2t_plan_query(h, b"padded_h") == 64 and lib.m2t_plan_query(h, b"padded_w") == 64
        for n, (o, k) in m.param_offsets().items():
            assert lib.m2t_plan_query(h, ("param:" + n).encode()) == o
            assert lib.m2t_plan_query(h, ("numel:" + n).encode()) == k
        assert lib.m2t_plan_query(h, b"ws:no_such_tensor") == -1
        lib.m2t_plan_destroy(h)


def test_plan_argument_errors_are_reported():
    from m2trans_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.m2t_plan_create(C.byref(h), 1, 32, 32, 5, 8, 0) < 0          # bad scale
    assert b"bad argument" in lib.m2t_last_error_string()
    assert lib.m2t_plan_create(C.byref(h), 1, 16, 64, 4, 8, 0) < 0          # reflect pad 16 -> 32 needs pad < size
    with pytest.raises(_lib.M2TError):
        _lib.check(lib.m2t_plan_create(C.byref(h), 0, 32, 32, 4, 8, 0), "plan")


def test_parameters_are_views_of_one_flat_buffer_and_survive_apply():
    from m2trans_amd.M2Trans_network import create_model
    m = create_model(_args(4, 2))
    flat = m.flat_params
    base = flat.data_ptr()
    for (n, p), (o, k, s) in zip(m._trainable(), m._slots):
        assert p.data_ptr() == base + 4 * o and p.numel() == k
    m = m.float()                                  # _apply re-flattens
    flat2 = m.flat_params
    for (n, p), (o, k, s) in zip(m._trainable(), m._slots):
        assert p.data_ptr() == flat2.data_ptr() + 4 * o
    g = m.attach_flat_grads()
    for (n, p), (o, k, s) in zip(m._trainable(), m._slots):
        assert p.grad.data_ptr() == g.data_ptr() + 4 * o
    assert not m.sub_mean.weight.requires_grad and not m.add_mean.bias.requires_grad


def test_seed_33_init_checksums_match_reference_fixture(golden_dir):
    """Same torch seed -> same initial weights as the reference (fixture written by
    oracle/pin_against_reference.py from the real reference under seed 33)."""
    from m2trans_amd.M2Trans_network import create_model
    g = np.load(os.path.join(golden_dir, "init_seed33_x4.npz"))
    torch.manual_seed(33)
    m = create_model(_args(4, 8))
    sd = m.state_dict()
    names = [str(s) for s in g["names"]]
    assert names == list(sd.keys())
    sums = np.array([float(sd[k].double().sum()) for k in names])
    asums = np.array([float(sd[k].double().abs().sum()) for k in names])
    assert np.allclose(sums, g["sums"], rtol=0, atol=1e-9)
    assert np.allclose(asums, g["abs_sums"], rtol=0, atol=1e-9)
    assert torch.equal(sd["body.3.attn2.rel_w"], torch.from_numpy(g["body3_attn2_rel_w"]))


def test_load_state_dict_reference_semantics():
    from m2trans_amd.M2Trans_network import create_model
    m4 = create_model(_args(4, 1))
    m2 = create_model(_args(2, 1))
    sd4 = {("module." + k): v.clone() for k, v in m4.state_dict().items()}   # DataParallel prefix (train.py:73,345)
    m4b = create_model(_args(4, 1))
    m4b.load_state_dict(sd4, strict=True)
    for k, v in m4.state_dict().items():
        assert torch.equal(v, m4b.state_dict()[k])
    m2.load_state_dict(m4.state_dict())            # tail mismatch tolerated (M2Trans_network.py:96-98)
    assert torch.equal(m2.state_dict()["head.weight"], m4.state_dict()["head.weight"])
    bad = dict(m4.state_dict())
    bad["head.weight"] = torch.zeros(1)
    with pytest.raises(RuntimeError):
        m4b.load_state_dict(bad)
    with pytest.raises(KeyError):
        m4b.load_state_dict({"nope": torch.zeros(1)}, strict=True)


def test_forward_refuses_cpu_tensors():
    from m2trans_amd._lib import M2TError
    from m2trans_amd.M2Trans_network import create_model
    with pytest.raises(M2TError):
        create_model(_args(4, 1))(torch.zeros(1, 3, 32, 32))


def test_cosine_lr_matches_oracle():
    from m2trans_amd.train_step import cosine_lr
    for e in (0, 1, 57, 200):
        assert abs(cosine_lr(e) - O.cosine_lr(e)) < 1e-15


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "m2trans_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("# oracle", ""), fn


def test_gradient_buckets_tile_the_flat_buffer():
    """The ranges m2t_backward completes one after the other (tail, block pairs from last to first, head) must be
    contiguous, disjoint and cover the whole flat gradient buffer, in that order, for every configuration."""
    from m2trans_amd.M2Trans_network import create_model, Plan
    from m2trans_amd import _lib
    for scale in (2, 3, 4):
        for nb in (1, 2, 3, 8):
            m = create_model(_args(scale, nb))
            plan = Plan.__new__(Plan)
            h = C.c_void_p()
            _lib.check(_lib.load().m2t_plan_create(C.byref(h), 2, 32, 32, scale, nb, _lib.F32), "m2t_plan_create")
            plan.handle = h
            try:
                buckets = plan.grad_buckets()
                n = plan.query("num_params")
                assert buckets[0][1] == n and buckets[-1][0] == 0
                for (lo, hi), (lo2, hi2) in zip(buckets[:-1], buckets[1:]):
                    assert lo < hi and hi2 == lo                      # walks downwards without gaps
                assert buckets[0][0] == plan.query("param:tail.0.weight")
                first_body0 = min(o for nme, (o, k) in m.param_offsets().items() if nme.startswith("body.0."))
                assert buckets[-1][1] == first_body0
                assert sum(hi - lo for lo, hi in buckets) == n == m.flat_params.numel()
            finally:
                plan.handle = None
                _lib.load().m2t_plan_destroy(h)


# ---- data parallel over gloo, world_size 2 ---------------------------------------------------
def _dp_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from m2trans_amd.dist import GradBucket, global_divisor, shard_size, broadcast_params
        from m2trans_amd.M2Trans_network import create_model
        torch.set_num_threads(2)
        scale, nb, Bg, H, W = 4, 1, 4, 32, 32
        B = shard_size(Bg, world)
        p = O.closed_form_params(64, scale, nb)
        model = create_model(_args(scale, nb))          # CPU: only used for the flat layout
        torch.nn.Module.load_state_dict(model, {k: v.clone() for k, v in p.items()}, strict=True)
        flat = model.flat_params
        if rank == 1:
            flat.add_(1.0)                               # diverge, then re-sync from rank 0
        broadcast_params(flat, 0)
        x = O.closed_form_image(Bg, 3, H, W)
        hr = O.closed_form_image(Bg, 3, H * scale, W * scale, phase=0.7)
        xs, hs = x[rank * B:(rank + 1) * B], hr[rank * B:(rank + 1) * B]
        pr = {k: v for k, v in p.items()}
        _, _, g = O.l1_loss_and_grads(xs, hs, pr, scale, nb, loss_divisor=global_divisor(hs.numel(), world))
        grads = model.attach_flat_grads()
        for n, (o, k) in model.param_offsets().items():
            grads[o:o + k].copy_(g[n].reshape(-1))
        g_local = grads.clone()
        GradBucket(grads).all_reduce()
        # the overlapped exchange issues the same SUM bucket by bucket (same ranges, same order on every rank)
        gb = g_local.clone()
        bucket = GradBucket(gb)
        n = gb.numel()
        cuts = [n, int(0.8 * n), int(0.35 * n), 0]
        for hi, lo in zip(cuts[:-1], cuts[1:]):
            bucket.all_reduce_range(lo, hi)
        assert torch.equal(gb, grads)
        _, _, gfull = O.l1_loss_and_grads(x, hr, p, scale, nb)
        worst = 0.0
        for n, (o, k) in model.param_offsets().items():
            want = gfull[n].reshape(-1)
            worst = max(worst, float((grads[o:o + k] - want).abs().max() / (want.abs().max() + 1e-30)))
        # bf16 wire format: looser, but still the same gradient
        grads2 = grads.clone()
        GradBucket(grads2, comm_dtype=torch.bfloat16).all_reduce()
        wire = float((grads2 / world - grads).abs().max() / grads.abs().max())
        # bf16 wire, bucket by bucket (the overlapped exchange): each range is staged through its slice of the wire
        # buffer; the result must equal the one-shot bf16 exchange exactly (the cast and the SUM are element-wise)
        gb2 = g_local.clone()
        b2 = GradBucket(gb2, comm_dtype=torch.bfloat16, expect_world=world)
        for hi, lo in zip(cuts[:-1], cuts[1:]):
            b2.all_reduce_range(lo, hi)
        g_one = g_local.clone()
        GradBucket(g_one, comm_dtype=torch.bfloat16).all_reduce()
        assert torch.equal(gb2, g_one)
        # a world size the process group does not have must raise (TrainStep would otherwise scale by 1/N silently)
        try:
            GradBucket(g_local.clone(), expect_world=world + 1)
            raise AssertionError("GradBucket accepted a wrong world size")
        except RuntimeError:
            pass
        ret[rank] = (worst, wire, float(flat.sum()))
    finally:
        dist.destroy_process_group()


def test_data_parallel_sum_of_shard_grads_equals_full_batch_grad_gloo():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29600 + os.getpid() % 200
    mp.spawn(_dp_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        worst, wire, psum = ret[r]
        assert worst < 2e-4, worst          # fp32 sharded-sum vs full-batch gradient
        assert wire < 2e-2, wire            # bf16 bucket
    assert ret[0][2] == ret[1][2]           # replicas hold identical weights after the broadcast


def test_shard_size_rejects_uneven_batches():
    from m2trans_amd.dist import shard_size
    assert shard_size(256, 8) == 32
    with pytest.raises(ValueError):
        shard_size(10, 4)


def test_world_size_without_process_group_raises():
    """TrainStep(world_size=N > 1) divides the loss by N: without an initialised process group of N ranks that would
    silently train on gradients scaled by 1/N, so the bucket refuses."""
    from m2trans_amd.dist import GradBucket
    assert not dist.is_initialized()
    with pytest.raises(RuntimeError):
        GradBucket(torch.zeros(16), expect_world=2)


class _FakeStep:                         # the flat-buffer part of TrainStep, on the CPU
    def __init__(self, m, step_count=7, lr=5e-5):
        self.exp_avg = torch.randn_like(m.flat_params)
        self.exp_avg_sq = torch.rand_like(m.flat_params)
        self.step_count, self.lr, self.betas, self.eps = step_count, lr, (0.9, 0.999), 1e-8

    def set_lr(self, lr):
        self.lr = lr


def test_checkpoint_round_trip_reference_format():
    """train.py:341-349 / :92-108 wire format: module.-prefixed model keys, torch Adam state indexed in
    model.parameters() order (frozen MeanShift tensors occupy indices 0..3 and carry no state)."""
    from m2trans_amd.M2Trans_network import create_model
    from m2trans_amd.checkpoint import export_checkpoint, import_checkpoint
    m = create_model(_args(4, 1))
    fs = _FakeStep(m)
    ck = export_checkpoint(m, fs, epoch=3)
    assert all(k.startswith("module.") for k in ck["model_state_dict"])
    # a stock torch Adam over ALL parameters (like train.py:81) accepts the optimizer state
    ref_like = create_model(_args(4, 1))
    opt = torch.optim.Adam(ref_like.parameters(), lr=1e-4)
    opt.load_state_dict(ck["optimizer_state_dict"])
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, 200.0, eta_min=1e-6)
    sched.load_state_dict(ck["scheduler_state_dict"])
    assert sched.last_epoch == 2                       # saved BEFORE epoch 3's scheduler.step() (train.py:341-358)
    i_head = [n for n, _ in ref_like.named_parameters()].index("head.weight")
    o, k, shp = m._slots[0]
    assert torch.equal(opt.state_dict()["state"][i_head]["exp_avg"], fs.exp_avg[o:o + k].view(shp))
    m2 = create_model(_args(4, 1))
    fs2 = _FakeStep(m2)
    assert import_checkpoint(ck, m2, fs2) == 4
    assert torch.equal(m2.flat_params, m.flat_params)
    assert torch.equal(fs2.exp_avg, fs.exp_avg) and torch.equal(fs2.exp_avg_sq, fs.exp_avg_sq)
    assert fs2.step_count == 7 and fs2.lr == 5e-5 and fs2.scheduler_last_epoch == 2


def test_checkpoint_without_scheduler_state_is_a_pretrain_load_or_an_error():
    """train.py:85-88 (--pretrain): model weights alone -> fresh Adam moments, fresh cosine schedule, start_epoch 1.  A file with Adam
    state but no scheduler state is not something train.py:341-349 writes: resuming its moments on a restarted schedule would train
    something else, so it is refused."""
    from m2trans_amd.M2Trans_network import create_model
    from m2trans_amd.checkpoint import export_checkpoint, import_checkpoint
    m = create_model(_args(4, 1))
    ck = export_checkpoint(m, _FakeStep(m), epoch=5)
    weights_only = {"epoch": 5, "model_state_dict": ck["model_state_dict"]}
    m2 = create_model(_args(4, 1))
    fs2 = _FakeStep(m2)
    fs2.scheduler_last_epoch = 9
    lr_before = fs2.lr
    assert import_checkpoint(weights_only, m2, fs2) == 1               # start_epoch stays 1 (train.py:62)
    assert torch.equal(m2.flat_params, m.flat_params)
    assert fs2.step_count == 0 and fs2.scheduler_last_epoch == 0 and fs2.lr == lr_before
    assert float(fs2.exp_avg.abs().max()) == 0.0 and float(fs2.exp_avg_sq.abs().max()) == 0.0
    # model only, no step driver: the epoch bookkeeping of test.py-style loads is unchanged
    assert import_checkpoint(weights_only, create_model(_args(4, 1))) == 6
    broken = {k: v for k, v in ck.items() if k != "scheduler_state_dict"}
    m3 = create_model(_args(4, 1))
    fs3 = _FakeStep(m3)
    before = fs3.exp_avg.clone()
    with pytest.raises(ValueError, match="scheduler_state_dict"):
        import_checkpoint(broken, m3, fs3)
    assert torch.equal(fs3.exp_avg, before) and fs3.step_count == 7     # nothing half-restored


def test_resume_checkpoint_saved_before_the_first_optimizer_step_resets_the_moments():
    """A resume checkpoint with scheduler state and param_groups but an EMPTY Adam state (train.py:341-349 called before any
    optimizer.step): optimizer.load_state_dict (train.py:101) leaves zero moments and step 0 -- the importing TrainStep must not keep
    whatever it had accumulated before the load (round-5 advisor finding)."""
    from m2trans_amd.M2Trans_network import create_model
    from m2trans_amd.checkpoint import export_checkpoint, import_checkpoint
    m = create_model(_args(4, 1))
    ck = export_checkpoint(m, _FakeStep(m, step_count=0, lr=3e-5), epoch=2)
    assert ck["optimizer_state_dict"]["state"] == {} and ck["optimizer_state_dict"]["param_groups"]
    m2 = create_model(_args(4, 1))
    fs2 = _FakeStep(m2, step_count=11)                      # stale moments of an earlier run
    assert float(fs2.exp_avg.abs().max()) > 0
    assert import_checkpoint(ck, m2, fs2) == 3
    assert fs2.step_count == 0 and fs2.lr == 3e-5 and fs2.scheduler_last_epoch == 1
    assert float(fs2.exp_avg.abs().max()) == 0.0 and float(fs2.exp_avg_sq.abs().max()) == 0.0


def test_data_parallel_replica_surface_without_a_gpu(monkeypatch):
    """nn.DataParallel over several devices (the unchanged train.py:73 on a multi-GPU node) calls _replicate_for_data_parallel on the
    module every forward.  Defined behaviour, checked here without a GPU: the replica is marked, owns no flat buffer / plans of the
    master (they are bound per device in forward), a one-time warning names the fast path (torch.distributed.run), M2T_DATA_PARALLEL=error
    turns it into an M2TError, and a replica -- like the master -- refuses a CPU tensor (no CPU fallback)."""
    import warnings
    from m2trans_amd import M2Trans_network as N
    from m2trans_amd._lib import M2TError
    m = N.create_model(_args(4, 1))
    monkeypatch.setattr(N, "_DP_WARNED", False)
    monkeypatch.delenv("M2T_DATA_PARALLEL", raising=False)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        r = m._replicate_for_data_parallel()
        r2 = m._replicate_for_data_parallel()
    assert len(w) == 1 and "torch.distributed.run" in str(w[0].message)          # once per process
    assert r._dp_master is m and r.flat_params is None and r._plans is None and r is not r2
    assert m._dp_master is None and m.flat_params is not None                     # the master is untouched
    assert list(r.named_parameters(recurse=False)) == []                          # torch's replica contract
    with pytest.raises(M2TError, match="no CPU fallback"):
        r(torch.zeros(1, 3, 32, 32))
    monkeypatch.setenv("M2T_DATA_PARALLEL", "error")
    with pytest.raises(M2TError, match="torch.distributed.run"):
        m._replicate_for_data_parallel()


def test_bench_line_ends_with_the_compact_summary_and_the_cpu_baseline():
    """The driver's record keeps the last 2 000 characters of the line: `also_summary` (< 600 characters: every extra workload's value,
    ms_per_step, dominant category, frac) and `cpu_baseline` are the LAST keys, the long tables come first."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    others = [{"category": f"c{i}", "bound": "hbm", "frac": 0.25, "avg_launch_us": 30.0, "launches_per_step": 8, "est_ms_per_step": 0.5,
               "hbm_GBs": 2000.0, "mfma_TFs": 100.0} for i in range(5)]
    also = [{"workload": n, "what": "x" * 120, "dtype": "bf16", "value": 1234.567, "unit": "HR patches/s", "ms_per_step": 8.123, "steps": 6, "warmup": 2,
             "per_gpu_batch": 32, "final_loss": 0.4, "dominant_kernel": {"category": "attn_bwd_c256", "kernel": "k", "bound": "hbm", "frac": 0.2712,
                                                                        "avg_launch_us": 70.0}, "others": others}
            for n in ("config3", "config4", "config2")] + [{"workload": "config1_fp32", "error": "RuntimeError: " + "y" * 200}]
    out = {"metric": "m", "value": 1.0, "cpu_baseline": {"value": 1.63, "unit": "HR patches/s", "cores": 16, "kind": "port", "sample": "s" * 120},
           "roofline": {"frac": 0.29}, "also": also}
    line = bench.finalize_line(out)
    keys = list(line.keys())
    assert keys[-2:] == ["also_summary", "cpu_baseline"] and keys.index("also") < keys.index("also_summary")
    summ = json.dumps(line["also_summary"])
    assert len(summ) < 600, len(summ)
    assert line["also_summary"]["config3"] == [1234.567, 8.123, "attn_bwd_c256", 0.2712]
    assert line["also_summary"]["config1_fp32"].startswith("RuntimeError")
    tail = json.dumps(line)[-2000:]
    for n in ("config3", "config4", "config2", "config1_fp32", "cpu_baseline"):
        assert f'"{n}"' in tail
    assert bench.finalize_line({"metric": "m", "cpu_baseline": {"v": 1}}) == {"metric": "m", "cpu_baseline": {"v": 1}}


@pytest.mark.parametrize("fail_rank", [1, 0])
def test_bench_extra_workload_failure_on_one_rank_does_not_hang_the_others(fail_rank):
    """Multi-rank `also` run (round-5 advisor finding): a rank that raises before a collective leaves the others blocked in it.  With
    AlsoWatch every rank polls the rendezvous store; rank 0 still prints the ONE headline line (the failure under `also`), all ranks
    exit 0.  Exercised on two gloo ranks with the stub step; the failing rank is rank 1, then rank 0."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--stub-step",
                        "--stub-also-fail-rank", str(fail_rank)], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3
    assert d["also"][0]["workload"] == "config3" and f"rank {fail_rank}" in d["also"][0]["error"] and "stub failure" in d["also"][0]["error"]
    assert list(d.keys())[-1] == "also_summary" and "stub failure" in d["also_summary"]["config3"]
    assert "extra workload abandoned" in r.stderr


def test_checkpoint_matches_the_reference_manifest(golden_dir):
    """export_checkpoint against the manifest of what the REAL reference saves (DataParallel model + torch.optim.Adam +
    CosineAnnealingLR run through two epochs, oracle/pin_against_reference.py section 10): same keys at every level,
    same shapes / dtypes, same scalars (last_epoch = epoch - 1, _step_count = epoch, lr = cosine(epoch - 1))."""
    import json
    from m2trans_amd.M2Trans_network import create_model
    from m2trans_amd.checkpoint import export_checkpoint
    from m2trans_amd.train_step import cosine_lr
    manifest = O.checkpoint_manifest
    want = json.load(open(os.path.join(golden_dir, "checkpoint_manifest.json")))
    if want["torch_version"].split("+")[0] != torch.__version__.split("+")[0]:
        pytest.skip("manifest was written by another torch release (optimizer dict keys differ between releases)")
    m = create_model(_args(4, 1))
    got = manifest(export_checkpoint(m, _FakeStep(m, step_count=2, lr=cosine_lr(1)), epoch=2))

    def same(a, b, path=""):
        if isinstance(a, dict):
            assert isinstance(b, dict) and list(a.keys()) == list(b.keys()), (path, list(a.keys()), list(b.keys()) if isinstance(b, dict) else b)
            for k in a:
                same(a[k], b[k], path + "/" + str(k))
        elif isinstance(a, list):
            assert isinstance(b, list) and len(a) == len(b), (path, a, b)
            for i, (x, y) in enumerate(zip(a, b)):
                same(x, y, path + f"[{i}]")
        elif isinstance(a, float):
            assert abs(a - b) <= 1e-12 * max(1.0, abs(a)), (path, a, b)
        else:
            assert a == b, (path, a, b)
    for key in ("top_level_keys", "epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict"):
        same(want[key], got[key], key)
    assert len(want["optimizer_checksums"]) == len(got["optimizer_state_dict"]["state"]) == 21


def test_eval_and_data_entry_points_validate_arguments_without_a_gpu():
    """m2t_eval_metrics / m2t_crop_patches / m2t_image_to_tensor reject bad arguments before any launch (so this runs on
    the CPU box): sizes, descriptors out of the image, crops larger than the image."""
    import ctypes as C
    import numpy as np
    from m2trans_amd import _lib
    L = _lib.load()
    assert L.m2t_eval_metrics_scratch_bytes(1, 8, 8, 4) == 0                      # nothing left after the border crop
    assert L.m2t_eval_metrics_scratch_bytes(2, 64, 64, 4) == 8 * 2 * (64 + 2 * 2)  # 64 MSE chunks + 2x2 SSIM tiles per image
    one = C.c_void_p(16)                                                          # never dereferenced: the checks come first
    assert L.m2t_eval_metrics(one, one, 1, 8, 8, 4, 1.0, None, one, one, None) == -2
    assert b"m2t_eval_metrics" in L.m2t_last_error_string()
    desc = np.array([[0, 0, 53, 212, 50, 0, 0, 37]], dtype=np.int64)              # lx + 12 > 53
    assert L.m2t_crop_patches(one, one, desc.ctypes.data_as(C.c_void_p), 1, 3, 48, 4, one, one, None) == -2
    assert b"descriptor out of range" in L.m2t_last_error_string()
    desc[0, 4] = 0; desc[0, 6] = 8                                                # unknown flag bit
    assert L.m2t_crop_patches(one, one, desc.ctypes.data_as(C.c_void_p), 1, 3, 48, 4, one, one, None) == -2
    assert L.m2t_crop_patches(one, one, desc.ctypes.data_as(C.c_void_p), 1, 3, 50, 4, one, one, None) == -2   # 50 % 4 != 0
    assert L.m2t_image_to_tensor(one, 20, 30, 3, 21, 30, one, None) == -2         # crop taller than the image
    assert b"m2t_image_to_tensor" in L.m2t_last_error_string()


def test_roofline_work_table_matches_the_profiler_categories():
    """every category the work model emits is a profiler category, the fused / unfused variants partition the same FLOPs,
    and the dominant-kernel arithmetic of DESIGN.md holds (attention backward at C = 256 with the fused data gradient:
    10.6 GFLOP and 67 MB per launch at batch 16)."""
    from m2trans_amd import profile as P
    wd = P.algorithmic_work(16, 128, 4, "bf16")                     # defaults: the C = 16 / 64 backward recompute q | k | v
    w = P.algorithmic_work(16, 128, 4, "bf16", c16_recompute=False, c64_recompute=False, c16_prep=False)
    assert "gemm_qkv_dgrad" in w and "gemm_qkv_dgrad" not in wd      # "attn_bwd" = 3: the C = 16 projection gradient left the GEMM too
    assert set(w) <= set(P.CATS) and set(wd) <= set(P.CATS), set(w) - set(P.CATS)
    M16, M64 = 16 * 128 * 128, 16 * 64 * 64
    for cat, M, C in (("attn_bwd_c16", M16, 16), ("attn_bwd_c64", M64, 64)):
        assert abs((wd[cat][0] - w[cat][0]) - 8 * 2.0 * M * C * 3 * C) < 1.0          # the projection once more per launch ...
        assert abs((w[cat][1] - wd[cat][1]) - 8 * M * 2 * C * 2) < 1.0                # ... for 2 C fewer bf16 values read per pixel
    assert wd["attn_bwd_c256"] == w["attn_bwd_c256"]
    fl, by, n = P.algorithmic_work(16, 128, 4, "bf16", fused_prep_bwd=False)["attn_bwd_c256"]
    assert n == 16 and abs(fl / n - 10.64e9) < 0.05e9 and abs(by / n - 67.1e6) < 0.2e6
    # round 4: branch 3's launches (8 of the 16) carry branch 4's branch_prep_bwd: + 4 C bf16 values per low-resolution pixel = 100.7 MB
    fl2, by2, n2 = w["attn_bwd_c256"]
    assert n2 == 16 and fl2 == fl and abs(by2 / n2 - (67.1e6 + 100.66e6) / 2) < 0.3e6
    w0 = P.algorithmic_work(16, 128, 4, "bf16", fused_attn_fwd=False, fused_qkv_dgrad=False, c16_recompute=False, c64_recompute=False)
    tot = lambda t, keys: sum(t[k][0] for k in keys if k in t)
    fwd_keys = ["attn_fwd_c16", "attn_fwd_c64", "attn_fwd_c256", "attn_fused_c16", "attn_fused_c64", "attn_fused_c256", "gemm_qkv"]
    bwd_keys = ["attn_bwd_c16", "attn_bwd_c64", "attn_bwd_c256", "gemm_qkv_dgrad"]
    assert abs(tot(w, fwd_keys) - tot(w0, fwd_keys)) < 1e-3 * tot(w0, fwd_keys)
    assert abs(tot(w, bwd_keys) - tot(w0, bwd_keys)) < 1e-3 * tot(w0, bwd_keys)
    # the fused conv backward does the FLOPs of the two kernels it replaces on three tensor passes instead of four
    ws = P.algorithmic_work(16, 128, 4, "bf16", fused_conv_bwd=False)
    assert "conv3x3_bwd" in wd and "conv3x3_dgrad" not in wd and "conv3x3_bwd" not in ws
    assert wd["conv3x3_bwd"][0] == ws["conv3x3_dgrad"][0] + ws["conv3x3_wgrad"][0]
    assert wd["conv3x3_bwd"][1] * 4 == (ws["conv3x3_dgrad"][1] + ws["conv3x3_wgrad"][1]) * 3 and wd["conv3x3_bwd"][2] == 8


def test_bench_plain_multi_gpu_launch_builds_the_torchrun_child():
    """`python bench.py --gpus N` without a launcher (the driver's SCALE command): the parent -- before any GPU call --
    starts the ranks as a child process under torch.distributed.run and relays its status; it never re-execs."""
    import subprocess
    import sys
    import bench
    cmd = bench.child_command(["--gpus", "4", "--steps", "7", "--warmup", "2"], 4, 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    # the self-launched line lets the LAUNCHER bind its rendezvous port (no probe-then-reuse race), on 127.0.0.1
    cmd = bench.child_command(["--gpus", "2"], 2)
    assert "--standalone" in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1" and "--master-port" not in cmd
    assert "free_port" not in open(os.path.join(ROOT, "bench.py")).read()
    # HSA_ENABLE_IPC_MODE_LEGACY: defaulted to 0 (dmabuf IPC, what this pool's driver needs), a caller's value wins, and it is logged
    env, note = bench.rank_environment({"PATH": "/usr/bin"}, 2)
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "bench.py default" in note
    env, note = bench.rank_environment({"HSA_ENABLE_IPC_MODE_LEGACY": "1"}, 2)
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "1" and "caller" in note and "=1" in note
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "os.exec" not in src and "execv" not in src
    # fewer devices than ranks: the parent must fail loudly (non-zero, no JSON line, no fallback) instead of running on fewer devices.
    # It refuses by itself only when the KFD topology POSITIVELY says so ...
    import tempfile
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES",
                                                            "CUDA_VISIBLE_DEVICES")}
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, "0"))
        open(os.path.join(td, "0", "properties"), "w").write("simd_count 1024\n")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                           capture_output=True, text=True, env=dict(env, M2T_KFD_TOPOLOGY=td), timeout=300)
        assert r.returncode == 2 and "{" not in r.stdout
        assert "only 1 device(s) are visible" in r.stderr and "launching" not in r.stderr
        # ... and when the topology is unreadable (UNKNOWN, not zero) the ranks are started and fail with the real error
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                           capture_output=True, text=True, env=dict(env, M2T_KFD_TOPOLOGY=os.path.join(td, "missing")), timeout=300)
        if not torch.cuda.is_available():
            assert r.returncode != 0 and "{" not in r.stdout
        assert "device count unknown" in r.stderr and "launching" in r.stderr
    # under a launcher whose WORLD_SIZE disagrees with --gpus the rank refuses as well
    env.update(WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr
    # the parent counts devices from sysfs, never through the HIP runtime
    launch_src = src[src.index("def launch_ranks"):src.index("def stub_main")]
    assert "torch.cuda" not in launch_src and "visible_gpu_count()" in launch_src


def test_bench_visible_gpu_count_reads_kfd_topology(tmp_path, monkeypatch):
    """GPUs = KFD topology nodes with SIMDs (CPU nodes have none), narrowed by the *_VISIBLE_DEVICES lists."""
    import bench
    for i, simd in enumerate([0, 0, 1024, 1024, 1024]):
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\nmem_banks_count 1\n")
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert bench.visible_gpu_count(str(tmp_path)) == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.visible_gpu_count(str(tmp_path)) == 2
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "1")
    assert bench.visible_gpu_count(str(tmp_path)) == 1
    # a -1 token ends the list for the runtime: it is not a device
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "-1")
    assert bench.visible_gpu_count(str(tmp_path)) == 0
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1,-1,2")
    assert bench.visible_gpu_count(str(tmp_path)) == 1
    # unreadable / empty topology = UNKNOWN (None), not zero: the launcher then lets the ranks report the real error
    assert bench.visible_gpu_count(str(tmp_path / "missing")) is None
    (tmp_path / "empty").mkdir()
    assert bench.visible_gpu_count(str(tmp_path / "empty")) is None


def test_bench_two_rank_control_flow_end_to_end_on_a_stub_step():
    """`python bench.py --gpus 2 --stub-step`: the parent relays its arguments to a torch.distributed.run child, both gloo ranks
    pass the WORLD_SIZE guard, warm up, time EXACTLY K steps between barriers, MAX-reduce the time, and only rank 0 prints
    the one JSON line.  Rank 1's stand-in step is 2x slower than rank 0's: the reported time must be rank 1's."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--stub-step",
                        "--batch", "5"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # rank 0 only
    d = json.loads(lines[0])
    assert d["stub"] and d["invalid"] and d["value"] is None
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2                     # arguments relayed to the ranks
    assert d["config"]["world_size"] == 2 and d["config"]["global_batch"] == 10 and d["config"]["backend"] == "gloo"
    # rank 1 sleeps 20 ms per step, rank 0 10 ms: the barrier-bracketed MAX is >= 20 ms per step
    assert d["ms_per_step"] >= 19.5 and d["ms_per_step"] < 60.0, d
    assert "launching" in r.stderr and "torch.distributed.run" in r.stderr
    # audit fields of a multi-GPU record: ranks READ BACK from the process group, the device of every rank (all-gathered, in
    # rank order), the exposed communication time (rank 0 waits ~10 ms per step for the slower rank 1 inside the exchange)
    assert d["pg_ranks"] == 2 and d["rccl_ranks"] is None            # gloo here; the GPU line reports rccl_ranks = world size
    assert d["rank_devices"] == [0, 1]
    assert 5.0 <= d["exposed_comm_ms_per_step"] < 40.0, d
    assert d["config"]["ipc_mode_legacy"] == "0" and "HSA_ENABLE_IPC_MODE_LEGACY=0 (bench.py default)" in r.stderr
    assert "--standalone" in r.stderr


def test_bf16_gelu_approximation_error_bound():
    """bf16 mode evaluates the tail's GELU / GELU' by a sigmoid-quintic approximation (csrc/m2t_common.h, restated in the
    oracle's bf16-emulation mode): against the exact erf form of the reference (models/M2Trans_network.py:44,47) the error stays
    below 4e-5 / 1e-4 -- 1 % / 2.5 % of a bf16 ulp at 1.0 -- over the whole line, including beyond the clamp at |t| = 8."""
    t = torch.cat([torch.linspace(-20, 20, 400001, dtype=torch.float64), torch.tensor([-1e4, -30.0, 30.0, 1e4], dtype=torch.float64)])
    act, der = O.gelu_fast_both(t)
    assert float((act - torch.nn.functional.gelu(t)).abs().max()) < 4e-5
    assert float((der - O.gelu_derivative(t)).abs().max()) < 1e-4
    a32, d32 = O.gelu_fast_both(t.float())
    assert float((a32.double() - act).abs().max() / 1e4) < 1e-6 and float((d32.double() - der).abs().max()) < 1e-5


def test_every_plan_option_is_documented_in_the_header_and_readable():
    """include/m2t.h is the contract: every key m2t_set_option accepts must be described there (with its default in brackets), and
    m2t_plan_query("opt:<key>") must know it, so that a bench line can echo what was in force.  Parsed from the source: no GPU needed."""
    import os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    api = open(os.path.join(root, "m2trans_amd", "csrc", "m2t_api.hip")).read()
    hdr = open(os.path.join(root, "include", "m2t.h")).read()
    body = api[api.index('extern "C" int m2t_set_option'):]
    body = body[:body.index("\n}\n")]
    keys = sorted(set(re.findall(r'k == "([a-z_0-9]+)"', body)))
    assert len(keys) >= 13 and "fork_on_kernel" in keys and "fused_prep_fwd" in keys and "fused_prep_bwd" in keys
    query = api[api.index("long long m2t_plan_query"):]
    for k in keys:
        assert re.search(r'\*\s+"%s"\s+\[-?\d+\]' % k, hdr), f"option {k} is not documented (with its default) in include/m2t.h"
        if k != "debug_skip_side":
            assert f'o == "{k}"' in query, f'm2t_plan_query("opt:{k}") is missing'
    # INTEGRATION.md's list of options is the header's: same keys, same defaults (round-5 verdict: it still said "ten")
    integ = open(os.path.join(root, "INTEGRATION.md")).read()
    para = integ[integ.index("plan options of `include/m2t.h`") - 10:]
    para = para[:para.index("## Build")]
    listed = dict(re.findall(r"`([a-z_0-9]+)` \[(-?\d+)\]", para))
    documented = dict(re.findall(r'^ \*   "([a-z0-9_]+)"\s+\[(-?\d+)\]', hdr, re.M))
    assert listed == documented and set(keys) == set(documented), (set(listed) ^ set(documented), set(keys) ^ set(documented))
    assert f"The {len(documented)} plan options" in integ



def test_no_kernel_outside_the_fp32_whitelist_uses_scratch(monkeypatch, tmp_path):
    """Round 5 policy: the build FAILS when a non-fp32 kernel reports ScratchSize > 0 (a spilled value is reloaded by a VMEM operation
    that retires in order with the kernel's prefetches and behind its stores in flight).  The sidecar files written at compile time make
    an incremental build report what a clean one does."""
    from m2trans_amd import build as B
    B.build(force=False, verbose=False)                      # raises ScratchError on a violation
    assert B.scratch_violations() == []
    for src, rows in B.spill_report().items():
        for name, nbytes in rows:
            assert B._spill_allowed(name), (src, name, nbytes)
    for s in B.SOURCES:                                      # every object has its report (a missing sidecar would hide a spill)
        assert os.path.exists(os.path.join(B.HERE, "build", s.replace(".hip", ".spills.json"))), s
    # the remark parser and the policy, on a fabricated report
    remarks = ("x.hip:1:1: remark: Function Name: _ZN12_GLOBAL__N_122tail_bwd_stream_kernelILi3ELb1EEEvNS_6BSArgsE [-Rpass-analysis=kernel-resource-usage]\n"
               "x.hip:1:1: remark:     ScratchSize [bytes/lane]: 288 [-Rpass-analysis=kernel-resource-usage]\n"
               "x.hip:1:1: remark: Function Name: _ZN12_GLOBAL__N_122window_attn_bwd_kernelIfLi256ELi2EEEvPKT_ [-Rpass-analysis=kernel-resource-usage]\n"
               "x.hip:1:1: remark:     ScratchSize [bytes/lane]: 64 [-Rpass-analysis=kernel-resource-usage]\n"
               "x.hip:1:1: remark: Function Name: _ZN12_GLOBAL__N_111adam_kernelEPf [-Rpass-analysis=kernel-resource-usage]\n"
               "x.hip:1:1: remark:     ScratchSize [bytes/lane]: 0 [-Rpass-analysis=kernel-resource-usage]\n")
    rows = B._spill_report("x.hip", remarks)
    assert [n for _, n in rows] == [288, 64]
    assert not B._spill_allowed(rows[0][0]) and B._spill_allowed(rows[1][0])
    monkeypatch.setattr(B, "spill_report", lambda: {"x.hip": rows})
    assert [v[2] for v in B.scratch_violations()] == [288]
    monkeypatch.setattr(B, "SOURCES", [])
    (tmp_path / "none.so").write_bytes(b"")                  # present and newer than its (zero) objects: nothing to compile or link
    monkeypatch.setattr(B, "LIB", str(tmp_path / "none.so"))
    with pytest.raises(B.ScratchError, match="288 B/lane"):
        B.build(force=False, verbose=False)
