"""-m gpu: whole-model parity of the HIP path (through the C ABI) against the CPU oracle and
the committed golden vectors of the real reference.  Forward tests also compare every saved
intermediate (named workspace tensors) so that a failure points at one kernel."""
import os

import numpy as np
import math

import pytest
import torch

from oracle import m2trans_oracle as O
from tests.gpu_util import build_model, make_args, rel, rms_rel, ws_nchw

pytestmark = pytest.mark.gpu

BR_C = (16, 64, 256, 256)
BR_L = (0, 1, 2, 2)


def _forward_table(model, plan, cap, B, H, W, scale, nb):
    rows = []
    rows.append(("X0", rel(ws_nchw(plan, "X0", B, H, W, 64), cap["X0"])))
    for b in range(nb):
        for i in range(4):
            h, w = H >> BR_L[i], W >> BR_L[i]
            rows.append((f"b{b}.d{i+1}", rel(ws_nchw(plan, f"b{b}.d{i+1}", B, h, w, BR_C[i]), cap[f"b{b}.d{i+1}"])))
            rows.append((f"b{b}.qkv{i+1}", rel(ws_nchw(plan, f"b{b}.qkv{i+1}", B, h, w, 3 * BR_C[i]), cap[f"b{b}.qkv{i+1}"])))
        rows.append((f"b{b}.xc", rel(ws_nchw(plan, f"b{b}.xc", B, H, W, 64), cap[f"b{b}.xc"])))
        rows.append((f"X{b+1}", rel(ws_nchw(plan, f"X{b+1}", B, H, W, 64), cap[f"X{b+1}"])))
    r0 = 2 if scale == 4 else scale
    for nm in ("t1act", "t1der"):
        rows.append((nm, rel(ws_nchw(plan, nm, B, H * r0, W * r0, 64), cap[nm])))
    if scale == 4:
        for nm in ("t2act", "t2der"):
            rows.append((nm, rel(ws_nchw(plan, nm, B, H * 4, W * 4, 64), cap[nm])))
    pre = plan.ws_tensor("srpre", dtype=torch.float32).view(B, 3, H * scale, W * scale).cpu()
    rows.append(("srpre", rel(pre, cap["srpre_padded"])))
    return rows


def _oracle_forward_padded(x, p, scale, nb):
    """Oracle forward with capture; also the un-cropped pre-clamp output (padded size)."""
    cap = {}
    xp = O.pad_to_multiple(x)
    with torch.no_grad():
        pre = O.forward(xp, p, scale, nb, return_preclamp=True, cap=cap)
        sr = O.forward(x, p, scale, nb)
    cap["srpre_padded"] = pre
    return sr, cap


FWD_CASES = [
    # scale, n_blocks, B, H0, W0
    (4, 1, 2, 32, 32),
    (2, 2, 1, 32, 64),
    (3, 1, 1, 32, 32),
    (4, 2, 1, 40, 56),     # reflect pad to 64x64
]


@pytest.mark.parametrize("scale,nb,B,H0,W0", FWD_CASES)
def test_forward_fp32_all_intermediates(scale, nb, B, H0, W0):
    model, p = build_model(scale, nb, "fp32")
    x = O.closed_form_image(B, 3, H0, W0)
    sr_want, cap = _oracle_forward_padded(x, p, scale, nb)
    with torch.no_grad():
        sr = model(x.cuda())
    torch.cuda.synchronize()
    plan = model._plan_for(x.cuda())
    H, W = plan.query("padded_h"), plan.query("padded_w")
    rows = _forward_table(model, plan, cap, B, H, W, scale, nb)
    rows.append(("sr", rel(sr.cpu(), sr_want)))
    bad = [(n, e) for n, e in rows if not (e < 2e-4)]
    assert not bad, "\n".join(f"{n:12s} {e:.3e}" for n, e in rows)
    assert rel(sr.cpu(), sr_want) < 1e-4           # <= 1e-4 end-to-end forward


@pytest.mark.parametrize("name", ["x4_nf64_nb1_32", "x2_nf64_nb2_32x64", "x3_nf64_nb1_32", "x4_nf64_nb2_pad40x56"])
def test_forward_backward_vs_reference_golden(golden_dir, name):
    """HIP forward + backward against the REAL reference's outputs (fixtures)."""
    g = np.load(os.path.join(golden_dir, f"fwd_bwd_{name}.npz"))
    nf, scale, nb, B, H, W = (int(v) for v in g["meta"])
    model, p = build_model(scale, nb, "fp32")
    x = O.closed_form_image(B, 3, H, W).cuda()
    hr = O.closed_form_image(B, 3, H * scale, W * scale, phase=0.7).cuda()
    sr = model(x)
    loss = (sr - hr).abs().mean()
    loss.backward()
    assert rel(sr, torch.from_numpy(g["sr"])) < 1e-4
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5
    grads = {n: q.grad for n, q in model.named_parameters() if q.requires_grad}
    norms = dict(zip([str(s) for s in g["grad_names"]], g["grad_norms"]))
    rows = [(k, abs(float(v.double().norm()) - norms[k]) / (norms[k] + 1e-30)) for k, v in grads.items()]
    rows.append(("head.weight[full]", rel(grads["head.weight"], torch.from_numpy(g["grad_head_weight"]))))
    rows.append(("attn2.rel_h[full]", rel(grads["body.0.attn2.rel_h"], torch.from_numpy(g["grad_rel_h"]))))
    rows.append(("attn3.qkv[:8]", rel(grads["body.0.attn3.qkv_conv.weight"][:8], torch.from_numpy(g["grad_qkv3"]))))
    rows.append(("ff.bias[full]", rel(grads["body.0.feed_forward.0.bias"], torch.from_numpy(g["grad_ff_bias"]))))
    bad = [(n, e) for n, e in rows if not (e < 1e-3)]
    assert not bad, "\n".join(f"{n:40s} {e:.3e}" for n, e in rows)


@pytest.mark.parametrize("scale,nb,B,H0,W0", [(4, 2, 2, 32, 32), (3, 1, 1, 40, 56), (2, 1, 2, 32, 32)])
def test_backward_fp32_every_parameter(scale, nb, B, H0, W0):
    """Every parameter gradient, element-wise, against CPU autograd through the oracle
    (gradients <= 1e-4 relative, of the tensor's largest element)."""
    model, p = build_model(scale, nb, "fp32")
    x = O.closed_form_image(B, 3, H0, W0)
    hr = O.closed_form_image(B, 3, H0 * scale, W0 * scale, phase=0.7)
    loss_o, sr_o, g_o = O.l1_loss_and_grads(x, hr, p, scale, nb)
    sr = model(x.cuda())
    loss = torch.nn.L1Loss()(sr, hr.cuda())
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_o)) < 1e-5
    rows = [(n, rel(q.grad, g_o[n])) for n, q in model.named_parameters() if q.requires_grad]
    bad = [(n, e) for n, e in rows if not (e < 1e-4)]
    assert not bad, "\n".join(f"{n:40s} {e:.3e}" for n, e in rows)


def test_train_two_steps_vs_reference_golden(golden_dir):
    """Fused step driver (forward, L1, backward, Adam) x2 against torch.optim.Adam on the real
    reference (fixture) -- train.py:173-214."""
    from m2trans_amd.train_step import TrainStep, cosine_lr
    g = np.load(os.path.join(golden_dir, "train_2steps_x4_nf64_nb1_32.npz"))
    scale, nb, B, H, W = 4, 1, 2, 32, 32
    model, p = build_model(scale, nb, "fp32")
    ts = TrainStep(model, lr=cosine_lr(0), lambda_l1=1.0, world_size=1)
    for step in range(1, 3):
        x = O.closed_form_image(B, 3, H, W, phase=0.1 * step).cuda()
        hr = O.closed_form_image(B, 3, H * scale, W * scale, phase=0.7 + 0.1 * step).cuda()
        loss = ts.step(x, hr)
        assert abs(float(loss) - float(g["losses"][step - 1])) < 1e-5
    sd = model.state_dict()
    # Adam's first steps move each weight by ~lr whatever the gradient magnitude: compare the
    # UPDATE (w - w0), relative to lr
    for key, name in (("head_weight", "head.weight"), ("ff_bias", "body.0.feed_forward.0.bias"), ("tail6", "tail.6.weight")):
        want = torch.from_numpy(g[key])
        got = sd[name].cpu()
        assert float((got - want).abs().max()) < 0.05 * 2e-4, name     # 5 % of the 2-step update size


def test_the_reference_training_loop_as_written_runs_on_the_hip_module(golden_dir):
    """train.py:73,81-82,173-214 LITERALLY on the HIP module (round-5 verdict, Missing 2): ``nn.DataParallel(model).to(device)``, a stock
    ``torch.optim.Adam(model.parameters(), lr, weight_decay=0)`` + ``CosineAnnealingLR``, and the loop body ``optimizer.zero_grad(); sr =
    model(lr); loss = L1Loss()(sr, hr) * lambda_l1; loss.backward(); optimizer.step()`` for the two fixture steps of the REAL reference
    (``train_2steps_x4_nf64_nb1_32.npz``); then test.py:66-70: a fresh ``nn.DataParallel(M2Trans(args)).to(device)``,
    ``load_state_dict(ckpt['model_state_dict'], strict=True)`` THROUGH the wrapper with ``module.``-prefixed keys, ``model.eval()`` under
    ``torch.set_grad_enabled(False)``, forward equal to the trained model's."""
    import torch.nn as nn
    from torch.optim.lr_scheduler import CosineAnnealingLR
    from m2trans_amd.M2Trans_network import M2Trans, create_model
    g = np.load(os.path.join(golden_dir, "train_2steps_x4_nf64_nb1_32.npz"))
    scale, nb, B, H, W = 4, 1, 2, 32, 32
    args = make_args(scale, nb, "fp32")
    args.lr, args.epochs, args.eta_min, args.lambda_l1 = 1e-4, 200, 1e-6, 1.0
    device = torch.device("cuda")
    model = create_model(args)
    nn.Module.load_state_dict(model, {k: v.clone() for k, v in O.closed_form_params(64, scale, nb).items()}, strict=True)
    # ---- train.py:73,76,81-82
    model = nn.DataParallel(model).to(device)
    loss_l1 = torch.nn.L1Loss()
    lambda_l1 = args.lambda_l1
    optimizer = torch.optim.Adam(model.parameters(), lr=args.lr, weight_decay=0)
    scheduler = CosineAnnealingLR(optimizer, float(args.epochs), eta_min=args.eta_min)
    model = model.train()
    losses = []
    for it in range(1, 3):                                    # train.py:173-214
        optimizer.zero_grad()
        lr = O.closed_form_image(B, 3, H, W, phase=0.1 * it)
        hr = O.closed_form_image(B, 3, H * scale, W * scale, phase=0.7 + 0.1 * it)
        lr, hr = lr.to(device), hr.to(device)
        sr = model(lr)
        l1loss = loss_l1(sr, hr) * lambda_l1
        clip_loss = 0
        loss = l1loss + clip_loss
        loss.backward()
        optimizer.step()
        losses.append(float(loss))
    scheduler.step()                                          # train.py:358
    assert abs(scheduler.get_last_lr()[0] - O.cosine_lr(1)) < 1e-12
    assert max(abs(a - float(b)) for a, b in zip(losses, g["losses"])) < 1e-5, (losses, g["losses"])
    sd = model.state_dict()
    assert all(k.startswith("module.") for k in sd) and len(sd) == 4 + 2 + 14 * nb + 5       # the reference's inventory (123 at 8 blocks), DataParallel prefix
    for key, name in (("head_weight", "module.head.weight"), ("ff_bias", "module.body.0.feed_forward.0.bias"), ("tail6", "module.tail.6.weight")):
        assert float((sd[name].cpu() - torch.from_numpy(g[key])).abs().max()) < 0.05 * 2e-4, name   # 5 % of the 2-step update size
    # the optimiser wrote THROUGH the parameter views into the flat buffer the kernels read
    inner = model.module
    assert all(p.data_ptr() == inner.flat_params[o:o + k].data_ptr() for (n, p), (o, k, _) in zip(inner._trainable(), inner._slots))
    # ---- train.py:341-349 -> test.py:64-70
    checkpoint = {"epoch": 1, "model_state_dict": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}}
    lr_eval = O.closed_form_image(1, 3, 40, 56, phase=0.3).to(device)
    with torch.no_grad():
        want = model(lr_eval).clone()
    model2 = M2Trans(args)
    model2 = nn.DataParallel(model2).to(device)
    model2.load_state_dict(checkpoint["model_state_dict"], strict=True)
    model2 = model2.to(device)
    model2.eval()
    prev = torch.is_grad_enabled()
    torch.set_grad_enabled(False)
    try:
        model2.eval()
        got = model2(lr_eval)
    finally:
        torch.set_grad_enabled(prev)
    assert got.shape == (1, 3, 160, 224) and torch.equal(got, want)
    # strict=True means strict: a missing / unexpected key is an error through the wrapper as well
    broken = dict(checkpoint["model_state_dict"])
    broken.pop("module.head.bias")
    with pytest.raises(RuntimeError, match="head.bias"):
        model2.load_state_dict(broken, strict=True)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_data_parallel_over_two_replicas_matches_the_single_module(dtype):
    """nn.DataParallel spanning MORE than one device id (the unchanged train.py:73 on a multi-GPU node): torch replicates the module every
    forward, scatters the batch, runs the replicas in threads and reduces the gradients onto device 0.  The pool's boxes have one GPU, so
    the two replicas share it (device_ids=[0, 0]: torch's replicate / scatter / parallel_apply / gather run unchanged, the broadcast is
    a same-device copy): each replica binds its own flat parameter buffer and plans, two host threads drive the C ABI concurrently.
    sr rows equal the single module's (bitwise batch invariance), the reduced gradients equal the full-batch gradients to fp32 summation
    order, and a second step re-uses the pooled replica states."""
    import warnings
    import torch.nn as nn
    scale, nb, B, H, W = 4, 2, 4, 32, 64
    x = O.closed_form_image(B, 3, H, W).cuda()
    hr = O.closed_form_image(B, 3, H * scale, W * scale, phase=0.7).cuda()
    single, _ = build_model(scale, nb, dtype)
    sr1 = single(x)
    torch.nn.L1Loss()(sr1, hr).backward()
    g1 = {n: p.grad.clone() for n, p in single.named_parameters() if p.requires_grad}
    multi, _ = build_model(scale, nb, dtype)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        dp = nn.DataParallel(multi, device_ids=[0, 0])
        for step in range(2):
            for p in dp.parameters():
                p.grad = None
            sr2 = dp(x)
            torch.nn.L1Loss()(sr2, hr).backward()
            torch.cuda.synchronize()
            assert torch.equal(sr2, sr1), step
            rows = [(n, rel(p.grad, g1[n])) for n, p in multi.named_parameters() if p.requires_grad]
            tol = 2e-5 if dtype == "fp32" else 2e-2
            bad = [(n, e) for n, e in rows if not (e < tol)]
            assert not bad, (step, bad[:8])
    pool = multi._dp_pool[0]
    assert len(pool) == 2 and not any(s.busy for s in pool)          # two states made once, both returned
    assert all(len(s.plans) == 1 for s in pool)
    # eval through the wrapper: replicas of detached parameters, no autograd node
    with torch.no_grad():
        assert torch.equal(dp(x), sr1)


def test_bf16_forward_close_to_fp32_oracle():
    """bf16 MFMA mode (fp32 accumulate / statistics / softmax / master weights) against the FP32 oracle, i.e. the size of
    the bf16 effect itself: output rel-rms <= 5e-2, loss <= 2 %, whole-gradient cosine >= 0.99 (x2 / x3 / x4).  The
    per-tensor gradient gate is in tests/test_gpu_baseline_configs.py::test_bf16_small_model_against_bf16_rounding_oracle,
    against the oracle that rounds to bf16 at the same storage points (<= 2e-2 per tensor instead of the 25 % a comparison
    with un-rounded arithmetic needs)."""
    for scale in (4, 2, 3):
        nb, B, H, W = 2, 2, 32, 32
        model, p = build_model(scale, nb, "bf16")
        x = O.closed_form_image(B, 3, H, W)
        hr = O.closed_form_image(B, 3, H * scale, W * scale, phase=0.7)
        loss_o, sr_o, g_o = O.l1_loss_and_grads(x, hr, p, scale, nb)
        sr = model(x.cuda())
        loss = torch.nn.L1Loss()(sr, hr.cuda())
        loss.backward()
        assert rms_rel(sr, sr_o) < 5e-2
        assert abs(float(loss.detach()) - float(loss_o)) < 2e-2 * abs(float(loss_o)) + 1e-4
        names = [n for n, q in model.named_parameters() if q.requires_grad]
        got = torch.cat([dict(model.named_parameters())[n].grad.reshape(-1).double().cpu() for n in names])
        want = torch.cat([g_o[n].reshape(-1).double() for n in names])
        cos = float(torch.dot(got, want) / (got.norm() * want.norm()))
        assert cos > 0.99, (scale, cos)


def test_bf16_fast_kernels_match_plain_kernels():
    """A/B inside bf16 mode, at a size where every specialised kernel is live (512 conv tiles, 1024 C=16
    windows): whole-window-resident / wave-per-window attention backward, the fused tail backward and the gated
    side-stream schedule against the plain kernels on one stream.  The forward is untouched (bit-identical); the
    attention-backward variants round P / dS at different points, so the gradients agree to bf16 noise (stated:
    rel-rms <= 2e-2 per tensor, or <= 1e-3 of the whole gradient for the tiny-norm ones)."""
    from m2trans_amd import _lib
    scale, nb, B, H, W = 4, 2, 4, 128, 128
    x = O.closed_form_image(B, 3, H, W).cuda()
    hr = O.closed_form_image(B, 3, H * scale, W * scale, phase=0.7).cuda()
    outs = []
    for fast in (1, 0):
        model, _ = build_model(scale, nb, "bf16")
        plan = model._plan_for(x)
        for key, val in ((b"attn_bwd", 3 if fast else 0), (b"gate_branch", -1), (b"fused_conv_bwd", fast), (b"side_stream", fast), (b"fused_tail", 3 if fast else 0)):
            _lib.check(_lib.load().m2t_set_option(plan.handle, key, val), "m2t_set_option")
        sr = model(x)
        torch.nn.L1Loss()(sr, hr).backward()
        outs.append((sr.detach().clone(), {n: q.grad.detach().double().cpu() for n, q in model.named_parameters() if q.requires_grad}))
    (sr_a, g_a), (sr_b, g_b) = outs
    assert torch.equal(sr_a, sr_b)
    total = float(torch.cat([v.reshape(-1) for v in g_b.values()]).norm())
    bad = []
    for n in g_b:
        err, nrm = float((g_a[n] - g_b[n]).norm()), float(g_b[n].norm())
        if not (err <= 2e-2 * nrm or err <= 1e-3 * total):
            bad.append((n, err / (nrm + 1e-30), err / total))
    assert not bad, bad


@pytest.mark.parametrize("shape", [(4, 2, 128, 128), (4, 3, 64, 96)])
def test_branch_prep_inside_the_fused_forward_attention_is_bit_identical(shape):
    """"fused_prep_fwd": norm apply + branch mixing + DWT^L of the C = 64 / 256 branches run inside the fused forward attention kernel
    (each window also computes the blocks of its 36 halo keys) instead of in branch_prep launches in front of it.  Same operations,
    same rounding points: the output, the stored branch inputs d2..d4 and every gradient are bit-identical.  The second shape has
    windows on every image border (3 x 3 and 1 x ... window grids at the coarse levels)."""
    from m2trans_amd import _lib
    scale, B, H, W = shape
    nb = 2
    x = O.closed_form_image(B, 3, H, W).cuda()
    hr = O.closed_form_image(B, 3, H * scale, W * scale, phase=0.7).cuda()
    outs = []
    for val in (1, 0):
        model, _ = build_model(scale, nb, "bf16")
        plan = model._plan_for(x)
        _lib.check(_lib.load().m2t_set_option(plan.handle, b"fused_prep_fwd", val), "m2t_set_option")
        assert plan.query("opt:fused_prep_fwd") == val
        sr = model(x)
        ds = [plan.ws_tensor(f"b{b}.d{i}").clone() for b in range(nb) for i in (2, 3, 4)]
        torch.nn.L1Loss()(sr, hr).backward()
        outs.append((sr.detach().clone(), ds, torch.cat([q.grad.detach().reshape(-1) for q in model.parameters() if q.requires_grad]).clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)
    assert torch.equal(outs[0][2], outs[1][2])


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("shape", [(4, 2, 128, 128), (4, 3, 64, 96), (4, 2, 40, 56), (4, 5, 96, 128)])
def test_two_windows_per_cu_forward_kernel_matches_the_one_window_kernel(shape, variant):
    """"fused_attn_fwd2" (k_attn_fwd2.hip): the C = 256 forward branch as a 4-wave / 80 KB kernel (two windows per CU; projection in four
    output-channel chunks, scores / softmax / P in registers, v re-read from L2 for P V, IWT^2 straight from the accumulators) against
    the 8-wave one-window-per-CU kernel.  Same products in the same k order: the stored branch input d3, q | k | v of branch 3 (whose
    inputs are common to both runs) and the scores' inputs are BIT-identical; the branch output differs by the softmax form and the
    key order of the P V sums only (fp32 order -> rare 1-ulp flips of the bf16 result), and the step agrees to bf16 noise.  Shapes with
    windows on every border, a reflect-padded odd size and a 3 x 4 window grid."""
    from m2trans_amd import _lib
    scale, B, H, W = shape
    Hp, Wp = (H + 31) // 32 * 32, (W + 31) // 32 * 32
    x = O.closed_form_image(B, 3, H, W).cuda()
    hr = O.closed_form_image(B, 3, H * scale, W * scale, phase=0.7).cuda()
    outs = []
    for val in (variant, 0):
        model, _ = build_model(scale, 1, "bf16")
        plan = model._plan_for(x)
        _lib.check(_lib.load().m2t_set_option(plan.handle, b"fused_attn_fwd2", val), "m2t_set_option")
        # (variant 2 pairs neighbouring windows of one image: an odd window count per image falls back to variant 1)
        odd = ((Hp // 32) * (Wp // 32)) % 2 == 1
        assert plan.query("opt:fused_attn_fwd2") == (1 if (val == 2 and odd) else val)
        sr = model(x)
        torch.cuda.synchronize()
        keep = {n: plan.ws_tensor(n).clone() for n in ("b0.d3", "b0.qkv3", "b0.d4", "b0.qkv4", "b0.xc")}
        torch.nn.L1Loss()(sr, hr).backward()
        g = torch.cat([q.grad.detach().reshape(-1) for q in model.parameters() if q.requires_grad]).clone()
        outs.append((sr.detach().clone(), keep, g))
    (sr1, k1, g1), (sr0, k0, g0) = outs
    assert torch.equal(k1["b0.d3"], k0["b0.d3"])
    assert torch.equal(k1["b0.qkv3"], k0["b0.qkv3"])
    npix = B * Hp * Wp
    xc1, xc0 = k1["b0.xc"].float().view(4, npix, 16), k0["b0.xc"].float().view(4, npix, 16)
    assert torch.equal(xc1[:2], xc0[:2])                                   # planes 0, 1: other kernels
    for plane in (2, 3):
        dlt = (xc1[plane] - xc0[plane]).abs()
        ref = xc0[plane].abs()
        assert float(dlt.max()) <= 2.0 ** -7 * float(ref.max()) + 1e-6, (plane, float(dlt.max()), float(ref.max()))     # <= ~1 bf16 ulp of the largest value
        assert float((dlt > 0).float().mean()) < 0.08, (plane, float((dlt > 0).float().mean()))                           # and rare
        assert float(dlt.norm() / xc0[plane].norm()) < 1.5e-3
    assert float((k1["b0.qkv4"].float() - k0["b0.qkv4"].float()).norm() / k0["b0.qkv4"].float().norm()) < 3e-3
    assert float((sr1 - sr0).abs().max()) < 2e-2 and float((sr1 - sr0).norm() / sr0.norm()) < 2e-3
    assert float((g1 - g0).norm() / g0.norm()) < 5e-3


@pytest.mark.parametrize("shape", [(4, 2, 128, 128), (4, 3, 64, 96)])
def test_branch_prep_bwd_inside_the_attention_backward_is_bit_identical(shape):
    """"fused_prep_bwd": the backward of branch 4's branch_prep (overlap-add of the ring rows, IWT^2, the two halvings and the
    read-modify-write of g_xc[chunk 2]) runs in phase 0 of branch 3's attention backward instead of in a launch of its own.  Same
    operations in the same order and the same rounding points: every gradient bit-identical, including images whose coarse window
    grids are 3 x 2 and 2 x 1 (corner pixels with three ring sources, border windows with none on one side)."""
    from m2trans_amd import _lib
    scale, B, H, W = shape
    nb = 2
    x = O.closed_form_image(B, 3, H, W).cuda()
    hr = O.closed_form_image(B, 3, H * scale, W * scale, phase=0.7).cuda()
    outs = []
    for val in (1, 0):
        model, _ = build_model(scale, nb, "bf16")
        plan = model._plan_for(x)
        _lib.check(_lib.load().m2t_set_option(plan.handle, b"fused_prep_bwd", val), "m2t_set_option")
        assert plan.query("opt:fused_prep_bwd") == val
        grads = []
        for _ in range(2):
            model.zero_grad(set_to_none=True)
            torch.nn.L1Loss()(model(x), hr).backward()
            grads.append(torch.cat([q.grad.detach().reshape(-1) for q in model.parameters() if q.requires_grad]).clone())
        assert torch.equal(grads[0], grads[1])
        outs.append(grads[0])
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("shape", [(4, 3, 128, 128), (4, 2, 40, 56), (3, 2, 64, 96)])
def test_instnorm_backward_reduction_inside_the_c16_prep_launch(shape):
    """"fused_norm_red": the first reduction stage of the InstanceNorm backward (s1 = sum g_n, s2 = sum g_n xhat per image and channel)
    rides in the C = 16 prep launch -- extra workgroups for planes 1 .. 3, per-tile sums of plane 0 from the tiles that produce it --
    instead of in a launch of its own.  With ONE block everything in front of the reduction is bit-identical in both modes, so the two
    evaluations of (s1, s2) differ by fp32 addition order only; the bitwise batch invariance of the sums (an image's sums do not depend
    on its neighbours in the batch) must hold in the new mode too; at depth the whole gradient agrees to bf16 rounding flips."""
    from m2trans_amd import _lib
    scale, B, H, W = shape
    x = O.closed_form_image(B, 3, H, W).cuda()
    hr = O.closed_form_image(B, 3, H * scale, W * scale, phase=0.7).cuda()

    def run(nb, val, xin, hrin):
        model, _ = build_model(scale, nb, "bf16")
        plan = model._plan_for(xin)
        _lib.check(_lib.load().m2t_set_option(plan.handle, b"fused_norm_red", val), "m2t_set_option")
        assert plan.query("opt:fused_norm_red") == val
        model.zero_grad(set_to_none=True)
        torch.nn.L1Loss()(model(xin), hrin).backward()
        torch.cuda.synchronize()
        g = torch.cat([q.grad.detach().reshape(-1) for q in model.parameters() if q.requires_grad]).clone()
        return g, plan.ws_tensor("norm_s", dtype=torch.float32).clone().view(xin.shape[0], 64, 2)

    g1, s1 = run(1, 1, x, hr)
    g0, s0 = run(1, 0, x, hr)
    assert float((s1 - s0).abs().max() / s0.abs().max()) < 2e-6, (s1 - s0).abs().max()
    assert float((g1 - g0).norm() / g0.norm()) < 2e-3
    # batch invariance, bitwise: reverse the batch
    _, sr = run(1, 1, x.flip(0).contiguous(), hr.flip(0).contiguous())
    assert torch.equal(sr.flip(0), s1)
    # at depth 2 the summation order reaches bf16 roundings downstream: agreement to rounding flips
    h1, _ = run(2, 1, x, hr)
    h0, _ = run(2, 0, x, hr)
    assert float((h1 - h0).norm() / h0.norm()) < 3e-3


@pytest.mark.parametrize("shape", [(2, 128, 128), (3, 40, 56), (2, 72, 200)])
def test_l1_seed_inside_the_fused_tail_backward_is_bit_identical(shape):
    """m2t_l1_loss_deferred + "fused_l1": the clamp + L1 seed (train.py:199; clamp_l1_vec4_kernel) taken by the fused tail backward on the
    g(sr) halo it stages, instead of by a kernel of its own that writes the seed tensor.  Same arithmetic per pixel -> every gradient
    bit-identical; the loss value is the same sum of |clamp(sr) - hr| in another fp32 order.  Reflect-padded sizes (the crop: pixels of the
    padded output outside the image carry no seed and no loss), border and interior tiles, and the immediate m2t_l1_loss as a third arm."""
    from m2trans_amd import _lib
    from m2trans_amd.train_step import TrainStep
    B, H, W = shape
    scale, nb = 4, 2
    x = O.closed_form_image(B, 3, H, W).cuda()
    hr = O.closed_form_image(B, 3, H * scale, W * scale, phase=0.7).cuda()
    res = []
    for val in (1, 0):
        model, _ = build_model(scale, nb, "bf16")
        plan = model._plan_for(x)
        _lib.check(_lib.load().m2t_set_option(plan.handle, b"fused_l1", val), "m2t_set_option")
        assert plan.query("opt:fused_l1") == val
        ts = TrainStep(model, lr=1e-4, world_size=1)
        loss = ts.forward_backward(x, hr)
        torch.cuda.synchronize()
        res.append((float(loss), ts.grads.clone()))
    # third arm: the immediate API (loss and seed before m2t_backward)
    model, _ = build_model(scale, nb, "bf16")
    plan = model._plan_for(x)
    lib = _lib.load()
    ws, st = _lib.ptr(plan.workspace), _lib.stream_ptr()
    loss3 = torch.zeros(1, device="cuda")
    grads3 = torch.zeros_like(model.flat_params)
    _lib.check(lib.m2t_forward(plan.handle, _lib.ptr(model.flat_params), _lib.ptr(x), None, 1.0, 1, ws, st), "fwd")
    _lib.check(lib.m2t_l1_loss(plan.handle, _lib.ptr(hr), 1.0, float(hr.numel()), 1.0, _lib.ptr(loss3), ws, st), "l1")
    _lib.check(lib.m2t_backward(plan.handle, _lib.ptr(model.flat_params), _lib.ptr(x), _lib.ptr(grads3), ws, st), "bwd")
    torch.cuda.synchronize()
    (l1, g1), (l0, g0) = res
    assert torch.equal(g1, g0) and torch.equal(g1, grads3)
    assert abs(l1 - l0) <= 2e-6 * abs(l0) and abs(l0 - float(loss3)) == 0.0, (l1, l0, float(loss3))
    lo, _, _ = O.l1_loss_and_grads(x.cpu(), hr.cpu(), O.closed_form_params(64, scale, nb), scale, nb)
    assert abs(l1 - float(lo)) < 5e-3 * abs(float(lo))          # (bf16 forward against the fp32 oracle: the loss itself is the same quantity)


@pytest.mark.parametrize("dtype,scale,opts", [("fp32", 4, {}), ("fp32", 3, {}), ("bf16", 4, {b"fused_tail": 0}), ("bf16", 4, {b"fused_l1": 0}),
                                              ("bf16", 2, {b"fused_tail": 0})])
def test_deferred_l1_seed_is_ordered_before_the_side_stream_on_every_tail_path(dtype, scale, opts):
    """m2t_l1_loss_deferred on the paths where the clamp + L1 kernel runs inside m2t_backward in front of the tail (fp32 at every scale,
    bf16 with the plain tail kernels, x2 / x3): the tail conv's weight gradient reads the seed on the SIDE stream, so the seed kernel must
    precede the first fork (round-5 advisor finding: it was enqueued behind it -- the side stream could read gpre half-written or the
    previous step's).  Several consecutive steps with DIFFERENT batches through TrainStep (deferred) against the immediate m2t_l1_loss arm
    on a second model: every gradient bit-identical at every step (a stale seed would carry the previous batch's signs)."""
    from m2trans_amd import _lib
    from m2trans_amd.train_step import TrainStep
    nb, B, H, W = 2, 2, 64, 96
    lib = _lib.load()

    def make():
        model, _ = build_model(scale, nb, dtype)
        plan = model._plan_for(torch.empty(B, 3, H, W, device="cuda"))
        for key, val in opts.items():
            _lib.check(lib.m2t_set_option(plan.handle, key, val), "m2t_set_option")
        return model, plan
    m_def, _ = make()
    m_imm, plan = make()
    ts = TrainStep(m_def, lr=1e-4, world_size=1)
    ws, st = _lib.ptr(plan.workspace), _lib.stream_ptr()
    loss_i = torch.zeros(1, device="cuda")
    grads_i = torch.zeros_like(m_imm.flat_params)
    for step in range(4):
        x = O.closed_form_image(B, 3, H, W, phase=0.37 * step).cuda()
        hr = O.closed_form_image(B, 3, H * scale, W * scale, phase=0.7 + 0.91 * step).cuda()
        loss_d = ts.forward_backward(x, hr)
        _lib.check(lib.m2t_forward(plan.handle, _lib.ptr(m_imm.flat_params), _lib.ptr(x), None, 1.0, 1, ws, st), "fwd")
        _lib.check(lib.m2t_l1_loss(plan.handle, _lib.ptr(hr), 1.0, float(hr.numel()), 1.0, _lib.ptr(loss_i), ws, st), "l1")
        _lib.check(lib.m2t_backward(plan.handle, _lib.ptr(m_imm.flat_params), _lib.ptr(x), _lib.ptr(grads_i), ws, st), "bwd")
        torch.cuda.synchronize()
        assert float(loss_d) == float(loss_i), (step, float(loss_d), float(loss_i))
        if not torch.equal(ts.grads, grads_i):
            offs = m_def.param_offsets()
            bad = [n for n, (o, k) in offs.items() if not torch.equal(ts.grads[o:o + k], grads_i[o:o + k])]
            raise AssertionError(f"step {step}: gradients differ in {bad[:6]} ({len(bad)} tensors)")


def test_fork_event_on_the_dispatch_gives_the_same_bits_as_a_recorded_event():
    """"fork_on_kernel": the event that releases a branch's side-stream work rides on the attention-backward dispatch as its stop
    event (default) or is recorded behind it by a marker packet (0).  Same dependency either way: every gradient bit-identical,
    three steps in a row (a missing dependency would show as a race between the side stream's reads and the next block's writes)."""
    from m2trans_amd import _lib
    scale, nb, B, H, W = 4, 2, 4, 128, 128
    x = O.closed_form_image(B, 3, H, W).cuda()
    hr = O.closed_form_image(B, 3, H * scale, W * scale, phase=0.7).cuda()
    outs = []
    for val in (1, 0):
        model, _ = build_model(scale, nb, "bf16")
        plan = model._plan_for(x)
        _lib.check(_lib.load().m2t_set_option(plan.handle, b"fork_on_kernel", val), "m2t_set_option")
        assert plan.query("opt:fork_on_kernel") == val
        grads = []
        for _ in range(3):
            model.zero_grad(set_to_none=True)
            torch.nn.L1Loss()(model(x), hr).backward()
            grads.append(torch.cat([q.grad.detach().reshape(-1) for q in model.parameters() if q.requires_grad]).clone())
        assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])
        outs.append(grads[0])
    assert torch.equal(outs[0], outs[1])


def test_profile_sampling_times_every_nth_dispatch():
    """m2t_profile_sample_every(n): the dispatch-timed categories put their HIP events on every n-th launch only (bench.py uses
    n = 3 on the dominant kernel: an event-carrying dispatch costs ~10 us of launch path).  Two blocks launch the C = 256 attention
    backward four times per step: n = 1 records 4, n = 3 records launches 0 and 3, and the sampled average is a plausible duration."""
    from m2trans_amd import profile as P
    scale, nb, B = 4, 2, 2
    x = O.closed_form_image(B, 3, 64, 64).cuda()
    hr = O.closed_form_image(B, 3, 256, 256, phase=0.7).cuda()
    model, _ = build_model(scale, nb, "bf16")
    cat = P.CATS.index("attn_bwd_c256")
    counts = {}
    try:
        for n in (1, 3):
            P.enable(1 << cat, sample_every=n)
            torch.nn.L1Loss()(model(x), hr).backward()
            torch.cuda.synchronize()
            ms, cnt = P.read_all()["attn_bwd_c256"]
            counts[n] = cnt
            assert cnt > 0 and 1e-3 < ms / cnt < 5.0, (n, ms, cnt)
    finally:
        P.enable(0, sample_every=1)
    assert counts == {1: 4, 3: 2}, counts


def test_overlapped_gradient_exchange_path_single_rank():
    """The bucketed exchange (communication stream waiting on the per-bucket events of m2t_backward, then the compute
    stream waiting on the communication stream before Adam) with ONE rank: the collectives are identities, so two
    steps must give bit-identical parameters to the plain path, and the bucket events must order the streams
    correctly (a missing wait would let Adam read unfinished gradients)."""
    from m2trans_amd.train_step import TrainStep
    scale, nb, B, H, W = 4, 2, 2, 64, 64
    x = [O.closed_form_image(B, 3, H, W, phase=0.1 * i).cuda() for i in range(2)]
    hr = [O.closed_form_image(B, 3, H * scale, W * scale, phase=0.7 + 0.1 * i).cuda() for i in range(2)]
    finals = []
    for force in (False, True):
        model, _ = build_model(scale, nb, "bf16")
        ts = TrainStep(model, lr=1e-3, world_size=1, force_comm_path=force)
        assert ts.overlap_comm == force
        for i in range(2):
            ts.step(x[i], hr[i])
        torch.cuda.synchronize()
        finals.append(model.flat_params.detach().clone())
    assert torch.equal(finals[0], finals[1])


def test_config1_x2_64_vs_reference_golden(golden_dir):
    """BASELINE.json configs[0]: x2 forward on one 64x64 LR patch, full 8-block model."""
    g = np.load(os.path.join(golden_dir, "config1_x2_64.npz"))
    model, p = build_model(2, 8, "fp32")
    with torch.no_grad():
        sr = model(O.closed_form_image(1, 3, 64, 64).cuda())
    assert sr.shape == (1, 3, 128, 128)
    assert rel(sr, torch.from_numpy(g["sr"])) < 2e-4


def test_psnr_delta_fp32_and_bf16_vs_oracle():
    """PSNR (Y, reference eval formula utils.py:121-146,179-184) of HIP output vs oracle output
    on identical weights/data: |dPSNR| <= 0.02 dB (BASELINE target)."""
    scale, nb, B, H, W = 4, 8, 1, 64, 64
    p = O.closed_form_params(64, scale, nb)
    x = O.closed_form_image(B, 3, H, W)
    hr = O.closed_form_image(B, 3, H * scale, W * scale, phase=0.7)
    with torch.no_grad():
        sr_o = O.forward(x, p, scale, nb)
    ps_o = O.psnr_y(sr_o, hr, scale)
    for dt, tol in (("fp32", 1e-3), ("bf16", 0.02)):
        model, _ = build_model(scale, nb, dt, params=p)
        with torch.no_grad():
            sr = model(x.cuda()).cpu()
        assert abs(O.psnr_y(sr, hr, scale) - ps_o) <= tol, (dt, O.psnr_y(sr, hr, scale), ps_o)


def test_full_size_properties_128x128_x4():
    """At BASELINE's full size (x4, 128x128 LR) the oracle is too slow for every test run, so
    check size-independent properties: finite output in [0,1], batch-permutation equivariance
    (samples are independent), and determinism (no atomics anywhere)."""
    scale, nb, B, H, W = 4, 8, 2, 128, 128
    model, p = build_model(scale, nb, "fp32")
    x = O.closed_form_image(B, 3, H, W).cuda()
    with torch.no_grad():
        a = model(x).clone()
        b = model(x.flip(0)).flip(0)
        c = model(x)
    assert torch.isfinite(a).all() and float(a.min()) >= 0.0 and float(a.max()) <= 1.0
    assert torch.equal(a, c)
    assert torch.equal(a, b)


def test_no_cpu_fallback():
    from m2trans_amd._lib import M2TError
    from m2trans_amd.M2Trans_network import create_model
    model = create_model(make_args(4, 1))
    with pytest.raises(M2TError):
        model(torch.zeros(1, 3, 32, 32))


def test_training_drift_psnr_vs_oracle():
    """N identical Adam steps (same weights, same synthetic data, fp32 master weights) in the oracle on
    the CPU and in the build (fp32 and bf16 compute); PSNR-Y of a held-out pair through the eval formula.
    Target |dPSNR| <= 0.02 dB; the loss curves must agree step by step."""
    import torch.nn.functional as F
    from m2trans_amd.train_step import TrainStep
    scale, nb, B, H, W, N = 4, 2, 2, 32, 32, 24

    def pair(phase):
        hr = O.closed_form_image(B, 3, H * scale, W * scale, phase=phase)
        return F.avg_pool2d(hr, scale).contiguous(), hr

    p = {k: v.clone() for k, v in O.closed_form_params(64, scale, nb).items()}
    names = O.trainable_names(p)
    m = {k: torch.zeros_like(p[k]) for k in names}
    v = {k: torch.zeros_like(p[k]) for k in names}
    losses_o = []
    for s in range(1, N + 1):
        lr_img, hr_img = pair(0.3 * s)
        loss, _, g = O.l1_loss_and_grads(lr_img, hr_img, p, scale, nb)
        losses_o.append(float(loss))
        for k in names:
            p[k], m[k], v[k] = O.adam_update(p[k], g[k], m[k], v[k], s, 1e-4)
    lr_v, hr_v = pair(9.1)
    with torch.no_grad():
        ps_o = O.psnr_y(O.forward(lr_v, p, scale, nb), hr_v, scale)
        ps_0 = O.psnr_y(O.forward(lr_v, O.closed_form_params(64, scale, nb), scale, nb), hr_v, scale)
    assert abs(ps_o - ps_0) > 0.05, "the N steps must move the held-out PSNR for the comparison to mean anything"
    for dt, loss_tol, tol in (("fp32", 2e-5, 2e-3), ("bf16", 5e-3, 0.02)):
        model, _ = build_model(scale, nb, dt)
        ts = TrainStep(model, lr=1e-4, lambda_l1=1.0, world_size=1)
        for s in range(1, N + 1):
            lr_img, hr_img = pair(0.3 * s)
            loss = float(ts.step(lr_img.cuda(), hr_img.cuda()))
            assert abs(loss - losses_o[s - 1]) <= loss_tol * max(1.0, abs(losses_o[s - 1])), (dt, s, loss, losses_o[s - 1])
        with torch.no_grad():
            ps = O.psnr_y(model(lr_v.cuda()).cpu(), hr_v, scale)
        print(f"training drift {dt}: PSNR {ps:.4f} dB vs oracle {ps_o:.4f} dB (untrained {ps_0:.4f})")
        assert abs(ps - ps_o) <= tol, (dt, ps, ps_o)


@pytest.mark.parametrize("side", [1, 0])
def test_whole_step_is_hip_graph_capturable_and_replays_the_eager_bits(side):
    """include/m2t.h promises that every launch goes to the caller's stream (the backward's side stream is forked / joined with
    events), so the whole step -- forward, L1 loss, two-stream backward, fused Adam -- can be captured in a HIP graph.  Capture one
    step after a warm-up step (the first backward uploads its reduction table), replay it twice from a reset state and compare
    parameters, moments and loss with two eager steps from the same state: identical bits."""
    from m2trans_amd import _lib
    from m2trans_amd.train_step import TrainStep
    scale, nb, B, H0, W0 = 4, 2, 2, 64, 64
    x = O.closed_form_image(B, 3, H0, W0).cuda()
    hr = O.closed_form_image(B, 3, H0 * scale, W0 * scale, phase=0.7).cuda()

    def fresh():
        model, _ = build_model(scale, nb, "bf16")
        ts = TrainStep(model, lr=1e-3, world_size=1)
        plan = model._plan_for(x)
        _lib.check(_lib.load().m2t_set_option(plan.handle, b"side_stream", side), "m2t_set_option")
        ts.step(x, hr)                                   # warm-up: plan, workspace, reduction table
        torch.cuda.synchronize()
        return model, ts

    model_e, ts_e = fresh()
    for _ in range(2):
        ts_e.step(x, hr)
    torch.cuda.synchronize()
    want = (model_e.flat_params.clone(), ts_e.exp_avg.clone(), ts_e.exp_avg_sq.clone(), float(ts_e.loss))

    model_g, ts_g = fresh()
    state = (model_g.flat_params.clone(), ts_g.exp_avg.clone(), ts_g.exp_avg_sq.clone(), ts_g.step_count)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ts_g.step(x, hr)                                 # side-stream warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    # reset to the post-warm-up state, capture ONE step (step counter 2 -> bias correction of step 2), replay = steps 2 and 3?  The
    # fused Adam takes the step number as an argument, so a captured step repeats ITS bias correction: compare like with like --
    # eager reference below re-runs the same step number twice from the same state.
    model_g.flat_params.copy_(state[0]); ts_g.exp_avg.copy_(state[1]); ts_g.exp_avg_sq.copy_(state[2]); ts_g.step_count = state[3]
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        ts_g.step(x, hr)
    torch.cuda.synchronize()
    captured_step = ts_g.step_count
    model_g.flat_params.copy_(state[0]); ts_g.exp_avg.copy_(state[1]); ts_g.exp_avg_sq.copy_(state[2])
    gr.replay()
    torch.cuda.synchronize()
    got1 = (model_g.flat_params.clone(), ts_g.exp_avg.clone(), ts_g.exp_avg_sq.clone(), float(ts_g.loss))
    # eager: the same single step (same step number) from the same state
    model_e.flat_params.copy_(state[0]); ts_e.exp_avg.copy_(state[1]); ts_e.exp_avg_sq.copy_(state[2]); ts_e.step_count = captured_step - 1
    ts_e.step(x, hr)
    torch.cuda.synchronize()
    assert ts_e.step_count == captured_step
    assert torch.equal(got1[0], model_e.flat_params) and torch.equal(got1[1], ts_e.exp_avg) and torch.equal(got1[2], ts_e.exp_avg_sq)
    assert got1[3] == float(ts_e.loss)
    # and a second replay from the same state gives the same bits again (no hidden host state in the captured work)
    model_g.flat_params.copy_(state[0]); ts_g.exp_avg.copy_(state[1]); ts_g.exp_avg_sq.copy_(state[2])
    gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(model_g.flat_params, got1[0]) and torch.equal(ts_g.exp_avg_sq, got1[2])
    del want


def test_training_is_bitwise_reproducible_with_the_two_stream_schedule():
    """No atomics anywhere and every cross-stream hand-over is an event: two runs of the same six bf16 steps (full
    depth, 128x128, side stream + gates + deferred reductions live) must end with bit-identical parameters and Adam
    state.  A missing dependency between the two streams shows up here as a run-to-run difference."""
    from m2trans_amd.train_step import TrainStep
    scale, nb, B, H, W = 4, 8, 4, 128, 128
    finals = []
    for run in range(2):
        model, _ = build_model(scale, nb, "bf16")
        ts = TrainStep(model, lr=1e-4, lambda_l1=1.0, world_size=1)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):            # a non-default caller stream, like bench.py
            for s in range(6):
                x = O.closed_form_image(B, 3, H, W, phase=0.2 * s).cuda()
                hr = O.closed_form_image(B, 3, H * scale, W * scale, phase=0.5 + 0.2 * s).cuda()
                loss = ts.step(x, hr)
            side.synchronize()
        finals.append((float(loss), model.flat_params.detach().clone(), ts.exp_avg.clone(), ts.exp_avg_sq.clone()))
    (la, pa, ma, va), (lb, pb, mb, vb) = finals
    assert la == lb and torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)


def test_checkpoint_resume_is_bit_identical():
    """F1 (train.py:92-108,341-349): 2 steps -> export_checkpoint -> (through torch.save / torch.load) -> a FRESH model and
    TrainStep -> import_checkpoint -> 2 more steps must equal 4 uninterrupted steps bit for bit (weights, both Adam
    moments, step count), in bf16 mode with the two-stream schedule live."""
    import io
    from m2trans_amd.checkpoint import export_checkpoint, import_checkpoint
    from m2trans_amd.train_step import TrainStep, cosine_lr
    scale, nb, B, H, W = 4, 2, 2, 64, 64
    data = [(O.closed_form_image(B, 3, H, W, phase=0.2 * s).cuda(),
             O.closed_form_image(B, 3, H * scale, W * scale, phase=0.5 + 0.2 * s).cuda()) for s in range(4)]
    model, _ = build_model(scale, nb, "bf16")
    ts = TrainStep(model, lr=cosine_lr(2), world_size=1)
    for s in range(4):
        ts.step(*data[s])
    torch.cuda.synchronize()
    want = (model.flat_params.clone(), ts.exp_avg.clone(), ts.exp_avg_sq.clone())
    model_a, _ = build_model(scale, nb, "bf16")
    ts_a = TrainStep(model_a, lr=cosine_lr(2), world_size=1)
    for s in range(2):
        ts_a.step(*data[s])
    buf = io.BytesIO()
    torch.save(export_checkpoint(model_a, ts_a, epoch=3), buf)
    buf.seek(0)
    ck = torch.load(buf, weights_only=False)
    assert ck["scheduler_state_dict"]["last_epoch"] == 2 and abs(ck["optimizer_state_dict"]["param_groups"][0]["lr"] - cosine_lr(2)) < 1e-18
    model_b, _ = build_model(scale, nb, "bf16", params=O.closed_form_params(64, scale, nb, gain=0.5))   # different weights
    ts_b = TrainStep(model_b, lr=1.0, world_size=1)
    assert import_checkpoint(ck, model_b, ts_b) == 4
    assert ts_b.step_count == 2 and ts_b.lr == cosine_lr(2) and ts_b.scheduler_last_epoch == 2
    for s in range(2, 4):
        ts_b.step(*data[s])
    torch.cuda.synchronize()
    assert torch.equal(model_b.flat_params, want[0]) and torch.equal(ts_b.exp_avg, want[1]) and torch.equal(ts_b.exp_avg_sq, want[2])


def test_two_rank_data_parallel_over_rccl_when_two_gpus_are_visible():
    """tools/dp2_check.py under torch.distributed.run with one rank per GPU over RCCL: overlapped == plain exchange bit
    for bit, replicas identical, DP == single-process full batch.  Skipped on the 1-GPU boxes (there the same tool
    runs both ranks on one GPU over gloo: `tools/run_tests_bench.sh`)."""
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", M2T_DP_BACKEND="nccl")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", os.path.join(root, "tools", "dp2_check.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "overlap == plain: True" in r.stdout


def test_autograd_path_accepts_non_contiguous_and_half_inputs():
    """The kernels read raw contiguous fp32 NCHW memory; model(x) under autograd must therefore convert once and hand
    the CONVERTED tensor to m2t_backward too (the head weight gradient re-reads the input): a channels_last view and a
    float64 copy of the same values must give the gradients of the plain tensor bit for bit."""
    scale, nb, B, H, W = 4, 1, 2, 32, 32
    x = O.closed_form_image(B, 3, H, W).cuda()
    hr = O.closed_form_image(B, 3, H * scale, W * scale, phase=0.7).cuda()
    outs = []
    for variant in ("plain", "channels_last", "float64", "sliced"):
        model, _ = build_model(scale, nb, "fp32")
        if variant == "plain":
            xi = x
        elif variant == "channels_last":
            xi = x.to(memory_format=torch.channels_last)
            assert not xi.is_contiguous()
        elif variant == "float64":
            xi = x.double()
        else:
            big = torch.zeros(B, 3, H, 2 * W, device="cuda")
            big[..., ::2] = x
            xi = big[..., ::2]
            assert not xi.is_contiguous()
        torch.nn.L1Loss()(model(xi), hr).backward()
        outs.append({n: q.grad.clone() for n, q in model.named_parameters() if q.requires_grad})
    for o in outs[1:]:
        for n in outs[0]:
            assert torch.equal(o[n], outs[0][n]), n


def test_fused_qkv_attention_forward_matches_the_two_kernel_path():
    """bf16: the fused qkv-projection + window-attention kernel (k_attn_fused.hip, default for the C = 64 / 256 branches)
    against the GEMM + attention kernels it replaces, at a size with border, edge and interior windows in every branch
    (2 x 3 windows at C = 256) and a reflect-padded input.  Same products and rounding points; the fused kernel sums the
    softmax and P V over the keys in another order (own pixels first), so: saved q|k|v identical up to rare 1-ulp flips,
    block outputs within bf16 noise (stated: rel-rms <= 2e-3 for the first block's tensors)."""
    from m2trans_amd import _lib
    from tests.gpu_util import ws_nchw
    scale, nb, B, H0, W0 = 4, 2, 2, 60, 90            # padded to 64 x 96
    x = O.closed_form_image(B, 3, H0, W0).cuda()
    outs = []
    for fused in (1, 0):
        model, _ = build_model(scale, nb, "bf16")
        plan = model._plan_for(x)
        _lib.check(_lib.load().m2t_set_option(plan.handle, b"fused_attn_fwd", fused), "m2t_set_option")          # (1: qkv stored, compared below)
        _lib.check(_lib.load().m2t_set_option(plan.handle, b"fused_c16_fwd", 1), "m2t_set_option")
        with torch.no_grad():
            sr = model(x)
        torch.cuda.synchronize()
        H, W = plan.query("padded_h"), plan.query("padded_w")
        t = {f"b0.qkv{i+1}": ws_nchw(plan, f"b0.qkv{i+1}", B, H >> l, W >> l, 3 * c) for i, (c, l) in enumerate(zip(BR_C, BR_L))}
        t["b0.xc"] = ws_nchw(plan, "b0.xc", B, H, W, 64)
        t["X1"] = ws_nchw(plan, "X1", B, H, W, 64)
        t["sr"] = sr.cpu()
        outs.append(t)
    a, b = outs
    assert torch.equal(a["b0.qkv1"], b["b0.qkv1"])                         # C = 16 branch: same kernels
    for k in ("b0.qkv2", "b0.qkv3", "b0.qkv4", "b0.xc", "X1"):
        assert rms_rel(a[k], b[k]) < 2e-3, (k, rms_rel(a[k], b[k]), rel(a[k], b[k]))
    assert rms_rel(a["sr"], b["sr"]) < 5e-2


def test_fused_c16_branch_forward_matches_the_three_kernel_path():
    """bf16: InstanceNorm apply + qkv projection + window attention + residual of the C = 16 branch in one
    wave-per-window kernel (k_attn_c16.hip, option fused_c16_fwd, default) against branch_prep + GEMM + attention.
    d1 (the normalised chunk) must be identical; q|k|v come from a 16-deep MFMA instead of a zero-padded 32-deep one
    (same 16 products, the hardware's summation may round differently: <= 1 bf16 ulp on rare elements), the block's
    tensors stay within bf16 noise.  Reflect-padded input, border / edge / interior windows."""
    from m2trans_amd import _lib
    from tests.gpu_util import ws_nchw
    scale, nb, B, H0, W0 = 4, 2, 2, 60, 90
    x = O.closed_form_image(B, 3, H0, W0).cuda()
    outs = []
    for fused in (1, 0):
        model, _ = build_model(scale, nb, "bf16")
        plan = model._plan_for(x)
        _lib.check(_lib.load().m2t_set_option(plan.handle, b"fused_c16_fwd", fused), "m2t_set_option")
        with torch.no_grad():
            sr = model(x)
        torch.cuda.synchronize()
        H, W = plan.query("padded_h"), plan.query("padded_w")
        t = {"b0.d1": ws_nchw(plan, "b0.d1", B, H, W, 16), "b0.qkv1": ws_nchw(plan, "b0.qkv1", B, H, W, 48),
             "b1.d1": ws_nchw(plan, "b1.d1", B, H, W, 16),
             "b0.xc": ws_nchw(plan, "b0.xc", B, H, W, 64), "X1": ws_nchw(plan, "X1", B, H, W, 64), "sr": sr.cpu()}
        outs.append(t)
    a, b = outs
    assert torch.equal(a["b0.d1"], b["b0.d1"])
    d = (a["b0.qkv1"].float() - b["b0.qkv1"].float()).abs()
    assert float((d > 0).float().mean()) < 2e-2 and rel(a["b0.qkv1"], b["b0.qkv1"]) < 8e-3, (float((d > 0).float().mean()), rel(a["b0.qkv1"], b["b0.qkv1"]))
    for k in ("b0.xc", "X1", "b1.d1"):
        assert rms_rel(a[k], b[k]) < 2e-3, (k, rms_rel(a[k], b[k]), rel(a[k], b[k]))
    assert rms_rel(a["sr"], b["sr"]) < 5e-2


@pytest.mark.parametrize("key,branch", [(b"fused_c16_fwd", 1), (b"fused_attn_fwd", 2)])
def test_backward_recomputes_qkv_with_identical_bits(key, branch):
    """bf16, C = 16 and C = 64 branches: with fused_c16_fwd = 2 / fused_attn_fwd = 2 (defaults) the forward does not store
    q | k | v of that branch and the backward kernel recomputes them from the branch input d with the forward kernel's own MFMAs
    (same fragments, k order and rounding points): the whole step must agree BIT FOR BIT with the stored-qkv path (option = 1).
    Reflect-padded input, border / edge / interior windows."""
    from m2trans_amd import _lib
    from tests.test_gpu_baseline_configs import fwd_bwd
    scale, nb, B, H0, W0 = 4, 2, 2, 60, 90
    x = O.closed_form_image(B, 3, H0, W0).cuda()
    hr = O.closed_form_image(B, 3, H0 * scale, W0 * scale, phase=0.7).cuda()
    outs = []
    for mode in (2, 1):
        model, _ = build_model(scale, nb, "bf16")
        plan = model._plan_for(x)
        _lib.check(_lib.load().m2t_set_option(plan.handle, key, mode), "m2t_set_option")
        assert plan.query("opt:" + key.decode()) == mode and plan.query(f"stores_qkv{branch}") == (0 if mode == 2 else 1)
        sr, loss, grads = fwd_bwd(model, x, hr, hr.numel())
        outs.append((sr.cpu(), grads.cpu()))
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])


def test_changing_a_storage_option_between_forward_and_backward_is_refused():
    """Which of q | k | v the forward stores depends on `attn_bwd` (and on fused_attn_fwd / fused_c16_fwd / fused_tail): a
    backward under another value would read workspace tensors the forward never wrote.  m2t_set_option therefore drops the
    activations, and m2t_l1_loss / m2t_backward return M2T_ERR_STATE until the forward has run again."""
    from m2trans_amd import _lib
    scale, nb, B, H0, W0 = 4, 1, 1, 32, 32
    x = O.closed_form_image(B, 3, H0, W0).cuda()
    hr = O.closed_form_image(B, 3, H0 * scale, W0 * scale, phase=0.7).cuda()
    lib = _lib.load()
    loss = torch.zeros(1, device="cuda")
    for key, val in ((b"attn_bwd", 1), (b"attn_bwd", 0), (b"fused_attn_fwd", 1), (b"fused_c16_fwd", 1), (b"fused_tail", 1)):
        model, _ = build_model(scale, nb, "bf16")
        plan = model._plan_for(x)
        ws, st = _lib.ptr(plan.workspace), _lib.stream_ptr()
        grads = torch.empty_like(model.flat_params)

        def fwd():
            _lib.check(lib.m2t_forward(plan.handle, _lib.ptr(model.flat_params), _lib.ptr(x), None, 1.0, 1, ws, st), "m2t_forward")

        def l1():
            return lib.m2t_l1_loss(plan.handle, _lib.ptr(hr), 1.0, float(hr.numel()), 1.0, _lib.ptr(loss), ws, st)

        def bwd():
            return lib.m2t_backward(plan.handle, _lib.ptr(model.flat_params), _lib.ptr(x), _lib.ptr(grads), ws, st)

        fwd()
        assert l1() == 0
        _lib.check(lib.m2t_set_option(plan.handle, key, val), "m2t_set_option")      # between the forward and the backward
        assert bwd() != 0 and "m2t_forward" in lib.m2t_last_error_string().decode()
        assert l1() != 0
        fwd()                                            # a fresh forward under the new option makes the plan usable again
        assert l1() == 0 and bwd() == 0
        torch.cuda.synchronize()
        assert bool(torch.isfinite(grads).all())


def test_fused_projection_data_gradient_matches_the_gemm_path():
    """bf16, C = 64 / 256 branches: the data gradient of the qkv projection taken inside the attention backward kernel
    (option fused_qkv_dgrad, default; the window multiplies its own dq | dK | dV contributions by Wqkv^T and the
    overlap-add over neighbouring windows happens on the C-wide product) against halo gather + GEMM.  Linear in dK | dV,
    so only the bf16 rounding points differ (per-window partial products are rounded before the overlap-add):
    every parameter gradient within 2e-2 of the GEMM path, the whole gradient within 3e-3.  Reflect-padded input with
    border / edge / interior windows in every branch.  (The ring rows are added by branch_prep_bwd while it loads the
    row; the separate gather launch that did the same bits was retired in round 3.)"""
    from m2trans_amd import _lib
    from tests.test_gpu_baseline_configs import fwd_bwd
    scale, nb, B, H0, W0 = 4, 2, 2, 60, 90
    x = O.closed_form_image(B, 3, H0, W0).cuda()
    hr = O.closed_form_image(B, 3, H0 * scale, W0 * scale, phase=0.7).cuda()
    outs = []
    for fused in (1, 0):
        model, _ = build_model(scale, nb, "bf16")
        plan = model._plan_for(x)
        _lib.check(_lib.load().m2t_set_option(plan.handle, b"attn_bwd", 2 if fused else 1), "m2t_set_option")
        assert plan.query("opt:attn_bwd") == (2 if fused else 1)
        sr, loss, grads = fwd_bwd(model, x, hr, hr.numel())
        outs.append((sr.cpu(), grads.cpu(), model.param_offsets()))
    (sa, ga, offs), (sb, gb, _) = outs
    assert torch.equal(sa, sb)                                  # the forward pass is untouched
    total = float(gb.double().norm())
    assert float((ga.double() - gb.double()).norm()) / total < 3e-3, float((ga.double() - gb.double()).norm()) / total
    for n, (o, k) in offs.items():
        a, b = ga[o:o + k].double(), gb[o:o + k].double()
        d = float((a - b).norm())
        assert d <= 2e-2 * float(b.norm()) or d <= 1e-5 * total, (n, d / max(float(b.norm()), 1e-30), d / total)


def test_c16_gather_projection_prep_kernel_is_bit_identical_to_the_three_kernels():
    """bf16, C = 16 branch: halo overlap-add + projection data gradient + branch_prep_bwd in one kernel (c16_dgrad_prep_kernel,
    "attn_bwd" = 3, default) against halo_gather + gemm_nt + branch_prep_bwd ("attn_bwd" = 2): same fp32 adds in the same order,
    the same two 32-deep MFMAs per 16 pixels, the same rounding points -- every gradient of the step identical, at sizes with
    image-border, edge and interior windows (60 x 90 reflect-padded to 64 x 96) and at 128 x 128."""
    from m2trans_amd import _lib
    for (B, H, W) in ((2, 60, 90), (4, 128, 128)):
        scale, nb = 4, 2
        x = O.closed_form_image(B, 3, H, W).cuda()
        hr = O.closed_form_image(B, 3, H * scale, W * scale, phase=0.7).cuda()
        outs = []
        for level in (3, 2):
            model, _ = build_model(scale, nb, "bf16")
            plan = model._plan_for(x)
            _lib.check(_lib.load().m2t_set_option(plan.handle, b"attn_bwd", level), "m2t_set_option")
            assert plan.query("opt:attn_bwd") == level
            sr = model(x)
            torch.nn.L1Loss()(sr, hr).backward()
            outs.append({n: q.grad.clone() for n, q in model.named_parameters() if q.requires_grad})
        for n in outs[0]:
            assert torch.equal(outs[0][n], outs[1][n]), (B, H, W, n)


def test_row_streaming_conv3x3_is_bit_identical_to_the_tile_kernel():
    """bf16 3x3 conv 64 -> 64: the row-streaming kernel (LDS-DMA rings for the input and residual rows, weights in registers,
    option conv_rows, default) keeps the products and their order of the tile kernel: with the same InstanceNorm statistics the
    whole step -- eight forward convs, eight data gradients -- is reproduced bit for bit from run to run (the DMA-depth-3 and pipelined-
    epilogue variants that were checked against it through round 3 were retired in round 4).
    The tile kernel's path takes the statistics with the two-stage kernel instead (another summation order: mean / rstd differ in the
    last bits), so against it the FIRST conv output, whose input statistics are common, must be identical and the step agree to
    bf16 noise.  Sizes: 128x128 batch 8 (256 segments of 32 rows), 96x160 batch 3 (5 strips, first / last at the image border) and
    60x90 reflect-padded to 64x96 batch 2 (segments of 16 rows)."""
    from m2trans_amd import _lib
    from tests.gpu_util import ws_nchw
    for (B, H, W) in ((8, 128, 128), (3, 96, 160), (2, 60, 90)):
        scale, nb = 4, 2
        x = O.closed_form_image(B, 3, H, W).cuda()
        hr = O.closed_form_image(B, 3, H * scale, W * scale, phase=0.7).cuda()
        outs = []
        for rows in (1, 1, 0):             # row-streaming kernel (twice: reproducible bits), tile kernel
            model, _ = build_model(scale, nb, "bf16")
            plan = model._plan_for(x)
            _lib.check(_lib.load().m2t_set_option(plan.handle, b"conv_rows", rows), "m2t_set_option")
            _lib.check(_lib.load().m2t_set_option(plan.handle, b"fused_conv_bwd", 0), "m2t_set_option")     # the data gradient is this kernel too
            assert plan.query("opt:conv_rows") == rows and plan.query("opt:fused_conv_bwd") == 0
            sr = model(x)
            torch.nn.L1Loss()(sr, hr).backward()
            torch.cuda.synchronize()
            Hp, Wp = plan.query("padded_h"), plan.query("padded_w")
            outs.append((sr.detach().clone(), torch.cat([q.grad.reshape(-1) for _, q in model.named_parameters() if q.requires_grad]).clone(),
                         ws_nchw(plan, "X1", B, Hp, Wp, 64)))
        for o in outs[1:2]:
            assert torch.equal(outs[0][0], o[0]), (B, H, W)
            assert torch.equal(outs[0][1], o[1]), (B, H, W)
        tile = outs[2]
        assert torch.equal(outs[0][2], tile[2]), (B, H, W)                      # first block: same statistics, same bits
        assert rms_rel(outs[0][0], tile[0]) < 5e-2, (B, H, W, rms_rel(outs[0][0], tile[0]))     # (as the other fused-vs-plain A/Bs: closed-form weights amplify last-bit flips)
        gd = float((outs[0][1].double() - tile[1].double()).norm()) / float(tile[1].double().norm())
        assert gd < 5e-2, (B, H, W, gd)


def test_conv_epilogue_statistics_match_the_two_stage_kernel():
    """The InstanceNorm statistics a block takes from the previous block's conv epilogue (per-segment (n, mean, M2) partials of the
    stored bf16 values, merged by the same second stage) against mean / rstd recomputed in fp64 from the stored block input: the
    same quantity the two-stage kernel measures, within fp32 rounding (1e-5 relative on rstd, 1e-6 of the value scale on mean)."""
    from tests.gpu_util import ws_nchw
    for (B, H, W) in ((4, 128, 128), (2, 60, 90)):
        scale, nb = 4, 3
        x = O.closed_form_image(B, 3, H, W).cuda()
        model, _ = build_model(scale, nb, "bf16")
        plan = model._plan_for(x)
        with torch.no_grad():
            model(x)
        torch.cuda.synchronize()
        Hp, Wp = plan.query("padded_h"), plan.query("padded_w")
        for b in (1, 2):
            X = ws_nchw(plan, f"X{b}", B, Hp, Wp, 64).double()
            mean = X.mean(dim=(2, 3))
            rstd = 1.0 / torch.sqrt(X.var(dim=(2, 3), unbiased=False) + 1e-5)
            got_m = plan.ws_tensor(f"b{b}.mean", dtype=torch.float32).reshape(B, 64).double().cpu()
            got_r = plan.ws_tensor(f"b{b}.rstd", dtype=torch.float32).reshape(B, 64).double().cpu()
            assert float((got_m - mean).abs().max()) <= 1e-6 * float(X.abs().max()) + 1e-7, (B, H, W, b)
            assert float(((got_r - rstd) / rstd).abs().max()) <= 1e-5, (B, H, W, b, float(((got_r - rstd) / rstd).abs().max()))


def test_fused_conv_backward_matches_the_two_kernel_path():
    """bf16 3x3 conv 64 -> 64 backward in one pass (conv3x3_c64_bwd_rows_kernel, option fused_conv_bwd, default): its data-gradient
    waves run the row-streaming kernel's products in the same order, so every gradient that flows THROUGH the conv -- all
    parameters but the conv's own -- is bit-identical to the two-kernel path; the conv weight / bias gradients are the same bf16
    products accumulated in fp32 in another order (one 32-pixel row per MFMA, <= 256 partial slabs): within 1e-5 of the gradient's
    norm.  Sizes as in the row-streaming test: segments of 32 and 16 rows, strips at the image border, reflect padding."""
    from m2trans_amd import _lib
    for (B, H, W) in ((8, 128, 128), (3, 96, 160), (2, 60, 90)):
        scale, nb = 4, 2
        x = O.closed_form_image(B, 3, H, W).cuda()
        hr = O.closed_form_image(B, 3, H * scale, W * scale, phase=0.7).cuda()
        outs = []
        for fused in (1, 0):
            model, _ = build_model(scale, nb, "bf16")
            plan = model._plan_for(x)
            _lib.check(_lib.load().m2t_set_option(plan.handle, b"fused_conv_bwd", fused), "m2t_set_option")
            assert plan.query("opt:fused_conv_bwd") == fused
            sr = model(x)
            torch.nn.L1Loss()(sr, hr).backward()
            outs.append((sr.detach().clone(), {n: q.grad.clone() for n, q in model.named_parameters() if q.requires_grad}))
        (sa, ga), (sb, gb) = outs
        assert torch.equal(sa, sb)
        for n in ga:
            if ".feed_forward.0." in n:
                d = float((ga[n].double() - gb[n].double()).norm()) / float(gb[n].double().norm())
                assert d < 1e-5, (B, H, W, n, d)
            else:
                assert torch.equal(ga[n], gb[n]), (B, H, W, n)


@pytest.mark.parametrize("scale,B,H,W", [(3, 2, 40, 56), (3, 1, 96, 128), (2, 2, 60, 90), (2, 1, 128, 64), (3, 2, 32, 32), (2, 1, 32, 64),
                                         (3, 1, 160, 64), (2, 1, 192, 32)])
def test_x2_x3_row_streaming_tail_matches_the_plain_kernels(scale, B, H, W):
    """bf16 x2 / x3: the whole tail (tail.0 expansion + PixelShuffle(r) + GELU + tail conv) as ONE row-streaming forward kernel and ONE
    recomputing backward kernel (option fused_tail >= 1, default; k_tail_stream.hip / k_tail_bwd_stream.hip) against tail_expand +
    final_conv_fwd / final_conv_dgrad + final_conv_wgrad + gemm_nt + wgrad_tn (fused_tail = 0), which store gelu(t) / gelu'(t) / g(t).
    The forward keeps operand fragments, k order, GELU and tap order: sr bit for bit.  The backward rounds the same tensors to bf16
    (g(t), gelu'(t)) and contracts the expansion's data gradient in the plain GEMM's order: the gradient it hands to the body, gT, is
    BIT-identical (measured: zero differing elements at every size), hence so is every gradient of the body and the head; the three
    tail parameter gradients are fp32 sums over all pixels taken in another order (slabs per workgroup): 1.4e-7 relative at most
    (measured), gated at 2e-6.  A dropped border strip or segment seam would show as 1e-3 .. 1e-1 here (round-4 advisor: the old 2e-2
    gate could not see it).  Reflect-padded inputs, border / interior strips, ONE row segment (H = 32) and several (H = 160, 192)."""
    from m2trans_amd import _lib
    nb = 1
    x = O.closed_form_image(B, 3, H, W).cuda()
    hr = O.closed_form_image(B, 3, H * scale, W * scale, phase=0.7).cuda()
    outs = []
    for fused in (3, 0):
        model, _ = build_model(scale, nb, "bf16")
        plan = model._plan_for(x)
        _lib.check(_lib.load().m2t_set_option(plan.handle, b"fused_tail", fused), "m2t_set_option")
        assert plan.query("opt:fused_tail") == fused and plan.query("stores_t1") == (0 if fused else 1)
        sr = model(x)
        torch.nn.L1Loss()(sr, hr).backward()
        torch.cuda.synchronize()
        outs.append((sr.detach().clone(), {n: q.grad.detach().clone() for n, q in model.named_parameters() if q.requires_grad},
                     plan.ws_tensor("gT").clone()))
    (sa, ga, ta), (sb, gb, tb) = outs
    assert torch.equal(sa, sb), (scale, B, H, W, float((sa - sb).abs().max()))
    assert torch.equal(ta, tb), (scale, B, H, W, float((ta.float() - tb.float()).abs().max()))
    for n in gb:
        if n.startswith("tail."):
            d = float((ga[n].double() - gb[n].double()).norm() / gb[n].double().norm())
            assert d < 2e-6, (scale, n, d)
        else:
            assert torch.equal(ga[n], gb[n]), (scale, n)


def test_fused_forward_tail_and_recomputing_backward_are_bit_identical():
    """bf16 x4: tail.3 expansion + PixelShuffle + GELU + tail conv in one kernel (gelu(t2) / gelu'(t2) never stored) -- the
    row-streaming kernel of round 4 (option fused_tail = 3, default) and the 16x16-tile kernel it replaced (= 2) -- and the
    fused tail backward that recomputes them per tile, against the kernels that store and re-read them (= 1).  Same operand
    fragments, k order, bias add, GELU evaluation and tap summation -> the output and EVERY gradient must agree bit for
    bit.  Sizes: a reflect-padded 40x56 input (64x64 padded: border strips / tiles only), 128x96 (interior too) and 72x200
    (several row segments and strips, the last ones clamped)."""
    from m2trans_amd import _lib
    for (B, H, W) in ((2, 40, 56), (3, 128, 96), (1, 72, 200)):
        scale, nb = 4, 1
        x = O.closed_form_image(B, 3, H, W).cuda()
        hr = O.closed_form_image(B, 3, H * scale, W * scale, phase=0.7).cuda()
        outs = []
        for fused in (3, 2, 1, 4):
            model, _ = build_model(scale, nb, "bf16")
            plan = model._plan_for(x)
            _lib.check(_lib.load().m2t_set_option(plan.handle, b"fused_tail", fused), "m2t_set_option")
            assert plan.query("opt:fused_tail") == fused and plan.query("stores_t2") == (1 if fused == 1 else 0)
            sr = model(x)
            torch.nn.L1Loss()(sr, hr).backward()
            outs.append((sr.detach().clone(), torch.cat([q.grad.reshape(-1) for _, q in model.named_parameters() if q.requires_grad]).clone()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), (B, H, W)     # 3 vs 2: the two forward kernels
        # option 1 runs the STORED form of the tile backward (16x16x32 MFMAs); the recomputing form is the 32x32x16 kernel since round 6:
        # g(t1) -- hence every gradient upstream of the tail -- and dW3 stay bit-identical, dWf / db3 sum the same products in another order.
        # option 4: the row-streaming BACKWARD (kept for A/B): g(t1) bit-identical, the three tail parameter gradients in another order
        model, _ = build_model(scale, nb, "bf16")
        offs = model.param_offsets()
        for other in (outs[2], outs[3]):
            assert torch.equal(outs[0][0], other[0]), (B, H, W)
            for n, (o, k) in offs.items():
                a, b = outs[0][1][o:o + k], other[1][o:o + k]
                if n in ("tail.3.weight", "tail.3.bias", "tail.6.weight"):
                    assert float((a.double() - b.double()).norm()) <= 1e-5 * float(a.double().norm()), (B, H, W, n)
                else:
                    assert torch.equal(a, b), (B, H, W, n)


@pytest.mark.parametrize("shape", [(2, 128, 128), (3, 40, 56), (1, 72, 200), (1, 32, 32)])
def test_tail_backward_on_32x32x16_mfma_matches_the_16x16x32_kernel(shape):
    """Option tail_bwd_mfma32 (round 6, default): the recomputing x4 tail backward rebuilt around v_mfma_f32_32x32x16_bf16 -- LDS pixel order
    by sub-pixel position, gelu'(t2) / g(t2) formed in registers, branch-free reflect-border gather, dW3 on four waves and g(t1) on the other
    four -- against the 16x16x32 kernel of rounds 2-5, through the whole step (TrainStep, L1 seed inside the kernel; and the immediate-seed
    arm).  Same products and the same bf16 rounding points: the loss, g(t1) -- i.e. every gradient upstream of the tail -- and dW3 are
    bit-identical; dWf and db3 are summed in another fp32 order.  Reflect-padded sizes put border tiles on every side."""
    from m2trans_amd import _lib
    from m2trans_amd.train_step import TrainStep
    B, H, W = shape
    scale, nb = 4, 1
    x = O.closed_form_image(B, 3, H, W).cuda()
    hr = O.closed_form_image(B, 3, H * scale, W * scale, phase=0.7).cuda()
    for fused_l1 in (1, 0):
        res = []
        for val in (1, 0):
            model, _ = build_model(scale, nb, "bf16")
            plan = model._plan_for(x)
            _lib.check(_lib.load().m2t_set_option(plan.handle, b"tail_bwd_mfma32", val), "m2t_set_option")
            _lib.check(_lib.load().m2t_set_option(plan.handle, b"fused_l1", fused_l1), "m2t_set_option")
            assert plan.query("opt:tail_bwd_mfma32") == val
            ts = TrainStep(model, lr=1e-4, world_size=1)
            for step in range(2):                            # (a second step: the strip accumulators and LDS buffers start clean)
                loss = ts.forward_backward(x, hr)
            torch.cuda.synchronize()
            res.append((float(loss), ts.grads.clone(), model.param_offsets()))
        (l1, g1, offs), (l0, g0, _) = res
        assert l1 == l0, (l1, l0)
        for n, (o, k) in offs.items():
            a, b = g1[o:o + k], g0[o:o + k]
            if n in ("tail.3.bias", "tail.6.weight"):
                assert float((a.double() - b.double()).norm()) <= 1e-5 * float(b.double().norm()), (shape, n)
            else:
                assert torch.equal(a, b), (shape, fused_l1, n)


@pytest.mark.parametrize("scale,B,H,W", [(4, 2, 64, 64), (4, 1, 96, 128), (4, 1, 96, 32), (2, 3, 40, 56)])
def test_fp32_mfma_32x32x2_gemms_match_the_16x16x4_kernels(scale, B, H, W):
    """fp32 parity mode, option fp32_fast (default): the kernels of round 5 -- the qkv projections, their data gradients and weight gradients
    and the x2 expansions (+ their weight / bias gradients) on v_mfma_f32_32x32x2_f32 (k_gemm.hip), the tail conv weight and data gradient
    on the VALU (k_conv.hip) -- against the 16x16x4 kernels of rounds 1-4.  Both sides are exact fp32 products accumulated in fp32; only the
    summation ORDER differs.  The loss is a mean squared error, not L1: an L1 seed is sign(sr - hr), and a 1e-7 change of sr flips signs, which
    says nothing about the kernels.  sr within 2e-6 of its scale; the gradients of the tail, the head and the convs within 3e-5 of their norm;
    the attention parameters (qkv, rel_h / rel_w) within 3e-3: on these smooth closed-form images their gradients are 1e-7 ... 1e-11 -- sums
    that cancel almost completely -- so that rounding-order noise shows at 1e-4 ... 6e-4 of them (measured), while a wrong tile, a dropped
    slab or a wrong index is an O(1) error in the tensor it touches.  The oracle gates of the fp32 tests above run on this default path.
    Shapes: image rows of 64 / 128 / 256 pixels at the two expansions (the 32x32x2 expansion kernel needs whole 128-pixel tiles per row:
    both kernels are exercised), reflect padding, odd window counts."""
    from m2trans_amd import _lib
    nb = 2
    x = O.closed_form_image(B, 3, H, W).cuda()
    hr = O.closed_form_image(B, 3, H * scale, W * scale, phase=0.7).cuda()
    outs = []
    for fast in (1, 0):
        model, _ = build_model(scale, nb, "fp32")
        plan = model._plan_for(x)
        _lib.check(_lib.load().m2t_set_option(plan.handle, b"fp32_fast", fast), "m2t_set_option")
        assert plan.query("opt:fp32_fast") == fast
        sr = model(x)
        ((sr - hr) ** 2).mean().backward()
        outs.append((sr.detach().clone(), {n: q.grad.clone() for n, q in model.named_parameters() if q.requires_grad}))
    (sa, ga), (sb, gb) = outs
    assert float((sa - sb).abs().max()) <= 2e-6 * max(1.0, float(sb.abs().max())), float((sa - sb).abs().max())
    for n in ga:
        d = float((ga[n].double() - gb[n].double()).norm()) / max(float(gb[n].double().norm()), 1e-30)
        tol = 3e-3 if (".attn" in n) else 3e-5
        assert d < tol, (n, d)


@pytest.mark.gpu
def test_bench_two_rank_control_flow_on_one_device():
    """`bench.py --gpus 2 --selftest-shared-device`: the REAL N-rank flow of the bench (GPU-free parent -> child torch.distributed.run -> two
    ranks; headline workload with a collective in every step, the two extra all-event steps that EVERY rank must run, configs[3] behind it in
    `also`, MAX-over-ranks time, rank-0-only line) with both ranks on device 0 and gloo as the transport, because the boxes of this pool have
    one GPU.  It cannot price RCCL; it proves that no rank-0-only branch contains a step (a collective only one rank enters never returns --
    the first version of `roofline.others` had exactly that) and that the line carries the audit fields.  The line says `invalid`."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--selftest-shared-device", "--steps", "3", "--warmup", "2"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["invalid"] is True and d["value"] is None and d["n_gpus"] == 2 and d["steps"] == 3
    assert d["config"]["world_size"] == 2 and d["config"]["backend"] == "gloo" and d["rank_devices"] == [0, 0]
    assert d["config"]["per_gpu_batch"] == 16 and d["config"]["global_batch"] == 32          # the headline's per-GPU work at every N
    assert d["config"]["grad_exchange"].startswith("bucketed all-reduce") and d["exposed_comm_ms_per_step"] is not None
    assert d["config"]["hw_queues"] == "8"
    assert d["roofline"] is not None and len(d["roofline"]["others"]) == 5                   # the extra steps ran (on both ranks) and returned
    (a,) = d["also"]
    assert a["workload"] == "config3" and a["n_gpus"] == 2 and a["per_gpu_batch"] == 32 and a["global_batch"] == 64 and a["ms_per_step"] > 0
