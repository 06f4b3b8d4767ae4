"""-m gpu: operator-level parity of the HIP kernels (through the C ABI) against the oracle."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import m2trans_oracle as O
from tests.gpu_util import nchw_to_nhwc, nhwc_to_nchw, rel, rms_rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from m2trans_amd import _lib
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return _lib


def _st(lib):
    return lib.stream_ptr()


def test_library_version(lib):
    assert lib.load().m2t_version() >= 100


@pytest.mark.parametrize("levels", [1, 2])
def test_dwt_iwt_bit_exact_fp32(lib, levels):
    """DWT/IWT (models/M2Trans_network.py:198-237) restated with the reference's association
    order: float32 results must be bit-identical."""
    L = lib.load()
    B, Cc, H, W = 2, 16, 24, 32
    x = (O.closed_form_image(B, Cc, H, W, phase=0.3) - 0.5).contiguous()
    want = x
    for _ in range(levels):
        want = O.dwt(want)
    xg = nchw_to_nhwc(x.cuda())
    S = 2 ** levels
    out = torch.empty(B, H // S, W // S, Cc * 4 ** levels, device="cuda")
    lib.check(L.m2t_dwt(lib.F32, levels, lib.ptr(xg), lib.ptr(out), B, H, W, Cc, _st(lib)), "m2t_dwt")
    got = nhwc_to_nchw(out).cpu()
    assert torch.equal(got, want)
    back = torch.empty(B, H, W, Cc, device="cuda")
    lib.check(L.m2t_iwt(lib.F32, levels, lib.ptr(out), lib.ptr(back), B, H, W, Cc, _st(lib)), "m2t_iwt")
    want_back = want
    for _ in range(levels):
        want_back = O.iwt(want_back)
    assert torch.equal(nhwc_to_nchw(back).cpu(), want_back)
    assert float((nhwc_to_nchw(back).cpu() - x).abs().max()) < 1e-6      # round trip


def test_dwt_golden_fixture(lib, golden_dir):
    """Against the committed output of the REAL reference's DWT/IWT modules."""
    import os
    g = np.load(os.path.join(golden_dir, "modules.npz"))
    L = lib.load()
    x = (O.closed_form_image(2, 16, 24, 32, phase=0.3, dtype=torch.float64) - 0.5).float()
    xg = nchw_to_nhwc(x.cuda())
    out = torch.empty(2, 12, 16, 64, device="cuda")
    lib.check(L.m2t_dwt(lib.F32, 1, lib.ptr(xg), lib.ptr(out), 2, 24, 32, 16, _st(lib)), "m2t_dwt")
    assert torch.equal(nhwc_to_nchw(out).cpu(), torch.from_numpy(g["dwt"]))
    back = torch.empty(2, 24, 32, 16, device="cuda")
    lib.check(L.m2t_iwt(lib.F32, 1, lib.ptr(out), lib.ptr(back), 2, 24, 32, 16, _st(lib)), "m2t_iwt")
    assert torch.equal(nhwc_to_nchw(back).cpu(), torch.from_numpy(g["iwt"]))


@pytest.mark.parametrize("r", [2, 3])
def test_pixel_shuffle_bit_exact(lib, r):
    L = lib.load()
    B, Cc, H, W = 2, 5, 7, 9
    x = torch.randn(B, Cc * r * r, H, W, generator=torch.Generator().manual_seed(1))
    want = torch.nn.functional.pixel_shuffle(x, r)
    xg = x.cuda()
    out = torch.empty(B, Cc, H * r, W * r, device="cuda")
    lib.check(L.m2t_pixel_shuffle(lib.ptr(xg), lib.ptr(out), B, Cc, H, W, r, _st(lib)), "m2t_pixel_shuffle")
    assert torch.equal(out.cpu(), want)
    back = torch.empty_like(xg)
    lib.check(L.m2t_pixel_unshuffle(lib.ptr(out), lib.ptr(back), B, Cc, H, W, r, _st(lib)), "m2t_pixel_unshuffle")
    assert torch.equal(back.cpu(), x)


def _attn_inputs(B, Cc, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    qkv = torch.randn(B, 3 * Cc, h, w, generator=g) * 0.7
    rel_h = torch.randn(1, 10, 1, Cc // 2, generator=g) * 0.8
    rel_w = torch.randn(1, 1, 10, Cc // 2, generator=g) * 0.8
    return qkv, rel_h, rel_w


@pytest.mark.parametrize("Cc,h,w", [(16, 16, 24), (64, 16, 16), (256, 8, 16)])
@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_window_attention_fwd(lib, Cc, h, w, dt):
    """TBlock attention core incl. zero-pad phantom keys at the borders."""
    L = lib.load()
    B = 2
    qkv, rel_h, rel_w = _attn_inputs(B, Cc, h, w, 3)
    code = lib.F32 if dt == "fp32" else lib.BF16
    tdt = torch.float32 if dt == "fp32" else torch.bfloat16
    qkv_dev = nchw_to_nhwc(qkv.cuda()).to(tdt)
    qkv_ref = nhwc_to_nchw(qkv_dev.float()).cpu()      # what the kernel actually sees
    q, k, v = torch.chunk(qkv_ref, 3, dim=1)
    want = O.window_attention_core(q, k, v, rel_h, rel_w)
    out = torch.empty(B, h, w, Cc, device="cuda", dtype=tdt)
    rh, rw = rel_h.reshape(-1).cuda(), rel_w.reshape(-1).cuda()
    lib.check(L.m2t_window_attention_fwd(code, lib.ptr(qkv_dev), lib.ptr(rh), lib.ptr(rw), lib.ptr(out), B, h, w, Cc,
                                         _st(lib)), "m2t_window_attention_fwd")
    got = nhwc_to_nchw(out.float()).cpu()
    tol = 2e-5 if dt == "fp32" else 2e-2
    assert rel(got, want) < tol, (rel(got, want), rms_rel(got, want))


@pytest.mark.parametrize("Cc,h,w", [(16, 16, 24), (64, 16, 16), (256, 8, 16)])
@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_window_attention_bwd(lib, Cc, h, w, dt):
    """dq/dk/dv (halo overlap-add over <=4 windows) and the rel-pos gradients (which sum over
    phantom positions too) against CPU autograd through the oracle."""
    L = lib.load()
    B = 2
    qkv, rel_h, rel_w = _attn_inputs(B, Cc, h, w, 5)
    code = lib.F32 if dt == "fp32" else lib.BF16
    tdt = torch.float32 if dt == "fp32" else torch.bfloat16
    g = torch.Generator().manual_seed(9)
    gout = torch.randn(B, Cc, h, w, generator=g)
    qkv_dev = nchw_to_nhwc(qkv.cuda()).to(tdt)
    gout_dev = nchw_to_nhwc(gout.cuda()).to(tdt)
    qkv_ref = nhwc_to_nchw(qkv_dev.float()).cpu().requires_grad_(True)
    gout_ref = nhwc_to_nchw(gout_dev.float()).cpu()
    rh_ref, rw_ref = rel_h.clone().requires_grad_(True), rel_w.clone().requires_grad_(True)
    q, k, v = torch.chunk(qkv_ref, 3, dim=1)
    out = O.window_attention_core(q, k, v, rh_ref, rw_ref)
    out.backward(gout_ref)
    gq = torch.empty(B, h, w, 3 * Cc, device="cuda", dtype=tdt)
    grh = torch.empty(10 * Cc // 2, device="cuda")
    grw = torch.empty(10 * Cc // 2, device="cuda")
    nb = L.m2t_window_attention_bwd_scratch_bytes(code, B, h, w, Cc)
    scratch = torch.empty(nb, dtype=torch.uint8, device="cuda")
    rh, rw = rel_h.reshape(-1).cuda(), rel_w.reshape(-1).cuda()
    lib.check(L.m2t_window_attention_bwd(code, lib.ptr(qkv_dev), lib.ptr(rh), lib.ptr(rw), lib.ptr(gout_dev), lib.ptr(gq),
                                         lib.ptr(grh), lib.ptr(grw), lib.ptr(scratch), B, h, w, Cc, _st(lib)),
              "m2t_window_attention_bwd")
    got = nhwc_to_nchw(gq.float()).cpu()
    tol = 5e-5 if dt == "fp32" else 4e-2
    errs = {
        "dq": rel(got[:, :Cc], qkv_ref.grad[:, :Cc]),
        "dk": rel(got[:, Cc:2 * Cc], qkv_ref.grad[:, Cc:2 * Cc]),
        "dv": rel(got[:, 2 * Cc:], qkv_ref.grad[:, 2 * Cc:]),
        "drel_h": rel(grh.cpu(), rh_ref.grad.reshape(-1)),
        "drel_w": rel(grw.cpu(), rw_ref.grad.reshape(-1)),
    }
    assert all(e < tol for e in errs.values()), errs


def test_layout_roundtrip(lib):
    L = lib.load()
    x = torch.randn(2, 8, 5 * 6)
    xg = x.cuda()
    for code, tdt in ((lib.F32, torch.float32), (lib.BF16, torch.bfloat16)):
        nh = torch.empty(2, 30, 8, device="cuda", dtype=tdt)
        lib.check(L.m2t_to_nhwc(code, lib.ptr(xg), lib.ptr(nh), 2, 8, 30, _st(lib)), "to_nhwc")
        assert torch.equal(nh.float().cpu(), x.permute(0, 2, 1).to(tdt).float())
        back = torch.empty(2, 8, 30, device="cuda")
        lib.check(L.m2t_to_nchw(code, lib.ptr(nh), lib.ptr(back), 2, 8, 30, _st(lib)), "to_nchw")
        assert torch.equal(back.cpu(), x.to(tdt).float())
