"""Evaluation metrics of the reference's test loop (test.py:77-122): oracle anchors on the CPU, and the device
kernels behind m2t_eval_metrics against the oracle (-m gpu).

SSIM is "parity unpinned" (pytorch_msssim is neither vendored by the reference nor installed here): the oracle's
restatement is anchored on closed forms and on an independent scipy float64 evaluation instead."""
import math

import numpy as np
import pytest
import torch

from oracle import m2trans_oracle as O


def _pair(B, H, W, noise=0.0, seed=0):
    a = O.closed_form_image(B, 3, H, W, phase=0.2)
    b = a + 0.03 * (O.closed_form_image(B, 3, H, W, phase=1.9) - 0.5)
    if noise:
        g = torch.Generator().manual_seed(seed)
        a = a + noise * torch.randn(a.shape, generator=g)
        b = b + noise * torch.randn(a.shape, generator=g)
    return b.clamp(0, 1).contiguous(), a.clamp(0, 1).contiguous()


# ------------------------------------------------------------------------------------------------ oracle (CPU)
def test_fsim_oracle_anchors():
    """FSIMc (piq.fsim, test.py:95-96, piq 0.8.0; parity unpinned -- the package is in neither tree): identical images -> 1;
    symmetric in its arguments; more noise -> lower index; on grey images the chroma similarities are 1, so FSIMc == FSIM; the
    whole index through explicit DFT matrices instead of numpy's FFT (an independent evaluation of every transform) agrees to
    1e-12; odd image sizes and the k = 2 pooling branch (min(H, W) = 392 -> round(1.53) = 2) run through."""
    from oracle import fsim_oracle as F
    sr, hr = _pair(1, 72, 88, noise=0.03)
    a, b = hr[0].double().numpy(), sr[0].double().numpy()
    assert abs(F.fsim(a, a) - 1.0) < 1e-12
    v = F.fsim(a, b)
    assert 0.3 < v < 0.999 and abs(v - F.fsim(b, a)) < 1e-14
    c = np.clip(b + 0.1 * np.random.default_rng(0).standard_normal(b.shape), 0, 1)
    assert F.fsim(a, c) < v
    ex = F.fsim(a, b, fft2=lambda z: F.dft2_explicit(z), ifft2=lambda z: F.dft2_explicit(z, True))
    assert abs(ex - v) < 1e-12
    ga, gb = np.repeat(a[:1], 3, 0), np.repeat(b[:1], 3, 0)
    assert abs(F.fsim(ga, gb) - F.fsim(ga, gb, chromatic=False)) < 1e-15
    sr2, hr2 = _pair(1, 41, 67, noise=0.02)
    assert 0.3 < F.fsim(hr2[0].double().numpy(), sr2[0].double().numpy()) < 1.0
    sr3, hr3 = _pair(1, 392, 400, noise=0.02)
    assert 0.3 < F.fsim(hr3[0].double().numpy(), sr3[0].double().numpy()) < 1.0
    # the filter bank: zero response at zero frequency, symmetric spread, the low-pass kills the corners
    bank = F.construct_filters(64, 48)
    assert bank.shape == (16, 64, 48) and float(np.abs(bank[:, 0, 0]).max()) == 0.0 and float(bank.max()) <= 1.0 and float(bank.min()) >= 0.0
    assert float(bank[:, 32, 24].max()) < 1e-3


def test_gmsd_oracle_anchors():
    """GMSD (piq.gmsd, test.py:98; parity unpinned): identical images -> every similarity is 1 -> deviation 0; symmetric in its
    arguments; a tiny case evaluated by hand (2x2-pooled luminance, Prewitt / 3 with zero padding, t = 170 / 255^2)."""
    sr, hr = _pair(2, 64, 48, noise=0.05)
    assert float(O.gmsd(hr.double(), hr.double()).abs().max()) == 0.0
    a, b = O.gmsd(hr.double(), sr.double()), O.gmsd(sr.double(), hr.double())
    assert torch.allclose(a, b, rtol=0, atol=1e-15) and float(a.min()) > 0
    v = 0.6
    x = torch.zeros(1, 3, 4, 4, dtype=torch.float64)
    x[..., 2:] = v                                # left half 0, right half v -> pooled map [[0, v], [0, v]]
    y = torch.full_like(x, 0.3)
    g = float(O.gmsd(x, y)[0])
    c = 170.0 / 255.0 ** 2
    lum = lambda t: t * (0.299 + 0.587 + 0.114)

    def gm(p):
        q = np.pad(np.array(p, dtype=np.float64), 1)
        out = np.zeros((2, 2))
        for i in range(2):
            for j in range(2):
                w = q[i:i + 3, j:j + 3]
                gx = (w[:, 2] - w[:, 0]).sum() / 3.0
                gy = (w[2, :] - w[0, :]).sum() / 3.0
                out[i, j] = math.sqrt(gx * gx + gy * gy)
        return out
    gx_, gy_ = gm([[0.0, lum(v)], [0.0, lum(v)]]), gm([[lum(0.3)] * 2] * 2)
    gms = (2 * gx_ * gy_ + c) / (gx_ ** 2 + gy_ ** 2 + c)
    assert abs(g - float(np.sqrt(((gms - gms.mean()) ** 2).mean()))) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,noise", [(1, 72, 88, 0.03), (2, 41, 67, 0.02), (1, 392, 400, 0.02), (1, 512, 512, 0.05)])
def test_device_fsim_matches_oracle(B, H, W, noise):
    """m2t_eval_fsim (k_fsim.hip: fp64, explicit DFTs, radix-select median) against the fp64 restatement: 1e-9 on the index; identical
    images give exactly the index 1 up to rounding; odd sizes, the pooling branch, batches."""
    from oracle import fsim_oracle as F
    from m2trans_amd.metrics import fsim_device
    sr, hr = _pair(B, H, W, noise=noise)
    got = fsim_device(hr.cuda(), sr.cuda(), 1.0).cpu()
    for b in range(B):
        want = F.fsim(hr[b].double().numpy(), sr[b].double().numpy())
        assert abs(float(got[b]) - want) < 1e-9, (b, float(got[b]), want)
    same = fsim_device(hr.cuda(), hr.cuda(), 1.0).cpu()
    assert float((same - 1.0).abs().max()) < 1e-12


@pytest.mark.gpu
def test_device_fsim_rejects_host_tensors():
    from m2trans_amd import _lib
    from m2trans_amd.metrics import fsim_device
    sr, hr = _pair(1, 32, 32)
    with pytest.raises(_lib.M2TError):
        fsim_device(hr, sr)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,noise", [(2, 128, 128, 0.05), (1, 99, 75, 0.1), (1, 64, 51, 0.02), (3, 33, 40, 0.0)])
def test_device_gmsd_matches_oracle(B, H, W, noise):
    from m2trans_amd.metrics import gmsd_device
    sr, hr = _pair(B, H, W, noise=noise, seed=W)
    got = gmsd_device(hr.cuda(), sr.cuda(), 1.0).cpu()
    want64 = O.gmsd(hr.double(), sr.double())
    want32 = O.gmsd(hr, sr).double()
    # fp32 map arithmetic (like the dependency), fp64 moments: within fp32 rounding of the fp64 formula and of the fp32 one
    assert torch.allclose(got, want64, rtol=0, atol=2e-6), (got, want64)
    assert torch.allclose(got, want32, rtol=0, atol=2e-6), (got, want32)
    assert float(gmsd_device(hr.cuda(), hr.cuda()).abs().max()) == 0.0


def test_mse_y_is_what_psnr_y_takes_the_log_of():
    sr, hr = _pair(1, 48, 40)
    assert abs(-10.0 * math.log10(float(O.mse_y(sr, hr, 4)[0])) - O.psnr_y(sr, hr, 4)) < 1e-12


def test_ssim_oracle_identity_symmetry_and_constant_closed_form():
    sr, hr = _pair(2, 40, 56, noise=0.05)
    one = O.ssim_y(hr, hr, 4, dtype=torch.float64)
    assert torch.allclose(one, torch.ones(2, dtype=torch.float64), atol=1e-12)
    assert torch.allclose(O.ssim_y(sr, hr, 4, dtype=torch.float64), O.ssim_y(hr, sr, 4, dtype=torch.float64), atol=1e-14)
    # constant images: every sigma is 0, the map is (2ab + C1) / (a^2 + b^2 + C1) everywhere
    x = torch.full((1, 3, 32, 32), 0.25)
    y = torch.full((1, 3, 32, 32), 0.75)
    a = float(O.y_channel_eval(x, 2).double()[0, 0, 0, 0])
    b = float(O.y_channel_eval(y, 2).double()[0, 0, 0, 0])
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    assert abs(float(O.ssim_y(x, y, 2, dtype=torch.float64)[0]) - (2 * a * b + C1) / (a * a + b * b + C1)) < 1e-4
    # exactly: the fp32-normalised window sums to S = 1 +- 1e-7, which leaves sigma = a^2 S (1 - S) != 0
    S = float(O.ssim_window(torch.float64).sum()) ** 2
    m1, m2 = a * S, b * S
    s1, s2, s12 = a * a * S - m1 * m1, b * b * S - m2 * m2, a * b * S - m1 * m2
    want = ((2 * m1 * m2 + C1) / (m1 * m1 + m2 * m2 + C1)) * ((2 * s12 + C2) / (s1 + s2 + C2))
    assert abs(float(O.ssim_y(x, y, 2, dtype=torch.float64)[0]) - want) < 1e-9


def test_ssim_oracle_against_independent_scipy_float64():
    from scipy.ndimage import correlate1d
    sr, hr = _pair(1, 64, 48, noise=0.03)
    X = O.y_channel_eval(sr, 3)[0, 0].double().numpy()
    Y = O.y_channel_eval(hr, 3)[0, 0].double().numpy()
    g = O.ssim_window(torch.float64).numpy()

    def filt(t):      # 'valid' part of a separable correlation
        t = correlate1d(correlate1d(t, g, axis=0, mode="constant"), g, axis=1, mode="constant")
        return t[5:-5, 5:-5]

    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    m1, m2 = filt(X), filt(Y)
    s1, s2, s12 = filt(X * X) - m1 * m1, filt(Y * Y) - m2 * m2, filt(X * Y) - m1 * m2
    want = (((2 * m1 * m2 + C1) / (m1 * m1 + m2 * m2 + C1)) * ((2 * s12 + C2) / (s1 + s2 + C2))).mean()
    assert abs(float(O.ssim_y(sr, hr, 3, dtype=torch.float64)[0]) - want) < 1e-10


def test_ssim_window_matches_published_values():
    g = O.ssim_window()
    assert g.dtype == torch.float32 and abs(float(g.sum()) - 1.0) < 1e-6
    # 11 taps, sigma 1.5: centre weight 0.26601 (Wang et al. 2004 window, 1-D marginal)
    assert abs(float(g[5]) - 0.266012) < 1e-5 and abs(float(g[0]) - float(g[10])) < 1e-9


def test_short_axis_is_not_filtered_like_the_dependency():
    sr, hr = _pair(1, 14, 40, noise=0.02)        # 10 rows after the crop of 2: the H axis is skipped
    v = O.ssim_y(sr, hr, 2, dtype=torch.float64)
    assert v.shape == (1,) and 0.0 < float(v[0]) < 1.0


# ------------------------------------------------------------------------------------------------ device
@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,scale,rgb_range,noise", [
    (2, 128, 128, 4, 1.0, 0.05),
    (1, 512, 512, 4, 1.0, 0.02),
    (3, 99, 75, 3, 1.0, 0.1),         # non-square, ragged tiles
    (1, 14, 40, 2, 1.0, 0.02),        # H axis shorter than the window after the crop
    (2, 64, 64, 2, 1.0, 0.0),         # smooth images: the ill-conditioned case for fp32
    (1, 48, 48, 2, 255.0, 0.05),      # rgb_range != 1: no x255
])
def test_device_metrics_match_oracle(B, H, W, scale, rgb_range, noise):
    from m2trans_amd.metrics import y_metrics_device
    sr, hr = _pair(B, H, W, noise=noise, seed=H)
    got = y_metrics_device(sr.cuda(), hr.cuda(), scale, rgb_range).cpu()
    mse = O.mse_y(sr, hr, scale, rgb_range)
    ssim64 = O.ssim_y(sr, hr, scale, rgb_range, dtype=torch.float64)
    # fp32 Y bit-identical + fp64 sums: only the summation order differs
    assert torch.allclose(got[:, 0], mse, rtol=1e-12, atol=0), (got[:, 0], mse)
    assert torch.allclose(got[:, 1], ssim64, rtol=0, atol=1e-10), (got[:, 1], ssim64)
    # the reference evaluates the same formula in fp32, where sigma = E[x^2] - mu^2 loses ~5 of 7 digits: its own
    # value moves by up to ~1e-3 with the convolution backend's summation order (measured 4e-4 between torch-CPU fp32
    # and fp64 on the 10x36 case).  Stated tolerance against an fp32 evaluation: 2e-3.
    ssim32 = O.ssim_y(sr, hr, scale, rgb_range).double()
    assert torch.allclose(got[:, 1], ssim32, rtol=0, atol=2e-3), (got[:, 1], ssim32)


@pytest.mark.gpu
def test_device_metrics_identity_and_errors():
    from m2trans_amd import _lib
    from m2trans_amd.metrics import y_metrics_device
    _, hr = _pair(2, 96, 64, noise=0.05)
    got = y_metrics_device(hr.cuda(), hr.cuda(), 4).cpu()
    assert float(got[:, 0].abs().max()) == 0.0 and torch.allclose(got[:, 1], torch.ones(2, dtype=torch.float64), atol=1e-12)
    with pytest.raises(_lib.M2TError):
        y_metrics_device(hr, hr, 4)                                   # host tensors: no CPU fallback
    with pytest.raises(_lib.M2TError):
        y_metrics_device(hr[:, :, :8, :8].cuda(), hr[:, :, :8, :8].cuda(), 4)   # nothing left after the crop


@pytest.mark.gpu
def test_evaluate_loop_matches_oracle_on_model_outputs():
    """test.py:77-122 end to end: model forward (fp32 compute) on odd-sized images, PSNR / SSIM averaged and
    rounded as the reference prints them."""
    import types
    from m2trans_amd.M2Trans_network import create_model
    from m2trans_amd.metrics import evaluate
    scale = 4
    args = types.SimpleNamespace(n_feats=64, scale=scale, rgb_range=1.0, n_blocks=1, colors=3, compute_dtype="fp32")
    torch.manual_seed(5)
    model = create_model(args).cuda().eval()
    sizes = [(40, 56), (33, 47)]
    pairs, ps, ss = [], [], []
    for i, (h, w) in enumerate(sizes):
        lr = O.closed_form_image(1, 3, h, w, phase=0.1 * i).cuda()
        hr = O.closed_form_image(1, 3, h * scale, w * scale, phase=0.1 * i + 0.05).cuda()
        pairs.append((lr, hr))
    got = evaluate(model, pairs, scale)
    got3 = evaluate(model, pairs, scale, with_gmsd=True)
    got4 = evaluate(model, pairs, scale, with_gmsd=True, with_fsim=True)
    from oracle import fsim_oracle as F
    gs, fs = [], []
    with torch.no_grad():
        for lr, hr in pairs:
            sr = model(lr).cpu()
            ps.append(O.psnr_y(sr, hr.cpu(), scale))
            ss.append(float(O.ssim_y(sr, hr.cpu(), scale, dtype=torch.float64)[0]))
            gs.append(float(O.gmsd(hr.cpu().double(), sr.double())[0]))
            fs.append(F.fsim(hr[0].cpu().double().numpy(), sr[0].double().numpy()))
    want = (round(sum(ps) / len(ps) + 5e-3, 2), round(sum(ss) / len(ss) + 5e-5, 4))
    assert got == want, (got, want)
    assert got3[:2] == want and abs(got3[2] - round(sum(gs) / len(gs) + 5e-5, 4)) <= 1e-4, (got3, gs)
    # the reference's print order: PSNR, SSIM, FSIM, GMSD (test.py:122)
    assert got4[:2] == want and got4[3] == got3[2] and abs(got4[2] - round(sum(fs) / len(fs) + 5e-5, 4)) <= 1e-4, (got4, fs)
