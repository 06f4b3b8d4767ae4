"""Y-channel PSNR exactly as the reference's eval loop computes it (host side, fp64).

Restates utils.rgb_to_ycbcr (utils.py:121-146; note the `/255` applied to inputs that are already
in [0,1]), the scale-pixel border crop and x255 of test.py:101-111 / train.py:299-309, and
utils.calc_psnr (utils.py:179-184)."""
from __future__ import annotations

import math

import torch


def rgb_to_y(img: torch.Tensor) -> torch.Tensor:
    img = img / 255.0
    return (65.481 * img[..., 0, :, :] + 128.553 * img[..., 1, :, :] + 24.966 * img[..., 2, :, :] + 16.0).unsqueeze(-3)


def calc_psnr(sr: torch.Tensor, hr: torch.Tensor) -> float:
    diff = (sr.double() - hr.double()) / 255.0
    return float(-10.0 * math.log10(float(diff.pow(2).mean())))


def psnr_y(sr: torch.Tensor, hr: torch.Tensor, scale: int, rgb_range: float = 1.0) -> float:
    s, h = rgb_to_y(sr), rgb_to_y(hr)
    s = s[..., scale:-scale, scale:-scale]
    h = h[..., scale:-scale, scale:-scale]
    if rgb_range == 1:
        s, h = s * 255.0, h * 255.0
    return calc_psnr(s, h)


_WINDOW = None


def _ssim_window() -> torch.Tensor:
    """11-tap Gaussian, sigma 1.5, evaluated and normalised in fp32 with torch ops exactly as `pytorch_msssim`
    builds its window (host tensor, read by m2t_eval_metrics during the call)."""
    global _WINDOW
    if _WINDOW is None:
        c = torch.arange(11, dtype=torch.float32) - 11 // 2
        g = torch.exp(-(c ** 2) / (2 * 1.5 ** 2))
        _WINDOW = (g / g.sum()).contiguous()
    return _WINDOW


def y_metrics_device(sr: torch.Tensor, hr: torch.Tensor, scale: int, rgb_range: float = 1.0) -> torch.Tensor:
    """PSNR input and SSIM of the eval loop (test.py:101-113, utils.py:179-184,232-234) on the device:
    [B,3,H,W] float32 pairs -> float64 [B,2] = (mean squared Y error, mean SSIM) per image.  HIP kernels
    behind `m2t_eval_metrics` (k_metrics.hip); no host fallback for device tensors."""
    from . import _lib
    if sr.shape != hr.shape or sr.dim() != 4 or sr.shape[1] != 3:
        raise _lib.M2TError(f"expected two [B,3,H,W] tensors of equal shape, got {tuple(sr.shape)} and {tuple(hr.shape)}")
    if not (sr.is_cuda and hr.is_cuda):
        raise _lib.M2TError("y_metrics_device needs HIP device tensors (use psnr_y for host tensors)")
    lib = _lib.load()
    sr, hr = sr.contiguous().float(), hr.contiguous().float()
    B, _, H, W = sr.shape
    nbytes = lib.m2t_eval_metrics_scratch_bytes(B, H, W, scale)
    if nbytes == 0:
        raise _lib.M2TError(f"image {H}x{W} is too small for a border crop of {scale}")
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=sr.device)
    out = torch.empty(B, 2, dtype=torch.float64, device=sr.device)
    with torch.cuda.device(sr.device):
        _lib.check(lib.m2t_eval_metrics(_lib.ptr(sr), _lib.ptr(hr), B, H, W, scale, float(rgb_range), _ssim_window().data_ptr(),
                                        _lib.ptr(scratch), _lib.ptr(out), _lib.stream_ptr()), "m2t_eval_metrics")
    return out


def gmsd_device(x: torch.Tensor, y: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    """``piq.gmsd(x, y, data_range=data_range, reduction='none')`` (test.py:98) on the device: [B,3,H,W] float32 pairs ->
    float64 [B].  HIP kernels behind `m2t_eval_gmsd` (k_metrics.hip); parity unpinned (`piq` is not vendored)."""
    from . import _lib
    if x.shape != y.shape or x.dim() != 4 or x.shape[1] != 3:
        raise _lib.M2TError(f"expected two [B,3,H,W] tensors of equal shape, got {tuple(x.shape)} and {tuple(y.shape)}")
    if not (x.is_cuda and y.is_cuda):
        raise _lib.M2TError("gmsd_device needs HIP device tensors")
    lib = _lib.load()
    x, y = x.contiguous().float(), y.contiguous().float()
    B, _, H, W = x.shape
    scratch = torch.empty(lib.m2t_eval_gmsd_scratch_bytes(B), dtype=torch.uint8, device=x.device)
    out = torch.empty(B, dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.m2t_eval_gmsd(_lib.ptr(x), _lib.ptr(y), B, H, W, float(data_range), _lib.ptr(scratch), _lib.ptr(out),
                                     _lib.stream_ptr()), "m2t_eval_gmsd")
    return out


def fsim_device(x: torch.Tensor, y: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    """``piq.fsim(x, y, data_range=data_range, reduction='none')`` (test.py:95; FSIMc, piq 0.8.0 defaults) on the device: [B,3,H,W]
    float32 pairs -> float64 [B].  HIP kernels behind `m2t_eval_fsim` (k_fsim.hip); parity unpinned (`piq` is not vendored)."""
    from . import _lib
    if x.shape != y.shape or x.dim() != 4 or x.shape[1] != 3:
        raise _lib.M2TError(f"expected two [B,3,H,W] tensors of equal shape, got {tuple(x.shape)} and {tuple(y.shape)}")
    if not (x.is_cuda and y.is_cuda):
        raise _lib.M2TError("fsim_device needs HIP device tensors")
    lib = _lib.load()
    x, y = x.contiguous().float(), y.contiguous().float()
    B, _, H, W = x.shape
    nbytes = lib.m2t_eval_fsim_scratch_bytes(H, W)
    if nbytes == 0:
        raise _lib.M2TError(f"image {H}x{W} is too small for FSIM")
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    out = torch.empty(B, dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.m2t_eval_fsim(_lib.ptr(x), _lib.ptr(y), B, H, W, float(data_range), _lib.ptr(scratch), _lib.ptr(out),
                                     _lib.stream_ptr()), "m2t_eval_fsim")
    return out


def evaluate(model, pairs, scale: int, rgb_range: float = 1.0, with_gmsd: bool = False, with_fsim: bool = False):
    """The reference's test loop (test.py:77-122): `pairs` yields (lr, hr) device tensors [1,3,h,w] / [1,3,h*scale,w*scale];
    returns (avg_psnr, avg_ssim) -- with_gmsd: + avg_gmsd; with_fsim: + avg_fsim, in the order the reference prints them
    (PSNR, SSIM, FSIM, GMSD; test.py:118-122) -- rounded as the reference rounds them.  One host synchronisation at the end
    (the reference synchronises per image)."""
    rows, grows, frows = [], [], []
    with torch.no_grad():
        for lr, hr in pairs:
            sr = model(lr)
            if sr.shape != hr.shape:
                raise ValueError(f"hr {tuple(hr.shape)} does not match sr {tuple(sr.shape)}")
            if with_fsim:
                frows.append(fsim_device(hr, sr, 1.0))          # BEFORE the Y conversion, on RGB, like test.py:95-99
            if with_gmsd:
                grows.append(gmsd_device(hr, sr, 1.0))
            rows.append(y_metrics_device(sr, hr, scale, rgb_range))
    if not rows:
        raise ValueError("no evaluation pairs")
    m = torch.cat(rows).cpu()
    psnr = [-10.0 * math.log10(float(v)) for v in m[:, 0]]
    avg_psnr = round(sum(psnr) / len(psnr) + 5e-3, 2)
    avg_ssim = round(float(m[:, 1].sum()) / len(psnr) + 5e-5, 4)
    out = [avg_psnr, avg_ssim]
    if with_fsim:
        out.append(round(float(torch.cat(frows).sum()) / len(psnr) + 5e-5, 4))
    if with_gmsd:
        out.append(round(float(torch.cat(grows).sum()) / len(psnr) + 5e-5, 4))
    return tuple(out)
