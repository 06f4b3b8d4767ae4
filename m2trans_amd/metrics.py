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
