"""Helpers shared by the -m gpu parity tests (HIP path vs the CPU oracle)."""
from __future__ import annotations

import types

import torch

from oracle import m2trans_oracle as O


def make_args(scale=4, n_blocks=8, compute_dtype="fp32"):
    return types.SimpleNamespace(n_feats=64, scale=scale, rgb_range=1.0, n_blocks=n_blocks, colors=3,
                                 compute_dtype=compute_dtype)


def build_model(scale, n_blocks, compute_dtype="fp32", params=None, device="cuda"):
    from m2trans_amd.M2Trans_network import create_model
    model = create_model(make_args(scale, n_blocks, compute_dtype))
    if params is None:
        params = O.closed_form_params(64, scale, n_blocks)
    torch.nn.Module.load_state_dict(model, {k: v.clone() for k, v in params.items()}, strict=True)
    return model.to(device), params


def nchw_to_nhwc(t: torch.Tensor) -> torch.Tensor:
    return t.permute(0, 2, 3, 1).contiguous()


def nhwc_to_nchw(t: torch.Tensor) -> torch.Tensor:
    return t.permute(0, 3, 1, 2).contiguous()


def rel(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rms_rel(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-30))


def ws_nchw(plan, name, B, H, W, C):
    """Workspace tensor (NHWC in the plan's dtype) -> float32 NCHW on the CPU.  The 64-channel low-resolution
    feature maps (X_b, b*.xc) are chunk-planar ("P64": [4][B*H*W][16], csrc/m2t_common.h)."""
    import re
    t = plan.ws_tensor(name)
    if C == 64 and re.fullmatch(r"X\d+|b\d+\.xc", name):
        t = t.view(4, B, H, W, 16).permute(1, 2, 3, 0, 4).reshape(B, H, W, 64)
        return nhwc_to_nchw(t.float()).cpu()
    return nhwc_to_nchw(t.view(B, H, W, C).float()).cpu()


BR_C = (16, 64, 256, 256)
BR_L = (0, 1, 2, 2)


def hip_forward_trace(plan, scale, n_blocks, B, H, W):
    """Every stored activation of the last m2t_forward, read back from the workspace as float32 NCHW on the CPU, under the
    names the oracle's ``force`` / ``cap`` dicts use (H, W = padded LR size)."""
    t = {"X0": ws_nchw(plan, "X0", B, H, W, 64)}
    for b in range(n_blocks):
        for i in range(4):
            h, w = H >> BR_L[i], W >> BR_L[i]
            t[f"b{b}.d{i+1}"] = ws_nchw(plan, f"b{b}.d{i+1}", B, h, w, BR_C[i])
            if i > 1 or plan.query(f"stores_qkv{i+1}") == 1:   # (bf16 default: q | k | v of the C = 16 / 64 branches are recomputed by the backward)
                t[f"b{b}.qkv{i+1}"] = ws_nchw(plan, f"b{b}.qkv{i+1}", B, h, w, 3 * BR_C[i])
        t[f"b{b}.xc"] = ws_nchw(plan, f"b{b}.xc", B, H, W, 64)
        t[f"X{b+1}"] = ws_nchw(plan, f"X{b+1}", B, H, W, 64)
    r0 = 2 if scale == 4 else scale
    if plan.query("stores_t1") == 1:                       # (x2 / x3 bf16: the row-streaming tail keeps gelu(t) / gelu'(t) in registers)
        for nm in ("t1act", "t1der"):
            t[nm] = ws_nchw(plan, nm, B, H * r0, W * r0, 64)
    if scale == 4 and plan.query("stores_t2") == 1:        # the fused forward tail keeps gelu(t2) / gelu'(t2) in LDS
        for nm in ("t2act", "t2der"):
            t[nm] = ws_nchw(plan, nm, B, H * 4, W * 4, 64)
    t["srpre"] = plan.ws_tensor("srpre", dtype=torch.float32).view(B, 3, H * scale, W * scale).cpu().clone()
    return t


def smooth_hr(B, size, seed, device="cpu", noise=0.026):
    """Band-limited synthetic 'tissue' in [0,1] (the bf16 quality tests): a low-frequency Fourier field, a few soft-edged
    blobs and mild speckle that the x4 box down-sampling removes only partly -- an image family on which the network
    reaches >= 25 dB within a few hundred steps, so that PSNR differences mean something."""
    import torch.nn.functional as F
    g = torch.Generator(device="cpu").manual_seed(seed)
    yy = torch.linspace(0, 1, size).view(1, 1, size, 1)
    xx = torch.linspace(0, 1, size).view(1, 1, 1, size)
    img = torch.zeros(B, 1, size, size)
    for _ in range(12):
        fx, fy = (torch.rand(B, 1, 1, 1, generator=g) * 14 - 7), (torch.rand(B, 1, 1, 1, generator=g) * 14 - 7)
        ph = torch.rand(B, 1, 1, 1, generator=g) * 6.283
        amp = torch.rand(B, 1, 1, 1, generator=g) * 0.12
        img = img + amp * torch.sin(6.283 * (fx * xx + fy * yy) + ph)
    for _ in range(5):
        cx, cy = torch.rand(B, 1, 1, 1, generator=g), torch.rand(B, 1, 1, 1, generator=g)
        r = 0.05 + 0.2 * torch.rand(B, 1, 1, 1, generator=g)
        a = (torch.rand(B, 1, 1, 1, generator=g) - 0.5) * 0.6
        d = ((xx - cx) ** 2 + (yy - cy) ** 2).sqrt()
        img = img + a * torch.sigmoid((r - d) * 60.0)
    speck = torch.randn(B, 1, size // 2, size // 2, generator=g)
    speck = F.interpolate(speck, size=(size, size), mode="bilinear", align_corners=False) * 0.03
    # full-resolution white speckle: the x4 box filter leaves a quarter of its amplitude in the LR image, the rest is
    # unrecoverable detail -- it caps the reachable PSNR-Y near 33 dB, the published CCA-US x4 operating point (32.72 dB)
    img = (0.45 + img + speck + noise * torch.randn(B, 1, size, size, generator=g)).clamp(0, 1)
    tint = torch.tensor([1.0, 0.97, 0.94]).view(1, 3, 1, 1)
    return (img * tint).clamp(0, 1).to(device)


def smooth_pair(B, lr_size, scale, seed, device="cpu"):
    """(LR, HR) with LR = avg_pool(HR, scale)."""
    import torch.nn.functional as F
    hr = smooth_hr(B, lr_size * scale, seed, device)
    return F.avg_pool2d(hr, scale).contiguous(), hr
