#!/usr/bin/env python3
"""bf16 quality probe at full depth (x4, 128x128 LR, 8 blocks) on the GPU -- product code only.

(a) inference: PSNR BETWEEN the bf16-HIP and the fp32-HIP outputs for several weight sets / inputs, and the shift it
    implies at the published 32.72 dB operating point, 10 log10(1 + 10^((32.72 - P)/10))   (target <= 0.01 dB: P >= 59 dB);
(b) training drift in a regime where PSNR means something: smooth synthetic HR, LR = avg_pool(HR); fp32-HIP
    pre-training until the held-out PSNR-Y is >= 25 dB, then N identical steps in fp32 and in bf16 compute from that
    state; |dPSNR-Y| on the held-out set (target <= 0.02 dB).

    python tools/bf16_quality.py [--pretrain 400] [--steps 60] [--json out.json]
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def smooth_hr(B, size, seed, device, noise=0.026):
    """Band-limited synthetic 'tissue': low-frequency Fourier field + a few soft-edged blobs + mild speckle that
    survives the x4 box down-sampling only partly.  Values in [0,1]."""
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


def pair(B, lr_size, scale, seed, device):
    hr = smooth_hr(B, lr_size * scale, seed, device)
    return F.avg_pool2d(hr, scale).contiguous(), hr


def psnr(a, b):
    mse = float((a.double() - b.double()).pow(2).mean())
    return 99.0 if mse == 0 else -10.0 * math.log10(mse)


def implied_shift(p_between, operating=32.72):
    return 10.0 * math.log10(1.0 + 10.0 ** ((operating - p_between) / 10.0))


def make(dtype, scale=4, nb=8, device="cuda"):
    from m2trans_amd.M2Trans_network import create_model
    a = types.SimpleNamespace(n_feats=64, scale=scale, rgb_range=1.0, n_blocks=nb, colors=3, compute_dtype=dtype)
    return create_model(a).to(device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pretrain", type=int, default=800)
    ap.add_argument("--pretrain-lr", type=float, default=5e-4, help="cosine from this to --lr over the pre-training")
    ap.add_argument("--lr", type=float, default=1e-4, help="learning rate of the compared steps (the reference's, train.py:81)")
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--held", type=int, default=8, help="held-out batches of 2 images")
    ap.add_argument("--decay-steps", type=int, default=200, help="second comparison: this many steps with the rate annealed --lr -> 1e-6 (end of training)")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    from m2trans_amd.metrics import psnr_y
    from m2trans_amd.train_step import TrainStep
    dev = torch.device("cuda")
    scale, lr_size = 4, 128
    out = {"inference": [], "drift": {}}

    # ---- (a) inference ----
    torch.manual_seed(33)
    m32 = make("fp32")
    m16 = make("bf16")
    m16.flat_params.copy_(m32.flat_params)
    g = torch.Generator(device="cpu").manual_seed(7)
    inputs = {"uniform": torch.rand(2, 3, lr_size, lr_size, generator=g).to(dev), "smooth": pair(2, lr_size, scale, 11, dev)[0]}
    for wname in ("seed33",):
        for iname, x in inputs.items():
            with torch.no_grad():
                a, b = m32(x), m16(x)
            p = psnr(a, b)
            row = {"weights": wname, "input": iname, "psnr_between_dB": round(p, 2), "implied_shift_dB": round(implied_shift(p), 5),
                   "frac_on_clamp": round(float(((a <= 0) | (a >= 1)).float().mean()), 3)}
            out["inference"].append(row)
            print(row, flush=True)

    # ---- (b) drift: pre-train in fp32 until the held-out PSNR-Y is meaningful ----
    ts = TrainStep(m32, lr=args.pretrain_lr, world_size=1)
    held = [pair(2, lr_size, scale, 900000 + i, dev) for i in range(args.held)]

    def held_psnr(model):
        with torch.no_grad():
            return sum(psnr_y(model(l).cpu(), h.cpu(), scale) for l, h in held) / len(held)

    print("untrained held-out PSNR-Y", round(held_psnr(m32), 3), flush=True)
    for s in range(args.pretrain):
        l, h = pair(args.batch, lr_size, scale, 1000 + s, dev)
        ts.set_lr(args.lr + 0.5 * (args.pretrain_lr - args.lr) * (1.0 + math.cos(math.pi * s / args.pretrain)))
        ts.step(l, h)
        if (s + 1) % 100 == 0:
            print("pretrain", s + 1, "loss", round(float(ts.loss), 5), "held-out PSNR-Y", round(held_psnr(m32), 3), flush=True)
    p0 = held_psnr(m32)
    out["drift"]["pretrained_psnr_y"] = round(p0, 4)
    # trained weights: inference comparison again (the operating point that matters)
    m16.flat_params.copy_(m32.flat_params)
    for i, (l, h) in enumerate(held[:2]):
        with torch.no_grad():
            a, b = m32(l), m16(l)
        p = psnr(a, b)
        row = {"weights": f"pretrained({args.pretrain} steps)", "input": f"held{i}", "psnr_between_dB": round(p, 2),
               "implied_shift_dB": round(implied_shift(p), 5), "psnr_y_fp32": round(psnr_y(a.cpu(), h.cpu(), scale), 4),
               "psnr_y_bf16": round(psnr_y(b.cpu(), h.cpu(), scale), 4)}
        out["inference"].append(row)
        print(row, flush=True)
    # N identical steps from the same state, fp32 vs bf16 compute
    state = (m32.flat_params.clone(), ts.exp_avg.clone(), ts.exp_avg_sq.clone(), ts.step_count)
    def arms(n_steps, decay):
        res = {}
        # "fp32+eps": the fp32 arm again from weights perturbed by 1e-6 relative -- how far two fp32 trajectories drift apart
        # on their own (the noise floor of this comparison)
        for dt, model in (("fp32", m32), ("bf16", m16), ("fp32+eps", m32)):
            model.flat_params.copy_(state[0])
            if dt == "fp32+eps":
                g = torch.Generator(device="cpu").manual_seed(99)
                model.flat_params.mul_(1.0 + 1e-6 * torch.randn(model.flat_params.numel(), generator=g).to(dev))
            t = TrainStep(model, lr=args.lr, world_size=1)
            t.exp_avg.copy_(state[1]); t.exp_avg_sq.copy_(state[2]); t.step_count = state[3]
            for s in range(n_steps):
                if decay:
                    t.set_lr(1e-6 + 0.5 * (args.lr - 1e-6) * (1.0 + math.cos(math.pi * s / n_steps)))
                l, h = pair(args.batch, lr_size, scale, 500000 + s, dev)
                t.step(l, h)
            res[dt] = held_psnr(model)
        tag = f"{n_steps} steps, " + ("lr annealed to 1e-6" if decay else f"lr {args.lr}")
        row = {"schedule": tag, "psnr_y_fp32": round(res["fp32"], 4), "psnr_y_bf16": round(res["bf16"], 4),
               "abs_delta_dB": round(abs(res["fp32"] - res["bf16"]), 4), "noise_floor_dB": round(abs(res["fp32"] - res["fp32+eps"]), 5)}
        print("DRIFT", row, flush=True)
        return row

    out["drift"]["constant_lr"] = arms(args.steps, False)
    out["drift"]["annealed"] = arms(args.decay_steps, True)
    if args.json:
        json.dump(out, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
