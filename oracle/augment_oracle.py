"""oracle/augment_oracle.py -- TEST INFRASTRUCTURE ONLY (see oracle/m2trans_oracle.py for the rules).

CPU restatement of the two on-device augmentations train.py can switch on (train.py:177-181; `cutmix: False` and no
`cutout` key in every shipped config): `utils.cutmix` / `_cutmix` / `rand_bbox` (utils.py:16-71) and `utils.cut_out` /
`_cut_out` (utils.py:74-108).  Random draws come from the same global generators in the same order as the reference
(np.random, torch's CPU generator, `random`), so seeding them identically reproduces the reference bit for bit; pinned
by oracle/pin_against_reference.py section 12 -> tests/golden/augment.npz.
"""
from __future__ import annotations

import random

import numpy as np
import torch


def rand_bbox(size, lam):
    """utils.py:16-33.  NOTE the reference's naming: `W, H = size[2], size[3]`, and the `x` pair later slices dim 3."""
    W, H = size[2], size[3]
    cut_rat = np.power(lam, 1 / 2)
    cut_w = np.int_(W * cut_rat)
    cut_h = np.int_(H * cut_rat)
    cx = np.random.randint(W)
    cy = np.random.randint(H)
    bbx1 = np.clip(cx - cut_w // 2, 0, W)
    bby1 = np.clip(cy - cut_h // 2, 0, H)
    bbx2 = np.clip(cx + cut_w // 2, 0, W)
    bby2 = np.clip(cy + cut_h // 2, 0, H)
    return int(bbx1), int(bby1), int(bbx2), int(bby2)


def _cutmix(data, target, alpha, n_patch, scale):
    """utils.py:36-51: every patch copies from the ORIGINAL tensors under a fresh permutation; later patches overwrite."""
    new_data, new_target = data.clone(), target.clone()
    if np.random.random() < 0.5:
        for _ in range(n_patch):
            indices = torch.randperm(data.size(0))
            lam = np.clip(np.random.beta(alpha, alpha), 0.1, 0.3)
            bbx1, bby1, bbx2, bby2 = rand_bbox(data.size(), lam)
            new_data[:, :, bby1:bby2, bbx1:bbx2] = data[indices, :, bby1:bby2, bbx1:bbx2]
            new_target[:, :, bby1 * scale:bby2 * scale, bbx1 * scale:bbx2 * scale] = \
                target[indices, :, bby1 * scale:bby2 * scale, bbx1 * scale:bbx2 * scale]
    return new_data, new_target


def cutmix(data, target, alpha=1.0, n_patch=1, scale=2):
    """utils.py:54-71: the batch is cut in two halves (torch.chunk) that are mixed independently."""
    if data.size(0) > 1:
        d1, d2 = data.chunk(2, dim=0)
        t1, t2 = target.chunk(2, dim=0)
        d1, t1 = _cutmix(d1, t1, alpha, n_patch, scale)
        d2, t2 = _cutmix(d2, t2, alpha, n_patch, scale)
        return torch.cat([d1, d2], dim=0), torch.cat([t1, t2], dim=0)
    return _cutmix(data, target, alpha, n_patch, scale)


def _cut_out(img, n_holes, length):
    """utils.py:74-92."""
    b, c, h, w = img.size()
    mask = np.ones((h, w), np.float32)
    if random.random() < 0.5:
        for _ in range(n_holes):
            y = np.random.randint(h)
            x = np.random.randint(w)
            y1 = np.clip(y - length // 2, 0, h)
            y2 = np.clip(y + length // 2, 0, h)
            x1 = np.clip(x - length // 2, 0, w)
            x2 = np.clip(x + length // 2, 0, w)
            mask[y1:y2, x1:x2] = 0.
        img = img * torch.from_numpy(mask).expand_as(img).to(img.dtype)
    return img


def cut_out(img, n_holes, length):
    """utils.py:95-108."""
    if img.size(0) > 1:
        i1, i2 = img.chunk(2, dim=0)
        return torch.cat([_cut_out(i1, n_holes, length), _cut_out(i2, n_holes, length)], dim=0)
    return _cut_out(img, n_holes, length)


def seed_all(seed: int):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
