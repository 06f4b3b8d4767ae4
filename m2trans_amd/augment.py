"""On-device batch augmentations of the reference's training loop (train.py:177-181): ``cutmix`` (utils.py:16-71) and
``cut_out`` (utils.py:74-108), same signatures and return values.  The random draws are made on the host from the same
global generators in the same order as the reference (``np.random``, torch's CPU generator, ``random``) and turned into
a small box table; one kernel (``m2t_box_mix``, csrc/k_datas.hip) applies it to the device batch -- no per-patch slicing
kernels, no host round trip of the images.  Bit-identical to the reference for identical generator states.
Both switches are off in every shipped config (``cutmix: False``, no ``cutout`` key)."""
from __future__ import annotations

import random
from typing import List, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import M2TError

MAX_BOXES = 16


def _halves(n: int) -> List[Tuple[int, int]]:
    """(start, length) of ``torch.chunk(2, dim=0)`` for a batch of n > 1, or the whole batch."""
    if n <= 1:
        return [(0, n)]
    first = (n + 1) // 2
    return [(0, first), (first, n - first)] if n - first > 0 else [(0, first)]


def _rand_bbox(size, lam):
    """utils.py:16-33 (the reference names size[2] `W`; its `x` pair later slices dim 3 -- kept as is)."""
    W, H = size[2], size[3]
    cut_rat = np.power(lam, 1 / 2)
    cut_w = np.int_(W * cut_rat)
    cut_h = np.int_(H * cut_rat)
    cx = np.random.randint(W)
    cy = np.random.randint(H)
    return (int(np.clip(cx - cut_w // 2, 0, W)), int(np.clip(cy - cut_h // 2, 0, H)),
            int(np.clip(cx + cut_w // 2, 0, W)), int(np.clip(cy + cut_h // 2, 0, H)))


def cutmix_table(shape, alpha=1.0, n_patch=1) -> List[List[Tuple[int, int, int, int, int]]]:
    """Draws of utils.cutmix for a data tensor of `shape` -> per sample the list of (x1, y1, x2, y2, source sample)."""
    B = shape[0]
    rows: List[List[Tuple[int, int, int, int, int]]] = [[] for _ in range(B)]
    for start, n in _halves(B):
        if np.random.random() < 0.5:                                   # utils.py:40
            for _ in range(n_patch):
                indices = torch.randperm(n)                            # :42
                lam = np.clip(np.random.beta(alpha, alpha), 0.1, 0.3)  # :44
                x1, y1, x2, y2 = _rand_bbox((n,) + tuple(shape[1:]), lam)
                for i in range(n):
                    rows[start + i].append((x1, y1, x2, y2, start + int(indices[i])))
    return rows


def cut_out_table(shape, n_holes, length) -> List[List[Tuple[int, int, int, int, int]]]:
    """Draws of utils.cut_out -> per sample the list of holes (x1, y1, x2, y2, 0)."""
    B, _, h, w = shape
    rows: List[List[Tuple[int, int, int, int, int]]] = [[] for _ in range(B)]
    for start, n in _halves(B):
        if random.random() < 0.5:                                      # utils.py:78
            holes = []
            for _ in range(n_holes):
                y = np.random.randint(h)
                x = np.random.randint(w)
                holes.append((int(np.clip(x - length // 2, 0, w)), int(np.clip(y - length // 2, 0, h)),
                              int(np.clip(x + length // 2, 0, w)), int(np.clip(y + length // 2, 0, h)), 0))
            for i in range(n):
                rows[start + i] = list(holes)
    return rows


def pack_table(rows, max_boxes: int = MAX_BOXES) -> torch.Tensor:
    t = torch.zeros(len(rows), 1 + 5 * max_boxes, dtype=torch.int32)
    for b, r in enumerate(rows):
        if len(r) > max_boxes:
            raise M2TError(f"augment: {len(r)} boxes for one sample (max {max_boxes})")
        t[b, 0] = len(r)
        for i, q in enumerate(r):
            t[b, 1 + 5 * i: 6 + 5 * i] = torch.tensor(q, dtype=torch.int32)
    return t


def _apply(src: torch.Tensor, table: torch.Tensor, mode: int, mult: int) -> torch.Tensor:
    if src.device.type != "cuda":
        raise M2TError("augment (MI355X build): needs a HIP device tensor; there is no CPU fallback")
    src = src.contiguous().float()
    dst = torch.empty_like(src)
    B, C, H, W = src.shape
    with torch.cuda.device(src.device):
        _lib.check(_lib.load().m2t_box_mix(_lib.ptr(src), _lib.ptr(dst), B, C, H, W, _lib.ptr(table), MAX_BOXES, mode, mult,
                                           _lib.stream_ptr()), "m2t_box_mix")
    return dst


def cutmix(data: torch.Tensor, target: torch.Tensor, alpha=1.0, n_patch=1, scale=2):
    """utils.cutmix(data, target, alpha, n_patch, scale) -> (new_data, new_target)."""
    rows = cutmix_table(tuple(data.shape), alpha, n_patch)
    table = pack_table(rows).to(data.device)
    return _apply(data, table, 0, 1), _apply(target, table, 0, int(scale))


def cut_out(img: torch.Tensor, n_holes, length):
    """utils.cut_out(img, n_holes, length) -> img."""
    rows = cut_out_table(tuple(img.shape), n_holes, length)
    table = pack_table(rows).to(img.device)
    return _apply(img, table, 1, 1)
