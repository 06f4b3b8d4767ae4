"""Training input pipeline with the image cache resident in HBM.

Host mirror of the reference's `datas/us1k.py` (`US1K`, `crop_patch`) and of the `DataLoader` it is wrapped in
(`datas/utils.py:7-22`): same constructor arguments, same cache directory layout (`<cache>/us1k_hr/rgb/0001.npy`,
`<cache>/us1k_lr_x4/rgb/0001x4.npy`, uint8 HWC as `np.save`d by the reference), same random draws in the same order
(column, row, hflip, vflip, rot90 on Python's `random`), bit-identical float32 tensors.  What differs is where the
work happens: the reference decodes in 8 worker processes, collates on the host and copies every batch over PCIe;
here the whole cache is uploaded once (1 000 image pairs are a few GB of a 288 GB HBM) and one HIP kernel
(`m2t_crop_patches`, k_datas.hip) cuts, flips, transposes, converts and scales a whole batch in place.

Device tensors only: there is no host fallback.  `colors == 1` (Y-channel training through skimage) is not built --
M2Trans trains with `colors: 3` (configs/M2Trans_x4.yml:5)."""
from __future__ import annotations

import ctypes as C
import os
import random
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

FLAG_HFLIP, FLAG_VFLIP, FLAG_ROT90 = 1, 2, 4


def _load_rgb(npy_name: str, png_name: str) -> np.ndarray:
    if os.path.exists(npy_name):
        return np.load(npy_name)
    from PIL import Image                     # imageio.imread(..., pilmode="RGB") is PIL's convert("RGB")
    return np.asarray(Image.open(png_name).convert("RGB"))


class US1K:
    """datas/us1k.py:39-170 for `train=True, colors=3`, with the cache in device memory.

    `images`: optional list of (hr, lr) uint8 HWC arrays used instead of the folders (synthetic data, tests)."""

    def __init__(self, HR_folder: Optional[str] = None, LR_folder: Optional[str] = None, CACHE_folder: Optional[str] = None,
                 train: bool = True, augment: bool = True, scale: int = 2, colors: int = 3, patch_size: int = 96,
                 repeat: int = 168, add_noise: bool = False, cutout: bool = False, device="cuda",
                 images: Optional[Sequence[Tuple[np.ndarray, np.ndarray]]] = None):
        if colors != 3:
            raise _lib.M2TError("US1K (MI355X build): only colors=3 is built (configs/M2Trans_x4.yml:5)")
        if add_noise or cutout:
            raise _lib.M2TError("add_noise / cutout are dead code in the reference (datas/us1k.py:156-167) and are not built")
        if not train:
            raise _lib.M2TError("only the training split is built (the reference validates through datas/benchmark.py)")
        if patch_size % scale:
            raise _lib.M2TError("patch_size must be a multiple of scale")
        self.scale, self.colors, self.patch_size, self.repeat = scale, colors, patch_size, repeat
        self.train, self.augment = train, augment
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.M2TError("US1K (MI355X build) keeps its cache in HBM: a HIP device is required")
        if images is None:
            start, end = (1, 1001) if train else (801, 901)                       # datas/us1k.py:74-79
            hr_dir = os.path.join(CACHE_folder, "us1k_hr", "rgb")
            lr_dir = os.path.join(CACHE_folder, "us1k_lr_x{}".format(scale), "rgb")
            images = []
            for i in range(start, end):
                idx = str(i).zfill(4)
                hr = _load_rgb(os.path.join(hr_dir, idx + ".npy"), os.path.join(HR_folder, idx + ".png"))
                lr = _load_rgb(os.path.join(lr_dir, f"{idx}x{scale}.npy"),
                               os.path.join(LR_folder, f"X{scale}", f"{idx}x{scale}.png"))
                images.append((hr, lr))
        self.nums_trainset = len(images)
        if self.nums_trainset == 0:
            raise _lib.M2TError("empty dataset")
        lp = patch_size // scale
        self._geo: List[Tuple[int, int, int, int, int, int]] = []    # lr_off, hr_off, lr_h, lr_w, hr_h, hr_w
        lr_off = hr_off = 0
        for hr, lr in images:
            for a in (hr, lr):
                if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != colors:
                    raise _lib.M2TError(f"cache entries must be uint8 [H,W,{colors}], got {a.dtype} {a.shape}")
            if train and (lr.shape[0] < lp or lr.shape[1] < lp or hr.shape[0] < lr.shape[0] * scale or hr.shape[1] < lr.shape[1] * scale):
                raise _lib.M2TError(f"image pair {lr.shape} / {hr.shape} too small for patch_size {patch_size} at x{scale}")
            self._geo.append((lr_off, hr_off, lr.shape[0], lr.shape[1], hr.shape[0], hr.shape[1]))
            lr_off += lr.size
            hr_off += hr.size
        self.lr_pool = torch.from_numpy(np.concatenate([np.ascontiguousarray(lr).reshape(-1) for _, lr in images])).to(self.device)
        self.hr_pool = torch.from_numpy(np.concatenate([np.ascontiguousarray(hr).reshape(-1) for hr, _ in images])).to(self.device)

    def __len__(self) -> int:                                                   # datas/us1k.py:140-144
        return self.nums_trainset * self.repeat if self.train else self.nums_trainset

    def draw(self, idx: int, rng=random):
        """The random draws of crop_patch (datas/us1k.py:21,27-29) for item `idx`, in the reference's order."""
        _, _, lr_h, lr_w, _, _ = self._geo[idx % self.nums_trainset]
        lp = self.patch_size // self.scale
        lx = rng.randrange(0, lr_w - lp + 1)
        ly = rng.randrange(0, lr_h - lp + 1)
        flags = 0
        if self.augment:
            flags |= FLAG_HFLIP if rng.random() > 0.5 else 0
            flags |= FLAG_VFLIP if rng.random() > 0.5 else 0
            flags |= FLAG_ROT90 if rng.random() > 0.5 else 0
        return lx, ly, flags

    def batch(self, indices: Sequence[int], rng=random, draws=None):
        """`default_collate([dataset[i] for i in indices])` of the reference: (lr [n,3,p/s,p/s], hr [n,3,p,p]) float32
        in [0,1] on the device.  `draws` (list of (lx, ly, flags)) overrides the random draws."""
        n = len(indices)
        desc = np.zeros((n, 8), dtype=np.int64)
        for k, idx in enumerate(indices):
            lr_off, hr_off, lr_h, lr_w, _, hr_w = self._geo[idx % self.nums_trainset]
            lx, ly, flags = draws[k] if draws is not None else self.draw(idx, rng)
            desc[k] = (lr_off, hr_off, lr_w, hr_w, lx, ly, flags, lr_h)
        lp, hp = self.patch_size // self.scale, self.patch_size
        lr = torch.empty(n, self.colors, lp, lp, dtype=torch.float32, device=self.device)
        hr = torch.empty(n, self.colors, hp, hp, dtype=torch.float32, device=self.device)
        lib = _lib.load()
        with torch.cuda.device(self.device):
            _lib.check(lib.m2t_crop_patches(_lib.ptr(self.lr_pool), _lib.ptr(self.hr_pool), desc.ctypes.data_as(C.c_void_p), n,
                                            self.colors, self.patch_size, self.scale, _lib.ptr(lr), _lib.ptr(hr),
                                            _lib.stream_ptr()), "m2t_crop_patches")
        return lr, hr

    def loader(self, batch_size: int, shuffle: bool = True, generator: Optional[torch.Generator] = None,
               rng=random) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        """One epoch like `DataLoader(us1k, batch_size, shuffle=True, drop_last=False)` (datas/utils.py:22)."""
        order = torch.randperm(len(self), generator=generator).tolist() if shuffle else list(range(len(self)))
        for i in range(0, len(order), batch_size):
            yield self.batch(order[i:i + batch_size], rng)


class Benchmark:
    """datas/benchmark.py:17-72 with the images resident in HBM: `len()`, `ds[i] -> (lr [1,3,h,w], hr [1,3,h*s,w*s],
    name)` float32 in [0,1] on the device (the reference's DataLoader adds the batch dimension of 1), HR cropped to the
    LR size x scale, bit-identical to the reference's tensors.  `images`: optional list of (hr, lr, name) uint8 HWC
    arrays instead of the folders."""

    def __init__(self, HR_folder: Optional[str] = None, LR_folder: Optional[str] = None, scale: int = 2, colors: int = 3,
                 device="cuda", images=None):
        if colors != 3:
            raise _lib.M2TError("Benchmark (MI355X build): only colors=3 is built (configs/M2Trans_x4.yml:5)")
        self.scale, self.colors, self.device = scale, colors, torch.device(device)
        if self.device.type != "cuda":
            raise _lib.M2TError("Benchmark (MI355X build) keeps its images in HBM: a HIP device is required")
        if images is None:
            from PIL import Image                                             # imageio pilmode="RGB" == PIL convert("RGB")
            images = []
            for tag in os.listdir(HR_folder):                                 # datas/benchmark.py:33-43
                ext = ".png" if "US1K_23" in HR_folder else ".jpg"
                lr_name = os.path.join(LR_folder, f"X{scale}", tag.replace(ext, f"x{scale}{ext}"))
                hr = np.asarray(Image.open(os.path.join(HR_folder, tag)).convert("RGB"))
                lr = np.asarray(Image.open(lr_name).convert("RGB"))
                images.append((hr, lr, tag))
        self.img_name = [n for _, _, n in images]
        self._items = []
        for hr, lr, _ in images:
            for a in (hr, lr):
                if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != colors:
                    raise _lib.M2TError(f"images must be uint8 [H,W,{colors}], got {a.dtype} {a.shape}")
            if hr.shape[0] < lr.shape[0] * scale or hr.shape[1] < lr.shape[1] * scale:
                raise _lib.M2TError(f"HR image {hr.shape} is smaller than LR {lr.shape} x {scale}")
            self._items.append((torch.from_numpy(np.ascontiguousarray(hr)).to(self.device), torch.from_numpy(np.ascontiguousarray(lr)).to(self.device)))

    def __len__(self) -> int:
        return len(self._items)

    def __getitem__(self, idx: int):
        hr_u8, lr_u8 = self._items[idx]
        lib = _lib.load()
        lh, lw = lr_u8.shape[0], lr_u8.shape[1]
        lr = torch.empty(1, self.colors, lh, lw, dtype=torch.float32, device=self.device)
        hr = torch.empty(1, self.colors, lh * self.scale, lw * self.scale, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(lib.m2t_image_to_tensor(_lib.ptr(lr_u8), lh, lw, self.colors, lh, lw, _lib.ptr(lr), _lib.stream_ptr()), "m2t_image_to_tensor")
            _lib.check(lib.m2t_image_to_tensor(_lib.ptr(hr_u8), hr_u8.shape[0], hr_u8.shape[1], self.colors, lh * self.scale, lw * self.scale,
                                               _lib.ptr(hr), _lib.stream_ptr()), "m2t_image_to_tensor")
        return lr, hr, self.img_name[idx]

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]
