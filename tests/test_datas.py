"""Training input pipeline (datas/us1k.py crop_patch + /255): oracle vs the golden vectors written from the real
reference (CPU), and the device kernel behind m2t_crop_patches vs the oracle, bit for bit (-m gpu)."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import m2trans_oracle as O

SCALE, PATCH, LR_H, LR_W = 4, 48, 37, 53


def _images(n=1):
    return [(O.closed_form_u8_image(LR_H * SCALE, LR_W * SCALE, phase=0.4 + 0.3 * i),
             O.closed_form_u8_image(LR_H, LR_W, phase=0.4 + 0.3 * i)) for i in range(n)]


def test_crop_patch_oracle_reproduces_reference_goldens(golden_dir):
    g = np.load(os.path.join(golden_dir, "crop_patch_seed33.npz"))
    (hr_img, lr_img), = _images(1)
    rng = random.Random(33)
    for k in range(16):
        d = O.crop_patch_draw(rng, LR_H, LR_W, PATCH, SCALE, True)
        assert [int(v) for v in d] == g["draws"][k].tolist()          # same draws in the same order
        a, b = O.crop_patch_apply(lr_img, hr_img, d, PATCH, SCALE)
        assert a.shape == (3, PATCH // SCALE, PATCH // SCALE) and b.shape == (3, PATCH, PATCH)
        assert float(a.double().sum()) == g["lr_sums"][k] and float(b.double().sum()) == g["hr_sums"][k]
        if k == 0:
            assert np.array_equal(a.numpy(), g["lr_first"])
        if k == 5:
            assert np.array_equal(b[:, :8, :8].numpy(), g["hr_corner"])


def test_crop_patch_oracle_geometry():
    (hr_img, lr_img), = _images(1)
    base_l, base_h = O.crop_patch_apply(lr_img, hr_img, (3, 5, False, False, False), PATCH, SCALE)
    assert torch.equal(base_l, torch.from_numpy(lr_img[5:17, 3:15].transpose(2, 0, 1).copy()).float() / 255.0)
    fl, fh = O.crop_patch_apply(lr_img, hr_img, (3, 5, True, True, True), PATCH, SCALE)
    assert torch.equal(fl, base_l.flip(1).flip(2).transpose(1, 2)) and torch.equal(fh, base_h.flip(1).flip(2).transpose(1, 2))


@pytest.mark.gpu
@pytest.mark.parametrize("scale,patch", [(4, 48), (2, 32), (3, 48)])
def test_device_batches_are_bit_identical_to_oracle(scale, patch):
    from m2trans_amd.datas import US1K
    imgs = [(O.closed_form_u8_image((30 + 3 * i) * scale, (41 + 2 * i) * scale, phase=0.2 * i),
             O.closed_form_u8_image(30 + 3 * i, 41 + 2 * i, phase=0.2 * i)) for i in range(5)]
    ds = US1K(scale=scale, patch_size=patch, repeat=3, images=imgs)
    assert len(ds) == 15
    idx = list(range(40))                      # more than one descriptor table (32 per launch), periodic index
    lr, hr = ds.batch(idx, rng=random.Random(7))
    rng = random.Random(7)
    for k, i in enumerate(idx):
        h, l = imgs[i % 5]
        d = O.crop_patch_draw(rng, l.shape[0], l.shape[1], patch, scale, True)
        a, b = O.crop_patch_apply(l, h, d, patch, scale)
        assert torch.equal(lr[k].cpu(), a) and torch.equal(hr[k].cpu(), b), (k, d)


@pytest.mark.gpu
def test_device_loader_epoch_and_errors():
    from m2trans_amd import _lib
    from m2trans_amd.datas import US1K
    imgs = _images(3)
    ds = US1K(scale=SCALE, patch_size=PATCH, repeat=2, images=imgs, augment=False)
    batches = list(ds.loader(4, generator=torch.Generator().manual_seed(1), rng=random.Random(2)))
    assert [b[0].shape[0] for b in batches] == [4, 2]                 # drop_last=False
    assert batches[0][0].shape == (4, 3, 12, 12) and batches[0][1].shape == (4, 3, 48, 48)
    assert float(batches[0][1].min()) >= 0.0 and float(batches[0][1].max()) <= 1.0
    with pytest.raises(_lib.M2TError):
        ds.batch([0], draws=[(LR_W, 0, 0)])                           # corner outside the image
    with pytest.raises(_lib.M2TError):
        US1K(scale=4, patch_size=48, images=imgs, colors=1)
    with pytest.raises(_lib.M2TError):
        US1K(scale=4, patch_size=4 * 64, images=imgs)                 # patch larger than the images


@pytest.mark.gpu
def test_train_step_runs_on_device_batches():
    """The pipeline feeds TrainStep directly: train.py:173-214 with device-resident data."""
    import types
    from m2trans_amd.M2Trans_network import create_model
    from m2trans_amd.datas import US1K
    from m2trans_amd.train_step import TrainStep
    imgs = [(O.closed_form_u8_image(64 * 4, 64 * 4, phase=0.1 * i), O.closed_form_u8_image(64, 64, phase=0.1 * i)) for i in range(2)]
    ds = US1K(scale=4, patch_size=128, repeat=2, images=imgs)
    args = types.SimpleNamespace(n_feats=64, scale=4, rgb_range=1.0, n_blocks=1, colors=3, compute_dtype="bf16")
    torch.manual_seed(3)
    model = create_model(args).cuda()
    ts = TrainStep(model)
    losses = []
    for lr, hr in ds.loader(2, generator=torch.Generator().manual_seed(0), rng=random.Random(0)):
        losses.append(float(ts.step(lr, hr)))
    assert len(losses) == 2 and all(np.isfinite(losses))


def test_benchmark_item_oracle_reproduces_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "benchmark_item.npz"))
    lr_b = O.closed_form_u8_image(21, 35, phase=1.1)
    hr_b = O.closed_form_u8_image(21 * 3 + 2, 35 * 3 + 1, phase=1.1)
    lr, hr = O.benchmark_item(lr_b, hr_b, 3)
    assert tuple(hr.shape) == tuple(g["hr_shape"]) == (3, 63, 105)
    assert float(lr.double().sum()) == float(g["lr_sum"]) and float(hr.double().sum()) == float(g["hr_sum"])
    assert np.array_equal(hr[:, -4:, -4:].numpy(), g["hr_corner"])


@pytest.mark.gpu
def test_device_benchmark_items_bit_identical_and_eval_loop():
    """datas/benchmark.py items on the device, then the whole test.py:77-122 loop (model + PSNR / SSIM) fed by them."""
    import types
    from m2trans_amd import _lib
    from m2trans_amd.M2Trans_network import create_model
    from m2trans_amd.datas import Benchmark
    from m2trans_amd.metrics import evaluate
    scale = 3
    imgs = [(O.closed_form_u8_image(h * scale + dh, w * scale + dw, phase=0.3 * i), O.closed_form_u8_image(h, w, phase=0.3 * i), f"{i}.jpg")
            for i, (h, w, dh, dw) in enumerate([(21, 35, 2, 1), (40, 33, 0, 0), (34, 34, 1, 2)])]
    ds = Benchmark(scale=scale, images=imgs)
    assert len(ds) == 3
    for i, (hr_u8, lr_u8, name) in enumerate(imgs):
        lr, hr, nm = ds[i]
        a, b = O.benchmark_item(lr_u8, hr_u8, scale)
        assert nm == name and torch.equal(lr[0].cpu(), a) and torch.equal(hr[0].cpu(), b)
    args = types.SimpleNamespace(n_feats=64, scale=scale, rgb_range=1.0, n_blocks=1, colors=3, compute_dtype="bf16")
    torch.manual_seed(11)
    model = create_model(args).cuda().eval()
    psnr, ssim = evaluate(model, ((lr, hr) for lr, hr, _ in ds), scale)
    assert np.isfinite(psnr) and 0.0 < ssim <= 1.0
    with pytest.raises(_lib.M2TError):
        Benchmark(scale=scale, images=[(imgs[0][1], imgs[0][0], "swapped.jpg")])      # HR smaller than LR x scale
