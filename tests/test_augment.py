"""cutmix / cut_out of the reference's training loop (train.py:177-181, utils.py:16-108).  Golden vectors: 12 seeded calls
of the real functions (oracle/pin_against_reference.py section 12).  CPU: the oracle restatement and the product's
host-side draws (box tables applied with numpy here) reproduce them bit for bit; GPU: so does the device path."""
import os

import numpy as np
import pytest
import torch

from oracle import augment_oracle as AO

GOLD = os.path.join(os.path.dirname(__file__), "golden", "augment.npz")
SEEDS = list(range(33, 45))


def _case(g, seed):
    n_patch, n_holes, length = (int(v) for v in g[f"args:{seed}"])
    return (torch.from_numpy(g[f"lr:{seed}"]), torch.from_numpy(g[f"hr:{seed}"]), torch.from_numpy(g[f"cm_lr:{seed}"]),
            torch.from_numpy(g[f"cm_hr:{seed}"]), torch.from_numpy(g[f"co:{seed}"]), n_patch, n_holes, length)


def _apply_rows(src, rows, mode, mult):
    out = src.clone()
    for b, boxes in enumerate(rows):
        for (x1, y1, x2, y2, sb) in boxes:           # in order: later boxes overwrite
            ys, xs = slice(y1 * mult, y2 * mult), slice(x1 * mult, x2 * mult)
            out[b, :, ys, xs] = src[sb, :, ys, xs] if mode == 0 else src[b, :, ys, xs] * 0.0
    return out


def test_oracle_reproduces_the_reference_draw_for_draw():
    g = np.load(GOLD)
    for seed in SEEDS:
        lr, hr, cm_lr, cm_hr, co, n_patch, n_holes, length = _case(g, seed)
        AO.seed_all(seed)
        a, b = AO.cutmix(lr, hr, alpha=1.0, n_patch=n_patch, scale=2)
        c = AO.cut_out(lr, n_holes=n_holes, length=length)
        assert torch.equal(a, cm_lr) and torch.equal(b, cm_hr) and torch.equal(c, co), seed


def test_host_side_box_tables_reproduce_the_reference():
    """the product's draws (m2trans_amd/augment.py) consume the generators exactly like utils.py"""
    from m2trans_amd import augment as A
    g = np.load(GOLD)
    for seed in SEEDS:
        lr, hr, cm_lr, cm_hr, co, n_patch, n_holes, length = _case(g, seed)
        AO.seed_all(seed)
        rows = A.cutmix_table(tuple(lr.shape), 1.0, n_patch)
        rows_co = A.cut_out_table(tuple(lr.shape), n_holes, length)
        assert torch.equal(_apply_rows(lr, rows, 0, 1), cm_lr), seed
        assert torch.equal(_apply_rows(hr, rows, 0, 2), cm_hr), seed
        assert torch.equal(_apply_rows(lr, rows_co, 1, 1), co), seed
        t = A.pack_table(rows)
        assert t.shape == (lr.shape[0], 1 + 5 * A.MAX_BOXES) and int(t[:, 0].max()) <= n_patch


def test_no_cpu_fallback():
    from m2trans_amd import augment as A
    from m2trans_amd._lib import M2TError
    with pytest.raises(M2TError):
        A.cut_out(torch.zeros(2, 3, 8, 8), 2, 3)


@pytest.mark.gpu
def test_device_cutmix_and_cut_out_are_bit_identical_to_the_reference():
    from m2trans_amd import augment as A
    g = np.load(GOLD)
    for seed in SEEDS:
        lr, hr, cm_lr, cm_hr, co, n_patch, n_holes, length = _case(g, seed)
        AO.seed_all(seed)
        a, b = A.cutmix(lr.cuda(), hr.cuda(), alpha=1.0, n_patch=n_patch, scale=2)
        c = A.cut_out(lr.cuda(), n_holes=n_holes, length=length)
        assert torch.equal(a.cpu(), cm_lr) and torch.equal(b.cpu(), cm_hr) and torch.equal(c.cpu(), co), seed


def test_halves_follow_torch_chunk():
    from m2trans_amd import augment as A
    for n in range(1, 12):
        want = [t.shape[0] for t in torch.arange(n).chunk(2)] if n > 1 else [1]
        got = A._halves(n)
        assert [l for _, l in got] == want, (n, got, want)
        assert [s for s, _ in got] == [0] + ([want[0]] if len(want) > 1 else []), (n, got)
