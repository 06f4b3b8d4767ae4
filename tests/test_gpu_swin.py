"""-m gpu: the SemanticLoss path (MedCLIP image tower = Swin-T forward) against the CPU oracle
(oracle/swin_oracle.py, itself validated against transformers.SwinModel).  Parity is UNPINNED with
respect to the real medclip package (not vendored by the reference); what is checked is the published
Swin-T arithmetic + the exact value/RNG semantics of losses.py:42-81."""
import pytest
import torch

from oracle import swin_oracle as S

pytestmark = pytest.mark.gpu


def _loss(dt="fp32", n_patches=3):
    from m2trans_amd.losses import SemanticLoss
    sl = SemanticLoss(criterion="l1", N_patches=n_patches, device="cuda", compute_dtype=dt, max_batch=4)
    p = S.closed_form_swin_params()
    sl.load_image_encoder(p)
    return sl, p


@pytest.mark.parametrize("dt,tol", [("fp32", 2e-4), ("bf16", 3e-2)])
def test_encode_image_matches_oracle(dt, tol):
    sl, p = _loss(dt)
    g = torch.Generator().manual_seed(11)
    src = torch.rand(2, 3, 256, 272, generator=g)
    crops = [(0, 5, 17), (1, 32, 48), (0, 0, 0)]
    want = torch.cat([S.encode_image(src[i:i + 1, :, y:y + 224, x:x + 224], p) for i, y, x in crops])
    got = sl._encoder().encode(src.cuda(), crops).cpu()
    assert got.shape == (3, 512)
    assert float((got.norm(dim=1) - 1).abs().max()) < 1e-4
    assert float((got - want).abs().max() / want.abs().max()) < tol


def test_bicubic_resize_matches_torch():
    from m2trans_amd import _lib
    g = torch.Generator().manual_seed(2)
    x = torch.rand(2, 3, 300, 260, generator=g)
    want = torch.nn.functional.interpolate(x, mode="bicubic", size=(224, 224), align_corners=True)
    xg = x.cuda()
    out = torch.empty(2, 3, 224, 224, device="cuda")
    _lib.check(_lib.load().m2t_bicubic_resize(_lib.ptr(xg), _lib.ptr(out), 6, 300, 260, 224, 224, _lib.stream_ptr()), "bicubic")
    assert float((out.cpu() - want).abs().max()) < 2e-5


@pytest.mark.parametrize("n_patches", [3, 1])
def test_semantic_loss_value_and_rng_order(n_patches):
    """Batched evaluation == the reference's per-sample loop (train.py:203-205): same value per sample,
    same consumption of the global torch CPU RNG (2(N-1) randint draws per sample, x before y)."""
    sl, p = _loss("fp32", n_patches)
    g = torch.Generator().manual_seed(4)
    B = 3
    sr = torch.rand(B, 3, 256, 240, generator=g)
    hr = torch.rand(B, 3, 256, 240, generator=g)
    caps = ["thyroid nodule, transverse", "carotid artery long axis", "liver segment"]
    table = {c: torch.randn(512, generator=g) for c in caps}
    sl.set_text_features(table)
    torch.manual_seed(33)
    want = [S.semantic_loss_value(sr[i], hr[i], table[caps[i]], p, n_patches) for i in range(B)]
    after_ref = int(torch.randint(10 ** 6, ()))
    torch.manual_seed(33)
    tot = sl.batch(sr.cuda(), hr.cuda(), caps)
    after = int(torch.randint(10 ** 6, ()))
    assert after == after_ref
    per = sl.last_per_sample.cpu()
    for i in range(B):
        assert abs(float(per[i]) - float(want[i])) < 2e-5, (i, float(per[i]), float(want[i]))
    assert abs(float(tot) - float(sum(want))) < 5e-5
    # single-sample call surface (train.py:205)
    torch.manual_seed(33)
    one = sl(sr[0].cuda(), hr[0].cuda(), caps[0])
    assert one.shape == (1,) and abs(float(one) - float(want[0])) < 2e-5


def test_semantic_loss_requires_weights_and_gpu():
    from m2trans_amd._lib import M2TError
    from m2trans_amd.losses import SemanticLoss
    sl = SemanticLoss(device="cuda")
    with pytest.raises(M2TError):
        sl.batch(torch.zeros(1, 3, 256, 256, device="cuda"), torch.zeros(1, 3, 256, 256, device="cuda"), ["x"])
    with pytest.raises(M2TError):
        SemanticLoss(device="cpu").batch(torch.zeros(1, 3, 256, 256), torch.zeros(1, 3, 256, 256), ["x"])


def test_train_step_with_semantic_loss_term_is_constant_offset():
    """config 3 semantics: the regulariser changes the reported loss by lambda_clip * sum_i value_i
    and leaves the gradients untouched."""
    from m2trans_amd.train_step import TrainStep
    from oracle import m2trans_oracle as O
    from tests.gpu_util import build_model
    sl, p = _loss("fp32", 3)
    scale, nb, B, H, W = 4, 1, 2, 64, 64
    model, _ = build_model(scale, nb, "fp32")
    x = O.closed_form_image(B, 3, H, W).cuda()
    hr = O.closed_form_image(B, 3, H * scale, W * scale, phase=0.7).cuda()
    ts0 = TrainStep(model, world_size=1)
    l0 = float(ts0.forward_backward(x, hr))
    g0 = ts0.grads.clone()
    ts1 = TrainStep(model, world_size=1, semantic_loss=sl, lambda_clip=0.01)
    # a caption without an injected text feature must raise, not silently get a stand-in embedding
    from m2trans_amd._lib import M2TError
    with pytest.raises(M2TError, match="no text feature"):
        sl.batch(torch.rand(2, 3, 256, 256, device="cuda"), torch.rand(2, 3, 256, 256, device="cuda"), ["a", "b"])
    g = torch.Generator().manual_seed(8)
    sl.set_text_features({"a": torch.randn(512, generator=g), "b": torch.randn(512, generator=g)})
    torch.manual_seed(1)
    l1 = float(ts1.forward_backward(x, hr, ["a", "b"]))
    assert torch.equal(ts1.grads, g0)
    assert abs((l1 - l0) - 0.01 * float(sl.last_per_sample.sum())) < 1e-6
    assert float(sl.last_per_sample.min()) >= 0.0
    # the encoder beside the backward pass (default) and after it give the same values, bit for bit
    per_overlapped = sl.last_per_sample.clone()
    ts2 = TrainStep(model, world_size=1, semantic_loss=sl, lambda_clip=0.01, overlap_semantic=False)
    torch.manual_seed(1)
    l2 = float(ts2.forward_backward(x, hr, ["a", "b"]))
    assert l2 == l1 and torch.equal(sl.last_per_sample, per_overlapped) and torch.equal(ts2.grads, g0)
