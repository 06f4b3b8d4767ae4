"""CPU: the Swin-T restatement in oracle/swin_oracle.py against transformers.SwinModel (the
third-party implementation of the published algorithm that medclip wraps; version in this image:
5.x, the reference pins 4.24.0 -- same arithmetic, different parameter names)."""
import pytest
import torch

from oracle import swin_oracle as S


def _hf_model_with(params):
    transformers = pytest.importorskip("transformers")
    cfg = transformers.SwinConfig()
    model = transformers.SwinModel(cfg).eval()
    sd = model.state_dict()
    ren = {}
    for k, v in params.items():
        if k == "projection_head.weight":
            continue
        k5 = (k.replace("attention.self.query", "attention.q_proj").replace("attention.self.key", "attention.k_proj")
               .replace("attention.self.value", "attention.v_proj").replace("attention.output.dense", "attention.o_proj")
               .replace("attention.self.relative_position_bias_table", "attention.relative_position_bias.relative_position_bias_table")
               .replace("intermediate.dense", "mlp.fc1").replace("output.dense", "mlp.fc2"))
        cand = [k, k5]
        hit = [c for c in cand if c in sd]
        assert hit, (k, [n for n in sd if "blocks.0" in n][:20])
        ren[hit[0]] = v
    missing = [n for n in sd if n not in ren and "relative_position_index" not in n]
    assert not missing, missing[:10]
    model.load_state_dict(ren, strict=False)
    return model


def test_swin_restatement_matches_transformers_swinmodel():
    p = S.closed_form_swin_params()
    model = _hf_model_with(p)
    g = torch.Generator().manual_seed(7)
    img = torch.rand(2, 3, 224, 224, generator=g)
    with torch.no_grad():
        want = model(pixel_values=img).pooler_output
        got = S.swin_pooled(img, p)
    assert got.shape == (2, 768)
    assert float((got - want).abs().max() / want.abs().max()) < 2e-5


def test_param_count_is_swin_tiny():
    n = sum(int(torch.tensor(s).prod()) for k, s in S.swin_param_shapes().items() if k != "projection_head.weight")
    assert n == 27519354


def test_semantic_loss_value_rng_consumption_and_last_patch():
    """2 (N-1) randint draws per call, x before y; only the last patch matters (losses.py:35-37,67-69)."""
    p = S.closed_form_swin_params()
    g = torch.Generator().manual_seed(3)
    x = torch.rand(3, 256, 240, generator=g)
    y = torch.rand(3, 256, 240, generator=g)
    t = torch.randn(512, generator=g)
    torch.manual_seed(5)
    v = S.semantic_loss_value(x, y, t, p, 3)
    after = torch.randint(1000, ())
    torch.manual_seed(5)
    coords = S.draw_patch_coords(256, 240, 3)
    assert int(torch.randint(1000, ())) == int(after)
    xc, yc = coords[-1]
    xe = S.encode_image(x[None, :, xc:xc + 224, yc:yc + 224], p)
    ye = S.encode_image(y[None, :, xc:xc + 224, yc:yc + 224], p)
    tn = t / t.norm()
    want = (xe @ tn - ye @ tn).abs() / 3.0
    assert v.shape == (1,) and abs(float(v) - float(want)) < 1e-7
