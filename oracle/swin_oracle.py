"""CPU oracle for the SemanticLoss sub-path (losses.py:18-81)  --  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The reference calls ``medclip.MedCLIPModel(vision_cls=MedCLIPVisionModelViT)``
(losses.py:14-15,22-25,64-69), an un-vendored pip package with no version pin (README.md:42) that
wraps HF ``transformers==4.24.0`` ``microsoft/swin-tiny-patch4-window7-224`` (environment.yml:176);
neither the package nor its weights are under /root/reference.  What IS restated here:

  * the published Swin-T forward (Liu et al. 2021; HF ``modeling_swin.py``): patch projection,
    LayerNorm, (shifted) 7x7 window attention with relative-position bias table and the -100
    shift mask, MLP, patch merging, final LayerNorm + mean pooling  -- validated numerically
    against ``transformers.SwinModel`` (v5.15 in this image) by tests/test_swin_oracle.py;
  * the MedCLIP image head as recalled from the public package: Linear(768, 512, bias=False)
    then L2 normalisation (hypothesis);
  * the value SemanticLoss.__call__ actually returns, including its quirks (losses.py:42-81):
    only the LAST patch's embeddings survive the loop, the patch coordinates are drawn with
    torch.randint on the global CPU RNG (x before y, N-1 pairs per call), the bicubic resize result
    is discarded when N_patches > 1, everything runs under no_grad.

Weights are keyed by the HF 4.24 checkpoint names (``encoder.layers.{s}.blocks.{j}.attention.self.query.weight`` ...).
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

EMBED = 96
DEPTHS = (2, 2, 6, 2)
HEADS = (3, 6, 12, 24)
WIN = 7
IMG = 224
PATCH = 4
PROJ = 512
LN_EPS = 1e-5


def swin_param_shapes() -> "Dict[str, Tuple[int, ...]]":
    s: Dict[str, Tuple[int, ...]] = {}
    s["embeddings.patch_embeddings.projection.weight"] = (EMBED, 3, PATCH, PATCH)
    s["embeddings.patch_embeddings.projection.bias"] = (EMBED,)
    s["embeddings.norm.weight"] = (EMBED,)
    s["embeddings.norm.bias"] = (EMBED,)
    for st, (depth, heads) in enumerate(zip(DEPTHS, HEADS)):
        C = EMBED * 2 ** st
        for j in range(depth):
            p = f"encoder.layers.{st}.blocks.{j}."
            s[p + "layernorm_before.weight"] = (C,)
            s[p + "layernorm_before.bias"] = (C,)
            for nm in ("query", "key", "value"):
                s[p + f"attention.self.{nm}.weight"] = (C, C)
                s[p + f"attention.self.{nm}.bias"] = (C,)
            s[p + "attention.self.relative_position_bias_table"] = ((2 * WIN - 1) ** 2, heads)
            s[p + "attention.output.dense.weight"] = (C, C)
            s[p + "attention.output.dense.bias"] = (C,)
            s[p + "layernorm_after.weight"] = (C,)
            s[p + "layernorm_after.bias"] = (C,)
            s[p + "intermediate.dense.weight"] = (4 * C, C)
            s[p + "intermediate.dense.bias"] = (4 * C,)
            s[p + "output.dense.weight"] = (C, 4 * C)
            s[p + "output.dense.bias"] = (C,)
        if st < 3:
            p = f"encoder.layers.{st}.downsample."
            s[p + "norm.weight"] = (4 * C,)
            s[p + "norm.bias"] = (4 * C,)
            s[p + "reduction.weight"] = (2 * C, 4 * C)
    s["layernorm.weight"] = (8 * EMBED,)
    s["layernorm.bias"] = (8 * EMBED,)
    s["projection_head.weight"] = (PROJ, 8 * EMBED)       # MedCLIP vision projection (hypothesis)
    return s


def closed_form_swin_params(dtype=torch.float32) -> Dict[str, Tensor]:
    """Deterministic non-trivial weights (incl. NON-zero relative-position bias tables: a
    random HF init zero-fills them, which would leave that path untested)."""
    out: Dict[str, Tensor] = {}
    for k, (name, shp) in enumerate(swin_param_shapes().items()):
        n = int(math.prod(shp))
        idx = torch.arange(n, dtype=torch.float64)
        base = torch.sin(idx * (0.7548776662 + 0.0007 * k) + 0.41 * k) + 0.5 * torch.sin(idx * 1.32471795 + 0.9 * k)
        if name.endswith("norm.weight") or "layernorm" in name and name.endswith("weight"):
            v = 1.0 + 0.1 * base
        elif name.endswith("bias"):
            v = 0.05 * base
        elif "relative_position_bias_table" in name:
            v = 0.5 * base
        else:
            fan_in = int(math.prod(shp[1:]))
            v = base / math.sqrt(fan_in)
        out[name] = v.reshape(shp).to(dtype)
    return out


def relative_position_index() -> Tensor:
    """[49,49] index into the (2*7-1)^2 table: (dh + 6) * 13 + (dw + 6), tokens row-major."""
    co = torch.stack(torch.meshgrid(torch.arange(WIN), torch.arange(WIN), indexing="ij")).flatten(1)   # [2,49]
    rel = co[:, :, None] - co[:, None, :]
    return (rel[0] + WIN - 1) * (2 * WIN - 1) + (rel[1] + WIN - 1)


def shift_mask(H: int, W: int, shift: int) -> Tensor:
    """[nW,49,49] additive mask (-100 across cyclic-shift regions)."""
    hr = (torch.arange(H) >= H - WIN).long() + (torch.arange(H) >= H - shift).long()
    wr = (torch.arange(W) >= W - WIN).long() + (torch.arange(W) >= W - shift).long()
    ids = (hr[:, None] * 3 + wr[None, :]).float()
    mw = ids.view(H // WIN, WIN, W // WIN, WIN).permute(0, 2, 1, 3).reshape(-1, WIN * WIN)
    m = mw[:, None, :] - mw[:, :, None]
    return torch.where(m != 0, torch.full_like(m, -100.0), torch.zeros_like(m))


def swin_block(x: Tensor, H: int, W: int, p: Dict[str, Tensor], pre: str, heads: int, shift: int) -> Tensor:
    B, L, C = x.shape
    short = x
    h = F.layer_norm(x, (C,), p[pre + "layernorm_before.weight"], p[pre + "layernorm_before.bias"], LN_EPS).view(B, H, W, C)
    if shift > 0:
        h = torch.roll(h, shifts=(-shift, -shift), dims=(1, 2))
    win = h.view(B, H // WIN, WIN, W // WIN, WIN, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, WIN * WIN, C)
    nW = win.shape[0]
    dh = C // heads

    def proj(nm):
        return F.linear(win, p[pre + f"attention.self.{nm}.weight"], p[pre + f"attention.self.{nm}.bias"]) \
            .view(nW, WIN * WIN, heads, dh).permute(0, 2, 1, 3)
    q, k, v = proj("query"), proj("key"), proj("value")
    att = (q @ k.transpose(-1, -2)) * (dh ** -0.5)
    bias = p[pre + "attention.self.relative_position_bias_table"][relative_position_index().reshape(-1)] \
        .view(WIN * WIN, WIN * WIN, heads).permute(2, 0, 1)
    att = att + bias.unsqueeze(0)
    if shift > 0:
        m = shift_mask(H, W, shift).to(att.dtype)
        att = att.view(B, nW // B, heads, WIN * WIN, WIN * WIN) + m[None, :, None]
        att = att.view(nW, heads, WIN * WIN, WIN * WIN)
    att = torch.softmax(att, dim=-1)
    o = (att @ v).permute(0, 2, 1, 3).reshape(nW, WIN * WIN, C)
    o = F.linear(o, p[pre + "attention.output.dense.weight"], p[pre + "attention.output.dense.bias"])
    o = o.view(B, H // WIN, W // WIN, WIN, WIN, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, C)
    if shift > 0:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    x = short + o.reshape(B, L, C)
    h = F.layer_norm(x, (C,), p[pre + "layernorm_after.weight"], p[pre + "layernorm_after.bias"], LN_EPS)
    h = F.gelu(F.linear(h, p[pre + "intermediate.dense.weight"], p[pre + "intermediate.dense.bias"]))
    h = F.linear(h, p[pre + "output.dense.weight"], p[pre + "output.dense.bias"])
    return x + h


def patch_merge(x: Tensor, H: int, W: int, p: Dict[str, Tensor], pre: str) -> Tensor:
    B, L, C = x.shape
    x = x.view(B, H, W, C)
    x = torch.cat([x[:, r::2, c::2, :] for c in range(2) for r in range(2)], dim=-1).view(B, -1, 4 * C)
    x = F.layer_norm(x, (4 * C,), p[pre + "norm.weight"], p[pre + "norm.bias"], LN_EPS)
    return F.linear(x, p[pre + "reduction.weight"])


def swin_pooled(img: Tensor, p: Dict[str, Tensor], cap=None) -> Tensor:
    """[B,3,224,224] -> pooler_output [B,768] (== transformers.SwinModel(...).pooler_output)."""
    B = img.shape[0]
    x = F.conv2d(img, p["embeddings.patch_embeddings.projection.weight"], p["embeddings.patch_embeddings.projection.bias"],
                 stride=PATCH).flatten(2).transpose(1, 2)
    x = F.layer_norm(x, (EMBED,), p["embeddings.norm.weight"], p["embeddings.norm.bias"], LN_EPS)
    H = W = IMG // PATCH
    for st, (depth, heads) in enumerate(zip(DEPTHS, HEADS)):
        for j in range(depth):
            shift = 0 if (j % 2 == 0 or min(H, W) <= WIN) else WIN // 2
            x = swin_block(x, H, W, p, f"encoder.layers.{st}.blocks.{j}.", heads, shift)
        if cap is not None:
            cap[f"stage{st}"] = x
        if st < 3:
            x = patch_merge(x, H, W, p, f"encoder.layers.{st}.downsample.")
            H, W = H // 2, W // 2
    x = F.layer_norm(x, (8 * EMBED,), p["layernorm.weight"], p["layernorm.bias"], LN_EPS)
    return x.mean(dim=1)


def encode_image(img: Tensor, p: Dict[str, Tensor]) -> Tensor:
    """MedCLIP encode_image (hypothesis): Swin pooled -> Linear(768,512,no bias) -> L2 normalise."""
    e = F.linear(swin_pooled(img, p), p["projection_head.weight"])
    return e / e.norm(dim=-1, keepdim=True)


def draw_patch_coords(hs: int, ws: int, n_patches: int) -> List[Tuple[int, int]]:
    """createNRandompatches (losses.py:29-40): N-1 pairs, x then y, torch.randint on the global RNG;
    `myw` is size(2) (the HEIGHT) and indexes rows -- names kept from the reference."""
    out = []
    for _ in range(n_patches - 1):
        xc = int(torch.randint(hs - IMG, ()))
        yc = int(torch.randint(ws - IMG, ()))
        out.append((xc, yc))
    return out


def semantic_loss_value(x: Tensor, y: Tensor, text_feat: Tensor, p: Dict[str, Tensor], n_patches: int = 3) -> Tensor:
    """Value of SemanticLoss.__call__(x, y, caption) (losses.py:42-81) for [3,Hs,Ws] inputs and a
    given (already computed) text feature [512].  Consumes the global torch RNG like the reference."""
    x, y = x.unsqueeze(0), y.unsqueeze(0)
    px = [F.interpolate(x, mode="bicubic", size=(IMG, IMG), align_corners=True)]
    py = [F.interpolate(y, mode="bicubic", size=(IMG, IMG), align_corners=True)]
    if n_patches > 1:
        for xc, yc in draw_patch_coords(x.shape[2], x.shape[3], n_patches):
            px.append(x[:, :, xc:xc + IMG, yc:yc + IMG])
            py.append(y[:, :, xc:xc + IMG, yc:yc + IMG])
    with torch.no_grad():
        xe = encode_image(px[-1], p)          # only the last patch survives the loop (:67-69)
        ye = encode_image(py[-1], p)
        t = text_feat / text_feat.norm(dim=-1, keepdim=True)
        return ((xe @ t.reshape(-1, 1)).reshape(-1)[0:1] - (ye @ t.reshape(-1, 1)).reshape(-1)[0:1]).abs() / float(n_patches)
