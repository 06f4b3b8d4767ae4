"""CPU oracle for the M2Trans training-step hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (fp32/fp64, CPU) *restatement* of the arithmetic of the
reference's hot path.  It is the checker for the HIP kernels, never the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.  The product package ``m2trans_amd`` never imports anything from ``oracle/``.

Pinning status: PINNED.  ``oracle/pin_against_reference.py`` imports the real reference
(``/root/reference/models/M2Trans_network.py``, only available in the build container)
and checks this restatement against it (forward, all parameter gradients, Adam step) and
writes the golden fixtures under ``tests/golden/`` that ``tests/test_oracle_golden.py``
re-checks without the reference.  The MedCLIP sub-path (``semantic_loss_value``) is
"parity unpinned": the `medclip` package is not vendored in the reference (see
``oracle/swin_oracle.py``).

Every function cites the reference lines it restates (paths relative to the reference
root).  The formulation is deliberately different from the reference's module code: it is
functional, keyed by ``state_dict`` names, and builds the halo windows by explicit gather
so that the zero-pad + relative-position semantics are spelled out.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]

BLOCK = 8      # query window edge           (models/M2Trans_network.py:119-122, block_size=8)
HALO = 1       # halo on each side           (models/M2Trans_network.py:119-122, halo_size=1)
KWIN = BLOCK + 2 * HALO  # 10 -> 100 keys per window
PAD_MULTIPLE = 32        # lcm(8,16,32)     (models/M2Trans_network.py:78-86)


# --------------------------------------------------------------------------------------
# parameter inventory  (models/M2Trans_network.py:17-56, 114-130, 267-288, 370-379)
# --------------------------------------------------------------------------------------
def branch_channels(n_feats: int) -> Tuple[int, int, int, int]:
    """Channels of the four TBlocks of one CFTM (models/M2Trans_network.py:119-122)."""
    return (n_feats // 4, n_feats, n_feats * 4, n_feats * 4)


def param_shapes(n_feats: int = 64, scale: int = 4, n_blocks: int = 8, colors: int = 3
                 ) -> "Dict[str, Tuple[int, ...]]":
    """state_dict names -> shapes, in the reference's registration order (123 entries at
    the shipped config: 4 frozen MeanShift tensors + 119 trainable)."""
    s: Dict[str, Tuple[int, ...]] = {}
    s["sub_mean.weight"] = (3, 3, 1, 1)
    s["sub_mean.bias"] = (3,)
    s["add_mean.weight"] = (3, 3, 1, 1)
    s["add_mean.bias"] = (3,)
    s["head.weight"] = (n_feats, colors, 3, 3)
    s["head.bias"] = (n_feats,)
    for b in range(n_blocks):
        for i, c in enumerate(branch_channels(n_feats), start=1):
            s[f"body.{b}.attn{i}.rel_h"] = (1, KWIN, 1, c // 2)
            s[f"body.{b}.attn{i}.rel_w"] = (1, 1, KWIN, c // 2)
            s[f"body.{b}.attn{i}.qkv_conv.weight"] = (3 * c, c, 1, 1)
        s[f"body.{b}.feed_forward.0.weight"] = (n_feats, n_feats, 3, 3)
        s[f"body.{b}.feed_forward.0.bias"] = (n_feats,)
    if scale == 4:
        s["tail.0.weight"] = (4 * n_feats, n_feats, 1, 1)
        s["tail.0.bias"] = (4 * n_feats,)
        s["tail.3.weight"] = (4 * n_feats, n_feats, 1, 1)
        s["tail.3.bias"] = (4 * n_feats,)
        s["tail.6.weight"] = (3, n_feats, 3, 3)
    else:
        s["tail.0.weight"] = (n_feats * scale * scale, n_feats, 1, 1)
        s["tail.0.bias"] = (n_feats * scale * scale,)
        s["tail.3.weight"] = (3, n_feats, 3, 3)
    return s


FROZEN = ("sub_mean.weight", "sub_mean.bias", "add_mean.weight", "add_mean.bias")


def closed_form_params(n_feats=64, scale=4, n_blocks=8, colors=3, rgb_range=1.0,
                       dtype=torch.float32, gain: float = 1.0) -> Params:
    """Deterministic, RNG-free weights so every box regenerates the same tensors.

    Magnitudes follow the reference's init distributions (conv: kaiming-uniform bound
    1/sqrt(fan_in); qkv: kaiming-normal fan_out std sqrt(2/3C); rel-pos ~ N(0,1),
    models/M2Trans_network.py:34,342-345) but values come from a sine sequence keyed by
    the parameter's position in the inventory.
    """
    out: Params = {}
    for k, (name, shp) in enumerate(param_shapes(n_feats, scale, n_blocks, colors).items()):
        n = int(math.prod(shp))
        idx = torch.arange(n, dtype=torch.float64)
        base = torch.sin(idx * (0.61803398875 + 0.001 * k) + 0.37 * k) \
            + 0.5 * torch.sin(idx * 1.7320508 + 1.1 * k)
        if name in FROZEN:
            if name.endswith("weight"):
                v = torch.eye(3, dtype=torch.float64).reshape(3, 3, 1, 1)
            else:
                sign = -1.0 if name.startswith("sub") else 1.0
                v = sign * rgb_range * torch.tensor([0.4488, 0.4371, 0.4040], dtype=torch.float64)
            out[name] = v.to(dtype)
            continue
        if "rel_" in name:
            amp = 0.8
        elif "qkv_conv" in name:
            amp = math.sqrt(2.0 / shp[0]) * 1.2
        elif name.endswith("bias"):
            amp = 0.05
        else:
            fan_in = shp[1] * shp[2] * shp[3]
            amp = 1.0 / math.sqrt(fan_in) * 1.1
        out[name] = (gain * amp * base).reshape(shp).to(dtype)
    return out


def closed_form_image(b: int, c: int, h: int, w: int, phase: float = 0.0,
                      dtype=torch.float32) -> Tensor:
    """Deterministic image in [0,1] (the reference feeds float32 NCHW in [0,1],
    datas/us1k.py:169)."""
    yy = torch.arange(h, dtype=torch.float64).view(1, 1, h, 1)
    xx = torch.arange(w, dtype=torch.float64).view(1, 1, 1, w)
    cc = torch.arange(c, dtype=torch.float64).view(1, c, 1, 1)
    bb = torch.arange(b, dtype=torch.float64).view(b, 1, 1, 1)
    v = 0.5 + 0.25 * torch.sin(0.37 * yy + 0.11 * xx * (1 + cc) + 0.9 * bb + phase) \
        + 0.25 * torch.sin(0.23 * xx - 0.31 * yy + 1.3 * cc + 0.5 * bb * yy / h + 2 * phase)
    return v.clamp(0, 1).to(dtype)


# --------------------------------------------------------------------------------------
# bf16 emulation (test infrastructure for the bf16 throughput mode of the HIP path)
#
# The reference is fp32 end to end; the HIP path's bf16 mode keeps fp32 accumulation, statistics,
# softmax, loss and master weights but STORES activations and packed weights in bf16.  With ``emulate_bf16=True``
# the restatement below rounds to bf16 at exactly those storage points (listed in each function).
#
# Measured on MI355X: the network amplifies a bf16-sized perturbation by ~2x per CFTM block (closed-form weights;
# InstanceNorm + softmax), so after 3-4 blocks ANY two bf16 evaluations -- the HIP path and this emulation, which
# differ only in fp32 summation order and in exp / erf implementations -- are as far apart as bf16 is from fp32
# (X1: 1e-3 rel-rms, X2: 6e-3, X4: 8e-2).  An end-to-end bound therefore cannot be tight at full depth.  What CAN
# be tight is the per-kernel error: ``force`` (name -> tensor read back from the HIP workspace) replaces each stored
# tensor by the HIP path's own value right after this restatement has computed it from the HIP path's own inputs
# ("teacher forcing"), and ``stage_report`` records how far the two were apart -- one kernel's worth of error per
# entry.  The gradient then flows through THIS arithmetic along the HIP path's forward trajectory, un-rounded
# (straight through every rounding), so comparing it with the HIP gradients measures the bf16 storage noise of the
# HIP backward alone.  None of this is reference arithmetic: the pinned path is emulate_bf16=False.
# --------------------------------------------------------------------------------------
class _RoundFwd(torch.autograd.Function):
    """value -> bf16 -> fp32; the gradient passes straight through (fp32)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


class _Emu:
    def __init__(self, force=None, report=None):
        self.force, self.report = force, report


def _r(x: Tensor, e: "_Emu | None", name: str | None = None, sl=None) -> Tensor:
    """a bf16 storage point.  `name` (+ optional channel slice `sl`) identifies the HIP workspace tensor for forcing."""
    if e is None:
        return x
    y = _RoundFwd.apply(x)
    if e.force is not None and name is not None and name in e.force:
        f = e.force[name]
        if sl is not None:
            f = f[:, sl]
        f = f.to(y.dtype)
        if e.report is not None:
            d = (y.detach() - f).double()
            key = name if sl is None else f"{name}[{sl.start}:{sl.stop}]"
            e.report[key] = (float(d.abs().max() / (f.double().abs().max() + 1e-30)),
                             float(d.pow(2).mean().sqrt() / (f.double().pow(2).mean().sqrt() + 1e-30)))
        y = y + (f - y).detach()
    return y


def _rw(w: Tensor, e: "_Emu | None") -> Tensor:
    return _RoundFwd.apply(w) if e is not None else w


GELU_FAST = (1.59484566, 7.40076173e-2, -6.95025211e-4)     # csrc/m2t_common.h M2T_GELU_A / _B / _C


def gelu_fast_both(t: Tensor):
    """bf16 mode's GELU and GELU' (csrc/m2t_common.h gelu_fast_both): Phi(t) ~ sigma(t (a + b t^2 + c t^4)), polynomial at
    t clamped to [-8, 8].  NOT reference arithmetic (the reference uses the exact erf GELU, models/M2Trans_network.py:44,47)."""
    a, b, c = GELU_FAST
    tc = t.clamp(-8.0, 8.0)
    t2 = tc * tc
    u = (c * t2 + b) * t2 + a
    ex = torch.exp2(-1.4426950408889634 * (tc * u))
    s = 1.0 / (1.0 + ex)
    act = t * s
    du = (5.0 * c * t2 + 3.0 * b) * t2 + a
    return act, act * (ex * s) * du + s


def _gelu_stored(t: Tensor, e: "_Emu | None", act_name: str, der_name: str):
    """tail expansion of the HIP path: stores gelu(t) and gelu'(t) in bf16; its backward multiplies by the STORED
    derivative (csrc/k_gemm.hip tail_expand_kernel, k_tail_bwd.hip).  Returns (activation, stored derivative)."""
    if e is None:
        return F.gelu(t), gelu_derivative(t)
    # bf16 mode evaluates GELU / GELU' by the sigmoid-quintic approximation of csrc/m2t_common.h (gelu_fast_both: |error| <=
    # 3.8e-5 / 9.3e-5 against the exact erf form, tests/test_host_cpu.py) -- the same formula here, so that the teacher-forced
    # stage gates measure the kernels and not the approximation
    fa, fd = gelu_fast_both(t.detach())
    der = _r(fd, e, der_name)
    act_v = _r(fa, e, act_name)
    # value = the stored activation, d(value)/dt = the stored derivative
    act = act_v + (t - t.detach()) * der
    return act, der


# --------------------------------------------------------------------------------------
# Haar DWT / IWT   (models/M2Trans_network.py:198-237)
# --------------------------------------------------------------------------------------
def dwt(x: Tensor) -> Tensor:
    """[B,C,H,W] -> [B,4C,H/2,W/2], band-major (LL,HL,LH,HH) on channels
    (models/M2Trans_network.py:203-209).  a,b,c,d = (even r, even c), (odd r, even c),
    (even r, odd c), (odd r, odd c)."""
    a = x[:, :, 0::2, 0::2]
    b = x[:, :, 1::2, 0::2]
    c = x[:, :, 0::2, 1::2]
    d = x[:, :, 1::2, 1::2]
    ll = 0.5 * (a + b + c + d)
    hl = 0.5 * (-a - b + c + d)
    lh = 0.5 * (-a + b - c + d)
    hh = 0.5 * (a - b - c + d)
    return torch.cat((ll, hl, lh, hh), dim=1)


def iwt(x: Tensor) -> Tensor:
    """[B,4C,h,w] -> [B,C,2h,2w]  (models/M2Trans_network.py:219-234)."""
    B, C4, h, w = x.shape
    C = C4 // 4
    ll, hl, lh, hh = x[:, 0:C], x[:, C:2 * C], x[:, 2 * C:3 * C], x[:, 3 * C:4 * C]
    out = x.new_zeros(B, C, 2 * h, 2 * w)
    out[:, :, 0::2, 0::2] = 0.5 * (ll - hl - lh + hh)
    out[:, :, 1::2, 0::2] = 0.5 * (ll - hl + lh - hh)
    out[:, :, 0::2, 1::2] = 0.5 * (ll + hl - lh - hh)
    out[:, :, 1::2, 1::2] = 0.5 * (ll + hl + lh + hh)
    return out


# --------------------------------------------------------------------------------------
# halo window attention   (models/M2Trans_network.py:290-340 with sr=1, heads=1)
# --------------------------------------------------------------------------------------
def window_attention_core(q: Tensor, k: Tensor, v: Tensor, rel_h: Tensor, rel_w: Tensor,
                          emu: "_Emu | None" = None) -> Tensor:
    """q,k,v: [B,C,h,w] (already projected).  Returns [B,C,h,w].

    Restates models/M2Trans_network.py:310-332: 8x8 query windows, 10x10 key windows cut
    from the ZERO-padded k/v planes (F.unfold padding=1), relative-position embedding
    added to *every* one of the 100 keys -- including the zero-padded phantom ones, which
    therefore take softmax mass with value 0 -- first C/2 channels get the
    row embedding rel_h[r], last C/2 the column embedding rel_w[c] (:322-325).
    """
    B, C, h, w = q.shape
    nh, nw = h // BLOCK, w // BLOCK
    scale = float(C) ** -0.5                               # :311 (head_ch = C, heads = 1)
    emulate_bf16 = emu is not None
    qw = q.view(B, C, nh, BLOCK, nw, BLOCK).permute(0, 2, 4, 3, 5, 1)   # B nh nw 8 8 C
    qw = qw.reshape(B * nh * nw, BLOCK * BLOCK, C)
    if not emulate_bf16:
        qw = qw * scale                                    # (bf16 mode: the fp32 scores are scaled instead)
    kp = F.pad(k, (HALO, HALO, HALO, HALO))                # zero padding (:313 padding=halo)
    vp = F.pad(v, (HALO, HALO, HALO, HALO))
    # explicit gather of the 10x10 neighbourhoods: window (i,j) covers padded rows
    # 8i..8i+9, cols 8j..8j+9
    kw = kp.unfold(2, KWIN, BLOCK).unfold(3, KWIN, BLOCK)  # B C nh nw 10 10
    vw = vp.unfold(2, KWIN, BLOCK).unfold(3, KWIN, BLOCK)
    kw = kw.permute(0, 2, 3, 4, 5, 1).reshape(B * nh * nw, KWIN, KWIN, C)
    vw = vw.permute(0, 2, 3, 4, 5, 1).reshape(B * nh * nw, KWIN * KWIN, C)
    half = C // 2
    bias = torch.cat((rel_h.reshape(1, KWIN, 1, half).expand(1, KWIN, KWIN, half),
                      rel_w.reshape(1, 1, KWIN, half).expand(1, KWIN, KWIN, half)), dim=-1)
    kw = _r((kw + bias).reshape(B * nh * nw, KWIN * KWIN, C), emu)   # bf16 mode: K^ = bf16(k + rel) in LDS
    sim = torch.bmm(qw, kw.transpose(1, 2))                # :328
    if emulate_bf16:
        sim = sim * scale
    attn = _r(torch.softmax(sim, dim=-1), emu)             # :329   (bf16 mode: P is the bf16 operand of P V)
    out = torch.bmm(attn, vw)                              # :331  [BL,64,C]
    out = out.view(B, nh, nw, BLOCK, BLOCK, C).permute(0, 5, 1, 3, 2, 4)
    return out.reshape(B, C, h, w)                         # :332


def tblock(x: Tensor, p: Params, prefix: str, cap=None, cap_key: str = "", emu: "_Emu | None" = None) -> Tensor:
    """TBlock.forward (models/M2Trans_network.py:290-340), sr=1, no pad branch.
    ``cap`` (optional dict) records intermediates for the kernel-level parity tests.
    bf16 storage points: the packed qkv weight, the qkv tensor, K^ and P (window_attention_core)."""
    wq = _rw(p[prefix + "qkv_conv.weight"], emu)
    qkv = _r(F.conv2d(x, wq), emu, cap_key)                # :307
    if cap is not None:
        cap[cap_key] = qkv
    q, k, v = torch.chunk(qkv, 3, dim=1)                   # :308
    return window_attention_core(q, k, v, p[prefix + "rel_h"], p[prefix + "rel_w"], emu)


# --------------------------------------------------------------------------------------
# CFTM and the full network   (models/M2Trans_network.py:58-86, 132-164)
# --------------------------------------------------------------------------------------
def instance_norm(x: Tensor, eps: float = 1e-5) -> Tensor:
    """nn.InstanceNorm2d(nf): per-(b,c) biased variance, eps 1e-5, no affine
    (models/M2Trans_network.py:127)."""
    mu = x.mean(dim=(2, 3), keepdim=True)
    var = x.var(dim=(2, 3), unbiased=False, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps)


def cftm(x: Tensor, p: Params, prefix: str, cap=None, ck: str = "", emu: "_Emu | None" = None,
         extra_residual: Tensor | None = None, out_name: str | None = None) -> Tensor:
    """CFTM.forward, norm branch (models/M2Trans_network.py:132-164).
    bf16 storage points (csrc/k_pointwise.hip branch_prep, k_attn*.hip epilogue, k_conv.hip): each branch input
    xin = (norm chunk + previous branch) / 2, its transform d = DWT^L(xin), each branch output x_k' (a chunk of xc),
    the packed 3x3 weight and the block output.  ``extra_residual``: the `res + x` of M2Trans.forward:70, which the
    HIP path folds into the LAST block's conv epilogue (one rounding instead of two)."""
    e = emu
    x1, x2, x3, x4 = torch.chunk(instance_norm(x), 4, dim=1)
    d1 = _r(x1, e, ck + "d1")
    x1 = _r(tblock(d1, p, prefix + "attn1.", cap, ck + "qkv1", e) + d1, e, ck + "xc", slice(0, 16))
    x2 = _r((x2 + x1) / 2.0, e)
    d2 = _r(dwt(x2), e, ck + "d2")
    x2 = _r(iwt(tblock(d2, p, prefix + "attn2.", cap, ck + "qkv2", e)) + x2, e, ck + "xc", slice(16, 32))
    x3 = _r((x3 + x2) / 2.0, e)
    d3 = _r(dwt(dwt(x3)), e, ck + "d3")
    x3 = _r(iwt(iwt(tblock(d3, p, prefix + "attn3.", cap, ck + "qkv3", e))) + x3, e, ck + "xc", slice(32, 48))
    x4 = _r((x4 + x3) / 2.0, e)
    d4 = _r(dwt(dwt(x4)), e, ck + "d4")
    x4 = _r(iwt(iwt(tblock(d4, p, prefix + "attn4.", cap, ck + "qkv4", e))) + x4, e, ck + "xc", slice(48, 64))
    xc = torch.cat((x1, x2, x3, x4), dim=1)
    if cap is not None:
        cap[ck + "d1"], cap[ck + "d2"], cap[ck + "d3"], cap[ck + "d4"], cap[ck + "xc"] = d1, d2, d3, d4, xc
    y = F.conv2d(xc, _rw(p[prefix + "feed_forward.0.weight"], e), p[prefix + "feed_forward.0.bias"],
                 padding=1) + x                            # zero padding (:124-126,164)
    if extra_residual is not None:
        y = y + extra_residual
    return _r(y, e, out_name)


def pad_to_multiple(x: Tensor, m: int = PAD_MULTIPLE) -> Tensor:
    """check_image_size (models/M2Trans_network.py:78-86): reflect pad right/bottom."""
    h, w = x.shape[-2:]
    ph, pw = (m - h % m) % m, (m - w % m) % m
    if ph == 0 and pw == 0:
        return x
    return F.pad(x, (0, pw, 0, ph), mode="reflect")


def conv3x3_reflect(x: Tensor, w: Tensor, b=None) -> Tensor:
    return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w, b)


def gelu_derivative(t: Tensor) -> Tensor:
    """d/dt of the exact (erf) GELU that nn.GELU() applies (models/M2Trans_network.py:44,46,53)."""
    return 0.5 * (1.0 + torch.erf(t * 0.7071067811865476)) + t * torch.exp(-0.5 * t * t) * 0.3989422804014327


def tail(x: Tensor, p: Params, scale: int, cap=None, emu: "_Emu | None" = None) -> Tensor:
    """models/M2Trans_network.py:41-56.  The HIP path stores gelu(t) and gelu'(t) of each expansion
    (workspace tensors t1act/t1der, t2act/t2der) instead of the pre-activation t; bf16 storage points: those four
    tensors and the packed tail weights (the last conv's weight is rounded when it is staged, k_conv.hip)."""
    e = emu
    if scale == 4:
        t1 = F.pixel_shuffle(F.conv2d(x, _rw(p["tail.0.weight"], e), p["tail.0.bias"]), 2)
        a1, d1 = _gelu_stored(t1, e, "t1act", "t1der")
        t2 = F.pixel_shuffle(F.conv2d(a1, _rw(p["tail.3.weight"], e), p["tail.3.bias"]), 2)
        a2, d2 = _gelu_stored(t2, e, "t2act", "t2der")
        if cap is not None:
            cap["t1act"], cap["t2act"], cap["t1der"], cap["t2der"] = a1, a2, d1, d2
        return conv3x3_reflect(a2, _rw(p["tail.6.weight"], e))
    t1 = F.pixel_shuffle(F.conv2d(x, _rw(p["tail.0.weight"], e), p["tail.0.bias"]), scale)
    a1, d1 = _gelu_stored(t1, e, "t1act", "t1der")
    if cap is not None:
        cap["t1act"], cap["t1der"] = a1, d1
    return conv3x3_reflect(a1, _rw(p["tail.3.weight"], e))


def forward(x: Tensor, p: Params, scale: int, n_blocks: int, rgb_range: float = 1.0,
            return_preclamp: bool = False, cap=None, emulate_bf16: bool = False, force=None, stage_report=None) -> Tensor:
    """M2Trans.forward (models/M2Trans_network.py:58-76).  ``cap`` (optional dict) collects the
    intermediates under the names of the HIP workspace tensors (NCHW here).
    ``emulate_bf16``: round at the storage points of the HIP path's bf16 mode; ``force`` / ``stage_report``: see the
    note on bf16 emulation above (force maps workspace names -- X0.., b0.d1.., b0.qkv1.., b0.xc.., t1act.. -- to the
    HIP path's NCHW fp32 copies of them).  The head conv runs in fp32 on fp32 weights in the HIP path, only its output
    is stored in bf16; the final conv's fp32 output (srpre) is not rounded."""
    e = _Emu(force, stage_report) if emulate_bf16 else None
    H, W = x.shape[-2:]
    x = pad_to_multiple(x)
    res = _r(conv3x3_reflect(x, p["head.weight"], p["head.bias"]), e, "X0")   # :63
    y = res
    for b in range(n_blocks):
        if cap is not None:
            cap[f"X{b}"] = y
        last = b == n_blocks - 1
        y = cftm(y, p, f"body.{b}.", cap, f"b{b}.", e, extra_residual=res if (e is not None and last) else None,
                 out_name=f"X{b + 1}")
    if e is None:
        y = res + y                                              # :70
    if cap is not None:
        cap[f"X{n_blocks}"] = y
    y = tail(y, p, scale, cap, e)                                # :72
    if cap is not None:
        cap["srpre"] = y
    if e is not None and stage_report is not None and force is not None and "srpre" in force:
        f = force["srpre"]
        d = (y.detach() - f).double()
        stage_report["srpre"] = (float(d.abs().max() / (f.double().abs().max() + 1e-30)),
                                 float(d.pow(2).mean().sqrt() / (f.double().pow(2).mean().sqrt() + 1e-30)))
    if return_preclamp:
        return y[:, :, : H * scale, : W * scale]
    y = torch.clamp(y, 0.0, rgb_range)                           # :74
    return y[:, :, : H * scale, : W * scale]                     # :76


# --------------------------------------------------------------------------------------
# train step   (train.py:76-82, 173-214, 358)
# --------------------------------------------------------------------------------------
def trainable_names(p: Params) -> List[str]:
    return [k for k in p if k not in FROZEN]


def l1_loss_and_grads(lr_img: Tensor, hr_img: Tensor, p: Params, scale: int, n_blocks: int,
                      rgb_range: float = 1.0, lambda_l1: float = 1.0,
                      loss_divisor: float | None = None, emulate_bf16: bool = False, force=None, stage_report=None):
    """loss = lambda_l1 * mean|sr - hr|  (train.py:76,199); gradients by CPU autograd.
    ``loss_divisor`` overrides the mean's denominator (data-parallel shards divide by the
    GLOBAL element count so that the sum over ranks equals the full-batch gradient)."""
    names = trainable_names(p)
    leaves = {k: p[k].detach().clone().requires_grad_(True) for k in names}
    q = dict(p)
    q.update(leaves)
    sr = forward(lr_img, q, scale, n_blocks, rgb_range, emulate_bf16=emulate_bf16, force=force, stage_report=stage_report)
    if loss_divisor is None:
        loss = (sr - hr_img).abs().mean() * lambda_l1
    else:
        loss = (sr - hr_img).abs().sum() / loss_divisor * lambda_l1
    grads = torch.autograd.grad(loss, [leaves[k] for k in names])
    return loss.detach(), sr.detach(), dict(zip(names, grads))


def cosine_lr(epoch: int, lr0: float = 1e-4, eta_min: float = 1e-6, t_max: float = 200.0) -> float:
    """CosineAnnealingLR closed form, stepped once per epoch (train.py:82,358)."""
    return eta_min + 0.5 * (lr0 - eta_min) * (1.0 + math.cos(math.pi * epoch / t_max))


def adam_update(param: Tensor, grad: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
                beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8):
    """torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8, weight_decay=0) single-tensor
    update (train.py:81,210).  ``step`` is the 1-based step count AFTER increment."""
    m = beta1 * m + (1 - beta1) * grad
    v = beta2 * v + (1 - beta2) * grad * grad
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    param = param - (lr / bc1) * (m / denom)
    return param, m, v


def gmsd(x: Tensor, y: Tensor, data_range: float = 1.0) -> Tensor:
    """piq.gmsd(x, y, data_range, reduction='none') as test.py:98 calls it (Xue et al. 2014, restated from the `piq`
    package's published source; PARITY UNPINNED: `piq` is not vendored by the reference nor installed here).
    x, y [B,3,H,W] -> [B] (computed in the input dtype; pass .double() for the yardstick)."""
    def prep(t):
        t = t / float(data_range)
        g = (0.299 * t[:, 0] + 0.587 * t[:, 1] + 0.114 * t[:, 2]).unsqueeze(1)          # rgb2yiq(x)[:, :1]
        dp = max(g.shape[2] % 2, g.shape[3] % 2)
        g = F.pad(g, [0, dp, 0, dp])
        return F.avg_pool2d(g, kernel_size=2, stride=2, padding=0)
    kx = torch.tensor([[-1.0, 0.0, 1.0]] * 3, dtype=x.dtype) / 3.0
    k = torch.stack([kx, kx.t()]).unsqueeze(1)                                          # prewitt and its transpose
    def grad(t):
        return torch.sqrt(torch.sum(F.conv2d(t, k, padding=1) ** 2, dim=-3, keepdim=True))
    gx, gy = grad(prep(x)), grad(prep(y))
    c = 170.0 / (255.0 ** 2)
    gms = (2.0 * gx * gy + c) / (gx ** 2 + gy ** 2 + c)
    mean = gms.mean(dim=[1, 2, 3], keepdim=True)
    return torch.pow(gms - mean, 2).mean(dim=[1, 2, 3]).sqrt()


# --------------------------------------------------------------------------------------
# metric   (utils.py:121-146,179-184; test.py:101-114 / train.py:299-312)
# --------------------------------------------------------------------------------------
def psnr_y(sr: Tensor, hr: Tensor, scale: int, rgb_range: float = 1.0) -> float:
    """Y-channel PSNR exactly as the reference's eval loop computes it, including the
    /255 inside rgb_to_ycbcr applied to inputs already in [0,1] (utils.py:136)."""
    def y_of(img):
        img = img / 255.0
        return (65.481 * img[..., 0, :, :] + 128.553 * img[..., 1, :, :]
                + 24.966 * img[..., 2, :, :] + 16.0).unsqueeze(-3)
    s, h = y_of(sr), y_of(hr)
    s = s[..., scale:-scale, scale:-scale]
    h = h[..., scale:-scale, scale:-scale]
    if rgb_range == 1:
        s, h = s * 255.0, h * 255.0
    diff = (s.double() - h.double()) / 255.0
    mse = diff.pow(2).mean()
    return float(-10.0 * math.log10(float(mse)))


def y_channel_eval(img: Tensor, scale: int, rgb_range: float = 1.0) -> Tensor:
    """The tensor both eval metrics are computed on: Y of utils.rgb_to_ycbcr (utils.py:121-146, with its
    /255 on [0,1] inputs), border crop of `scale` pixels and x255 (test.py:101-111).  [B,3,H,W] -> [B,1,H-2s,W-2s]."""
    img = img / 255.0
    y = (65.481 * img[..., 0, :, :] + 128.553 * img[..., 1, :, :] + 24.966 * img[..., 2, :, :] + 16.0).unsqueeze(-3)
    y = y[..., scale:-scale, scale:-scale]
    return y * 255.0 if rgb_range == 1 else y


def mse_y(sr: Tensor, hr: Tensor, scale: int, rgb_range: float = 1.0) -> Tensor:
    """Per-image mean of ((Y_sr - Y_hr)/255)^2 in fp64 (utils.py:179-184: calc_psnr = -10 log10 of it)."""
    d = (y_channel_eval(sr, scale, rgb_range).double() - y_channel_eval(hr, scale, rgb_range).double()) / 255.0
    return d.pow(2).flatten(1).mean(1)


SSIM_WIN, SSIM_SIGMA, SSIM_K1, SSIM_K2, SSIM_RANGE = 11, 1.5, 0.01, 0.03, 255.0


def ssim_window(dtype=torch.float32) -> Tensor:
    """Normalised 1-D Gaussian, 11 taps, sigma 1.5, computed in fp32 like the dependency does."""
    c = torch.arange(SSIM_WIN, dtype=torch.float32) - SSIM_WIN // 2
    g = torch.exp(-(c ** 2) / (2 * SSIM_SIGMA ** 2))
    return (g / g.sum()).to(dtype)


def ssim_y(sr: Tensor, hr: Tensor, scale: int, rgb_range: float = 1.0, dtype=torch.float32) -> Tensor:
    """utils.calc_ssim (utils.py:232-234) on the tensors test.py:101-113 hands it: `pytorch_msssim.ssim(X, Y,
    size_average=True)` with that package's defaults (data_range 255, 11-tap Gaussian sigma 1.5, K = (0.01, 0.03),
    no non-negativity clamp).  Returned per image (the reference evaluates with batch 1, so size_average is a no-op).

    PARITY UNPINNED: `pytorch_msssim` (pip, no version pin in the reference's README/environment) is not
    vendored under /root/reference and is not installed here; this restates its published algorithm (Wang et al.
    2004 as implemented by that package): separable VALID Gaussian filtering, first along H then along W, of
    X, Y, X*X, Y*Y, X*Y; sigma = filtered square minus squared mean; ssim_map = (2 mu1 mu2 + C1)/(mu1^2 + mu2^2 + C1)
    * (2 sigma12 + C2)/(sigma1^2 + sigma2^2 + C2); mean over the map.  Anchors used by the tests instead:
    SSIM(x, x) = 1, the closed form for constant images, symmetry, and an independent float64 scipy evaluation."""
    X = y_channel_eval(sr, scale, rgb_range).to(dtype)
    Y = y_channel_eval(hr, scale, rgb_range).to(dtype)
    g = ssim_window(dtype)
    C1, C2 = (SSIM_K1 * SSIM_RANGE) ** 2, (SSIM_K2 * SSIM_RANGE) ** 2

    def filt(t):
        if t.shape[-2] >= SSIM_WIN:        # the dependency skips (with a warning) an axis shorter than the window
            t = F.conv2d(t, g.view(1, 1, -1, 1))
        if t.shape[-1] >= SSIM_WIN:
            t = F.conv2d(t, g.view(1, 1, 1, -1))
        return t

    mu1, mu2 = filt(X), filt(Y)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = filt(X * X) - mu1_sq
    s2 = filt(Y * Y) - mu2_sq
    s12 = filt(X * Y) - mu12
    cs = (2 * s12 + C2) / (s1 + s2 + C2)
    m = ((2 * mu12 + C1) / (mu1_sq + mu2_sq + C1)) * cs
    return m.flatten(1).mean(1)


# --------------------------------------------------------------------------------------
# training input pipeline   (datas/us1k.py:16-36,146-170; utils.py ndarray2tensor:237-240)
# --------------------------------------------------------------------------------------
def crop_patch_draw(rng, lr_h: int, lr_w: int, patch_size: int, scale: int, augment: bool = True):
    """The random draws of datas/us1k.py:21,27-29 in the reference's order on a `random.Random`-like `rng`:
    column then row of the LR corner, then hflip, vflip, rot90 (each `random() > 0.5`)."""
    lp = patch_size // scale
    lx = rng.randrange(0, lr_w - lp + 1)
    ly = rng.randrange(0, lr_h - lp + 1)
    hflip = vflip = rot90 = False
    if augment:
        hflip = rng.random() > 0.5
        vflip = rng.random() > 0.5
        rot90 = rng.random() > 0.5
    return lx, ly, hflip, vflip, rot90


def crop_patch_apply(lr, hr, draw, patch_size: int, scale: int):
    """datas/us1k.py:22-35 + :169 for given draws: lr / hr are HWC arrays (the npy cache holds uint8);
    returns float32 CHW tensors already divided by 255 (what `US1K.__getitem__` hands to the DataLoader)."""
    import numpy as np
    lx, ly, hflip, vflip, rot90 = draw
    hp, lp = patch_size, patch_size // scale
    hx, hy = lx * scale, ly * scale
    lp_, hp_ = lr[ly:ly + lp, lx:lx + lp, :], hr[hy:hy + hp, hx:hx + hp, :]
    if hflip:
        lp_, hp_ = lp_[:, ::-1, :], hp_[:, ::-1, :]
    if vflip:
        lp_, hp_ = lp_[::-1, :, :], hp_[::-1, :, :]
    if rot90:
        lp_, hp_ = lp_.transpose(1, 0, 2), hp_.transpose(1, 0, 2)

    def to_tensor(a):
        return torch.from_numpy(np.ascontiguousarray(a.transpose((2, 0, 1)))).float() / 255.0

    return to_tensor(lp_), to_tensor(hp_)


def closed_form_u8_image(h: int, w: int, c: int = 3, phase: float = 0.0):
    """Deterministic uint8 HWC image (stand-in for one entry of the us1k npy cache)."""
    import numpy as np
    v = closed_form_image(1, c, h, w, phase=phase, dtype=torch.float64)[0]
    return np.ascontiguousarray((v * 255.0).round().clamp(0, 255).to(torch.uint8).permute(1, 2, 0).numpy())


def benchmark_item(lr, hr, scale: int):
    """datas/benchmark.py:62-72 `Benchmark.__getitem__` for one uint8 HWC pair: HR cropped to the LR size x scale,
    HWC -> CHW float32, / 255."""
    import numpy as np
    lr_h, lr_w, _ = lr.shape
    hr = hr[0:lr_h * scale, 0:lr_w * scale, :]

    def to_tensor(a):
        return torch.from_numpy(np.ascontiguousarray(a.transpose((2, 0, 1)))).float() / 255.0

    return to_tensor(lr), to_tensor(hr)


# --------------------------------------------------------------------------------------
# checkpoint wire format   (train.py:341-349)
# --------------------------------------------------------------------------------------
def checkpoint_manifest(ckpt: dict) -> dict:
    """Key / shape / dtype / scalar-value manifest of the dict train.py:341-349 saves (what the F1 pin compares:
    written from the real reference by pin_against_reference.py, required of m2trans_amd.checkpoint by the tests)."""
    def desc(v):
        if torch.is_tensor(v):
            return ["tensor", list(v.shape), str(v.dtype)]
        if isinstance(v, (list, tuple)):
            return [desc(x) for x in v]
        if isinstance(v, dict):
            return {str(k): desc(x) for k, x in v.items()}
        if isinstance(v, float):
            return float(v)
        return v if isinstance(v, (int, bool, str)) or v is None else repr(v)
    m = {"top_level_keys": list(ckpt.keys()), "epoch": ckpt["epoch"],
         "model_state_dict": desc(ckpt["model_state_dict"])}
    if "optimizer_state_dict" in ckpt:
        o = ckpt["optimizer_state_dict"]
        m["optimizer_state_dict"] = {"state": desc(o["state"]), "param_groups": desc(o["param_groups"])}
        m["scheduler_state_dict"] = desc(ckpt["scheduler_state_dict"])
    return m
