"""Host mirror of the reference's ``util/rlutrans.py``: ``Mlp``, ``EffAttention``, ``TransBlock``
with the reference's constructor arguments, parameter names, shapes and default initialisation, so a ``state_dict``
moves between the two unchanged.  The reference never imports that file (dead code); the block is provided for
completeness, forward only: ``TransBlock.forward`` runs as HIP kernels behind ``m2t_transblock_forward``
(csrc/m2t_rlutrans.hip) and raises without the library or for a CPU tensor -- there is no eager fallback.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib
from ._lib import M2TError


class Mlp(nn.Module):
    """util/rlutrans.py:11-27 (parameters only; evaluated inside TransBlock.forward)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.ReLU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features // 4
        if act_layer is not nn.ReLU or drop != 0.:
            raise M2TError("Mlp: the MI355X build implements act_layer=nn.ReLU, drop=0 (the reference's only use)")
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, out_features)


class EffAttention(nn.Module):
    """util/rlutrans.py:30-67 (parameters only; evaluated inside TransBlock.forward)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        if qkv_bias or qk_scale is not None or attn_drop != 0. or proj_drop != 0.:
            raise M2TError("EffAttention: the MI355X build implements qkv_bias=False, qk_scale=None, no dropout")
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.reduce = nn.Linear(dim, dim, bias=qkv_bias)
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class TransBlock(nn.Module):
    """util/rlutrans.py:70-87.  ``compute_dtype``: "fp32" (parity) or "bf16"."""

    def __init__(self, n_feat=64, dim=64, num_heads=8, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.ReLU, norm_layer=nn.LayerNorm, compute_dtype: str = "fp32"):
        super().__init__()
        if dim != 64 or num_heads != 8 or norm_layer is not nn.LayerNorm:
            raise M2TError("TransBlock: the MI355X build implements dim=64, num_heads=8, LayerNorm (the reference's defaults)")
        self.dim = dim
        # module creation order = the reference's, so a seeded default init draws the same weights
        self.atten = EffAttention(dim, num_heads=num_heads, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.)
        self.norm1 = nn.LayerNorm(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=dim // 4, act_layer=act_layer, drop=drop)
        self.norm2 = nn.LayerNorm(dim)
        self.compute_dtype = compute_dtype
        self._ws = None

    def invalidate(self) -> None:
        """Drop the cached flat parameter vector (call after writing parameters through ``.data``)."""
        self._flat_key = None

    def _apply(self, fn, *a, **k):
        self._flat_key = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._flat_key = None
        return super().load_state_dict(*a, **k)

    def train(self, mode: bool = True):
        self._flat_key = None
        return super().train(mode)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.device.type != "cuda":
            raise M2TError("TransBlock (MI355X build): needs a HIP device tensor; there is no CPU fallback")
        if x.dim() != 3 or x.shape[2] != self.dim:
            raise M2TError("TransBlock: input must be [B, N, 64]")
        B, N, _ = x.shape
        if N < 16:
            raise M2TError("TransBlock: N >= 16 (the reference splits the tokens into chunks of N // 16, util/rlutrans.py:53)")
        lib = _lib.load()
        code = _lib.F32 if self.compute_dtype in ("fp32", "float32") else _lib.BF16
        xc = x.detach().contiguous().float()
        # the 22 928 parameters as one fp32 device vector in state_dict order: rebuilt only when a parameter changed (in-place
        # version counters) or moved, not on every forward
        # Contract: the cache is trusted only in eval mode under no_grad (inference loops); while the module trains or autograd is on
        # the vector is rebuilt on every call (one cat of 16 small tensors), because writes through ``p.data`` (EMA, weight clipping,
        # legacy loaders) do not bump the version counters the key is made of.  ``_apply`` (.to / .cuda / .half), ``load_state_dict``
        # and ``train()`` / ``eval()`` drop it; ``invalidate()`` does so explicitly.
        vals = list(self.state_dict().values())
        key = (str(x.device),) + tuple((v.data_ptr(), v._version) for v in vals)
        trust = (not self.training) and (not torch.is_grad_enabled())
        if not trust or getattr(self, "_flat_key", None) != key:
            self._flat = torch.cat([v.detach().reshape(-1).float() for v in vals]).to(x.device).contiguous()
            self._flat_key = key if trust else None
        flat = self._flat
        if flat.numel() != 22928:
            raise M2TError("TransBlock: unexpected parameter count")
        need = int(lib.m2t_transblock_workspace_bytes(B, N, code))
        if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        y = torch.empty_like(xc)
        with torch.cuda.device(x.device):
            _lib.check(lib.m2t_transblock_forward(_lib.ptr(flat), _lib.ptr(xc), _lib.ptr(y), B, N, code, _lib.ptr(self._ws),
                                                  _lib.stream_ptr()), "m2t_transblock_forward")
        return y
