"""oracle/rlutrans_oracle.py -- TEST INFRASTRUCTURE ONLY (see oracle/m2trans_oracle.py for the rules).

Functional CPU restatement of the reference's `util/rlutrans.py` token block (`Mlp` :11-27,
`EffAttention` :30-67, `TransBlock` :70-87).  The reference never imports that file (dead code), so this
row has no call site to anchor on; the restatement is pinned against the imported reference module itself
(oracle/pin_against_reference.py section 11 -> tests/golden/transblock.npz).

Parameters are the reference's `state_dict()` of `TransBlock(n_feat=64, dim=64)`, by name:
  atten.reduce.weight [64,64]  atten.qkv.weight [192,64]  atten.proj.weight [64,64]  atten.proj.bias [64]
  norm1.weight/bias [64]  mlp.fc1.weight [16,64]  mlp.fc1.bias [16]  mlp.fc2.weight [64,16]  mlp.fc2.bias [64]
  norm2.weight/bias [64]
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def chunk_length(n_tokens: int) -> int:
    """`torch.split(q, math.ceil(N//16), dim=-2)` (util/rlutrans.py:53-55): the integer division happens first, so
    the chunk length is floor(N / 16) and a token count that is not a multiple of 16 leaves a 17th, shorter chunk."""
    return math.ceil(n_tokens // 16)


def eff_attention(x: Tensor, p: Dict[str, Tensor], num_heads: int = 8) -> Tensor:
    """EffAttention.forward (util/rlutrans.py:47-67): reduce -> qkv -> softmax attention inside token chunks -> proj."""
    x = F.linear(x, p["atten.reduce.weight"])                                    # :48 (qkv_bias=False)
    B, N, C = x.shape
    hd = C // num_heads
    scale = hd ** -0.5                                                           # :35
    qkv = F.linear(x, p["atten.qkv.weight"]).reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)   # :50
    q, k, v = qkv[0], qkv[1], qkv[2]                                             # [B, heads, N, hd]
    L = chunk_length(N)
    out = []
    for q_c, k_c, v_c in zip(torch.split(q, L, dim=-2), torch.split(k, L, dim=-2), torch.split(v, L, dim=-2)):   # :53-58
        attn = ((q_c @ k_c.transpose(-2, -1)) * scale).softmax(dim=-1)           # :59-60
        out.append((attn @ v_c).transpose(1, 2))                                 # :62  [B, Lc, heads, hd]
    x = torch.cat(out, dim=1).reshape(B, N, C)                                   # :64-65
    return F.linear(x, p["atten.proj.weight"], p["atten.proj.bias"])             # :66


def mlp(x: Tensor, p: Dict[str, Tensor]) -> Tensor:
    """Mlp.forward (util/rlutrans.py:21-27), act = ReLU, dropout 0."""
    return F.linear(F.relu(F.linear(x, p["mlp.fc1.weight"], p["mlp.fc1.bias"])), p["mlp.fc2.weight"], p["mlp.fc2.bias"])


def trans_block(x: Tensor, p: Dict[str, Tensor], num_heads: int = 8) -> Tensor:
    """TransBlock.forward (util/rlutrans.py:82-87): x + atten(LN(x)); x + mlp(LN(x)).  LayerNorm eps 1e-5."""
    C = x.shape[-1]
    x = x + eff_attention(F.layer_norm(x, (C,), p["norm1.weight"], p["norm1.bias"], 1e-5), p, num_heads)
    x = x + mlp(F.layer_norm(x, (C,), p["norm2.weight"], p["norm2.bias"], 1e-5), p)
    return x


def closed_form_params(dim: int = 64, dtype=torch.float32) -> Dict[str, Tensor]:
    """Deterministic parameters (no RNG) for tests that must not depend on an init order."""
    shapes = {"atten.reduce.weight": (dim, dim), "atten.qkv.weight": (3 * dim, dim), "atten.proj.weight": (dim, dim),
              "atten.proj.bias": (dim,), "norm1.weight": (dim,), "norm1.bias": (dim,), "mlp.fc1.weight": (dim // 4, dim),
              "mlp.fc1.bias": (dim // 4,), "mlp.fc2.weight": (dim, dim // 4), "mlp.fc2.bias": (dim,),
              "norm2.weight": (dim,), "norm2.bias": (dim,)}
    out = {}
    for i, (k, s) in enumerate(shapes.items()):
        n = 1
        for d in s:
            n *= d
        t = torch.sin(torch.arange(n, dtype=torch.float64) * (0.37 + 0.11 * i) + i).reshape(s)
        if k.endswith("norm1.weight") or k.endswith("norm2.weight"):
            t = 1.0 + 0.1 * t
        elif len(s) == 2:
            t = t * (1.0 / math.sqrt(s[1]))
        else:
            t = 0.1 * t
        out[k] = t.to(dtype)
    return out
