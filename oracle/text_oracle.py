"""CPU oracle for the TEXT tower behind SemanticLoss (losses.py:22-25,64-65,74)  --  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  ``self.medmodel.encode_text`` belongs to the un-vendored, un-pinned pip package ``medclip``
(README.md:42); its checkpoint (Bio_ClinicalBERT inside medclip-vit-pretrained.zip, pretrained/medclip-vit/readme.md:1-5)
is not under /root/reference.  What is restated here, from the published algorithms:

  * BERT-base forward (Devlin et al. 2018; HF ``modeling_bert.py`` of transformers 4.24, environment.yml:176): word +
    position + token-type embeddings, LayerNorm(eps 1e-12), 12 x [self-attention (12 heads x 64, scores / 8, additive
    -inf mask on padded keys) -> dense + residual -> LayerNorm -> dense 3072 + erf-GELU -> dense + residual -> LayerNorm],
    returning every hidden state -- validated numerically against ``transformers.BertModel`` with random weights by
    tests/test_text_oracle.py (v5.x in this image; same parameter names as 4.24 for BERT);
  * the MedCLIP text head as the public package defines it (MedCLIPTextModel.forward): stack hidden states 1, 2 and -1,
    mean over the tokens (UNMASKED ``.mean(2)``), mean over the three layers, Linear(768, 512, bias=False);
    MedCLIPModel.encode_text divides by the L2 norm (losses.py:74 normalises once more: idempotent);
  * the reference's call-site quirk: ``encode_text(outputs['token_type_ids'], outputs['attention_mask'])`` (losses.py:65)
    -- the token-type ids (all zero for a single sentence) go where the input ids belong, so every caption of the same
    token count has the same text feature: ``reference_text_feature(n_tokens, params)``.

Weights are keyed by the HF 4.24 BertModel names (``encoder.layer.{i}.attention.self.query.weight`` ...) plus
``projection_head.weight``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

HIDDEN, HEADS, FF, LAYERS, VOCAB, POS, TYPES, PROJ = 768, 12, 3072, 12, 28996, 512, 2, 512
LN_EPS = 1e-12


def text_param_shapes(vocab: int = VOCAB) -> "Dict[str, Tuple[int, ...]]":
    s: Dict[str, Tuple[int, ...]] = {}
    s["embeddings.word_embeddings.weight"] = (vocab, HIDDEN)
    s["embeddings.position_embeddings.weight"] = (POS, HIDDEN)
    s["embeddings.token_type_embeddings.weight"] = (TYPES, HIDDEN)
    s["embeddings.LayerNorm.weight"] = (HIDDEN,)
    s["embeddings.LayerNorm.bias"] = (HIDDEN,)
    for l in range(LAYERS):
        p = f"encoder.layer.{l}."
        for nm in ("query", "key", "value"):
            s[p + f"attention.self.{nm}.weight"] = (HIDDEN, HIDDEN)
            s[p + f"attention.self.{nm}.bias"] = (HIDDEN,)
        s[p + "attention.output.dense.weight"] = (HIDDEN, HIDDEN)
        s[p + "attention.output.dense.bias"] = (HIDDEN,)
        s[p + "attention.output.LayerNorm.weight"] = (HIDDEN,)
        s[p + "attention.output.LayerNorm.bias"] = (HIDDEN,)
        s[p + "intermediate.dense.weight"] = (FF, HIDDEN)
        s[p + "intermediate.dense.bias"] = (FF,)
        s[p + "output.dense.weight"] = (HIDDEN, FF)
        s[p + "output.dense.bias"] = (HIDDEN,)
        s[p + "output.LayerNorm.weight"] = (HIDDEN,)
        s[p + "output.LayerNorm.bias"] = (HIDDEN,)
    s["projection_head.weight"] = (PROJ, HIDDEN)
    return s


def closed_form_text_params(vocab: int = VOCAB, dtype=torch.float32) -> "Dict[str, Tensor]":
    """Deterministic, RNG-free weights (HF init scale 0.02; LayerNorm gains near 1) so every box regenerates them."""
    out: Dict[str, Tensor] = {}
    for k, (name, shp) in enumerate(text_param_shapes(vocab).items()):
        n = int(math.prod(shp))
        idx = torch.arange(n, dtype=torch.float64)
        base = torch.sin(idx * (0.61803398875 + 0.0007 * k) + 0.41 * k) + 0.5 * torch.sin(idx * 1.7320508 + 0.9 * k)
        if "LayerNorm.weight" in name:
            v = 1.0 + 0.05 * base
        elif name.endswith("bias"):
            v = 0.02 * base
        elif "embeddings" in name:
            v = 0.05 * base
        elif name == "projection_head.weight":
            v = 0.04 * base
        else:
            v = 0.035 * base
        out[name] = v.reshape(shp).to(dtype)
    return out


def bert_hidden_states(input_ids: Tensor, attention_mask: Tensor, p: "Dict[str, Tensor]") -> "List[Tensor]":
    """BertModel(input_ids, attention_mask, output_hidden_states=True).hidden_states: 13 tensors [B, L, 768]
    (position ids 0..L-1, token types 0: the defaults BertModel uses when only these two arguments are given)."""
    B, L = input_ids.shape
    x = p["embeddings.word_embeddings.weight"][input_ids] + p["embeddings.token_type_embeddings.weight"][0] \
        + p["embeddings.position_embeddings.weight"][:L]
    x = F.layer_norm(x, (HIDDEN,), p["embeddings.LayerNorm.weight"], p["embeddings.LayerNorm.bias"], LN_EPS)
    hs = [x]
    neg = torch.finfo(x.dtype).min
    bias = (1.0 - attention_mask.to(x.dtype))[:, None, None, :] * neg          # [B,1,1,L]: additive mask on the keys
    for l in range(LAYERS):
        b = f"encoder.layer.{l}."
        def lin(t, nm):
            return t @ p[b + nm + ".weight"].T + p[b + nm + ".bias"]
        q = lin(x, "attention.self.query").view(B, L, HEADS, 64).transpose(1, 2)
        k = lin(x, "attention.self.key").view(B, L, HEADS, 64).transpose(1, 2)
        v = lin(x, "attention.self.value").view(B, L, HEADS, 64).transpose(1, 2)
        s = q @ k.transpose(-1, -2) / 8.0 + bias
        a = torch.softmax(s, dim=-1) @ v
        a = a.transpose(1, 2).reshape(B, L, HIDDEN)
        x = F.layer_norm(lin(a, "attention.output.dense") + x, (HIDDEN,), p[b + "attention.output.LayerNorm.weight"],
                         p[b + "attention.output.LayerNorm.bias"], LN_EPS)
        h = F.gelu(lin(x, "intermediate.dense"))                                # erf GELU ("gelu" in the BERT config)
        x = F.layer_norm(lin(h, "output.dense") + x, (HIDDEN,), p[b + "output.LayerNorm.weight"], p[b + "output.LayerNorm.bias"], LN_EPS)
        hs.append(x)
    return hs


def encode_text(input_ids: Tensor, attention_mask: Tensor, p: "Dict[str, Tensor]") -> Tensor:
    """MedCLIPModel.encode_text: MedCLIPTextModel.forward (hidden states 1, 2, -1 -> token mean -> layer mean ->
    projection) then L2 normalisation.  -> [B, 512]"""
    hs = bert_hidden_states(input_ids, attention_mask, p)
    e = torch.stack([hs[1], hs[2], hs[-1]]).permute(1, 0, 2, 3).mean(2).mean(1)
    e = e @ p["projection_head.weight"].T
    return e / e.norm(dim=-1, keepdim=True)


def reference_text_feature(n_tokens: int, p: "Dict[str, Tensor]") -> Tensor:
    """What losses.py:64-65,74 computes for ANY caption that tokenises to n_tokens ids: the tokenizer's token_type_ids
    (zeros) are passed as input_ids, attention_mask is all ones (one sentence, no padding)."""
    ids = torch.zeros(1, n_tokens, dtype=torch.long)
    return encode_text(ids, torch.ones(1, n_tokens, dtype=torch.long), p)[0]
