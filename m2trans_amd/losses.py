"""Drop-in for the reference's ``losses.SemanticLoss`` (losses.py:18-81) on MI355X.

Same constructor and call surface -- ``SemanticLoss(criterion='l1', N_patches=3)``,
``loss_clip(sr[i], hr[i], caption) -> Tensor[1]`` (train.py:78,205) -- but the MedCLIP image
tower (Swin-T 224 + Linear(768,512) + L2 norm) runs as hand-written gfx950 kernels behind the
C ABI (``m2t_swin_*``), and ``batch(sr, hr, captions)`` evaluates a whole batch with ONE encoder
launch sequence instead of the reference's B sequential B=1 calls, while drawing the patch
coordinates from the global torch CPU RNG in exactly the reference's order.

What the reference value really is (and what is reproduced): only the LAST patch's embeddings
survive the loop (losses.py:67-69), so for N_patches > 1 the value is
``|cos(E(sr_crop), T) - cos(E(hr_crop), T)| / N_patches`` on the last random 224x224 crop; the
bicubic 224x224 resize (losses.py:53-54) only matters for N_patches == 1.  Everything is
evaluated without gradient (losses.py:63): the term shifts the logged loss, not the update.

PARITY UNPINNED: the `medclip` package, its Swin/BERT checkpoints and tokenizer are not vendored
by the reference.  Weights are therefore injected (``load_state_dict`` with HF swin-tiny names,
4.24 or 5.x spelling, plus ``projection_head.weight``); text features are injected per caption
(``set_text_features``) or computed by the text tower built here (``load_text_encoder`` / ``load_medclip_state_dict``:
BERT-base forward + the MedCLIP head behind ``m2t_text_*``, with the reference's input_ids quirk of losses.py:65; the
tokenizer is a host callable) -- they are constants of the frozen text tower, cached per caption.  A caption with neither
RAISES (``synthetic_text=True`` opts into a deterministic hash stand-in for benchmarks).
"""
from __future__ import annotations

import ctypes as C
import hashlib
from typing import Dict, Iterable, List, Optional, Sequence

import torch
import torch.nn as nn

from . import _lib
from ._lib import M2TError

_V5_TO_V4 = (
    ("attention.q_proj", "attention.self.query"), ("attention.k_proj", "attention.self.key"),
    ("attention.v_proj", "attention.self.value"), ("attention.o_proj", "attention.output.dense"),
    ("attention.relative_position_bias.relative_position_bias_table", "attention.self.relative_position_bias_table"),
    ("mlp.fc1", "intermediate.dense"), ("mlp.fc2", "output.dense"),
)


class SwinEncoder:
    """m2t_swin handle + workspace + flat weights."""

    def __init__(self, max_images: int, dtype: int, device):
        lib = _lib.load()
        h = C.c_void_p()
        _lib.check(lib.m2t_swin_create(C.byref(h), max_images, dtype), "m2t_swin_create")
        self.handle, self.max_images, self.dtype, self.device = h, max_images, dtype, device
        self.names: List[str] = []
        self.slots: Dict[str, tuple] = {}
        for i in range(self.query("num_param_tensors")):
            n = lib.m2t_swin_param_name(h, i).decode()
            self.names.append(n)
            self.slots[n] = (self.query("param:" + n), self.query("numel:" + n))
        self.flat = torch.zeros(self.query("num_params"), dtype=torch.float32, device=device)
        self.workspace = torch.empty(self.query("workspace_bytes"), dtype=torch.uint8, device=device)
        self.loaded = False

    def query(self, key: str) -> int:
        v = _lib.load().m2t_swin_query(self.handle, key.encode())
        if v < 0:
            raise KeyError(key)
        return int(v)

    def load(self, state: Dict[str, torch.Tensor]):
        seen = set()
        for k, v in state.items():
            for a, b in _V5_TO_V4:
                k = k.replace(a, b)
            for prefix in ("vision_model.model.", "model.", "swin."):
                if k.startswith(prefix) and k[len(prefix):] in self.slots:
                    k = k[len(prefix):]
            if k in self.slots:
                o, n = self.slots[k]
                if v.numel() != n:
                    raise M2TError(f"shape mismatch for {k}: {tuple(v.shape)}")
                self.flat[o:o + n].copy_(v.reshape(-1).to(self.flat))
                seen.add(k)
        missing = [n for n in self.names if n not in seen]
        if missing:
            raise M2TError(f"missing Swin weights: {missing[:5]} ... ({len(missing)})")
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().m2t_swin_load_weights(self.handle, _lib.ptr(self.flat), _lib.ptr(self.workspace),
                                                         _lib.stream_ptr()), "m2t_swin_load_weights")
        self.loaded = True

    def encode(self, src: torch.Tensor, crops: Sequence[Sequence[int]]) -> torch.Tensor:
        """src [n_src,3,Hs,Ws] fp32 on the device; crops [(src index, row0, col0)] -> [n,512] unit-norm embeddings."""
        if not self.loaded:
            raise M2TError("SwinEncoder: weights not loaded")
        src = src.contiguous().float()
        n = len(crops)
        arr = (C.c_int * (3 * n))(*[int(v) for c in crops for v in c])
        emb = torch.empty(n, 512, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().m2t_swin_encode(self.handle, _lib.ptr(src), src.shape[0], src.shape[2], src.shape[3], arr, n,
                                                   _lib.ptr(emb), _lib.ptr(self.workspace), _lib.stream_ptr()), "m2t_swin_encode")
        return emb

    def encode_pair(self, src_a: torch.Tensor, src_b: torch.Tensor, crops: Sequence[Sequence[int]]) -> torch.Tensor:
        """like ``encode`` with the sources in two tensors (index < len(src_a): src_a, else src_b): no concatenation"""
        if not self.loaded:
            raise M2TError("SwinEncoder: weights not loaded")
        src_a, src_b = src_a.contiguous().float(), src_b.contiguous().float()
        if tuple(src_a.shape[1:]) != tuple(src_b.shape[1:]):
            raise M2TError("SwinEncoder.encode_pair: the two source tensors must have the same image shape")
        n = len(crops)
        arr = (C.c_int * (3 * n))(*[int(v) for c in crops for v in c])
        emb = torch.empty(n, 512, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().m2t_swin_encode_pair(self.handle, _lib.ptr(src_a), src_a.shape[0], _lib.ptr(src_b), src_b.shape[0],
                                                        src_a.shape[2], src_a.shape[3], arr, n, _lib.ptr(emb), _lib.ptr(self.workspace),
                                                        _lib.stream_ptr()), "m2t_swin_encode_pair")
        return emb

    def __del__(self):
        try:
            if self.handle:
                _lib.load().m2t_swin_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class TextEncoder:
    """m2t_text handle + workspace + flat weights: ``medmodel.encode_text`` (losses.py:65,74) = BERT-base -> mean of hidden
    states 1, 2, -1 -> Linear(768, 512) -> unit norm, as hand-written kernels behind ``m2t_text_*`` (csrc/m2t_text.hip)."""

    def __init__(self, max_seqs: int, max_len: int, dtype: int, device):
        lib = _lib.load()
        h = C.c_void_p()
        _lib.check(lib.m2t_text_create(C.byref(h), max_seqs, max_len, dtype), "m2t_text_create")
        self.handle, self.max_seqs, self.max_len, self.dtype, self.device = h, max_seqs, max_len, dtype, device
        self.names: List[str] = []
        self.slots: Dict[str, tuple] = {}
        for i in range(self.query("num_param_tensors")):
            n = lib.m2t_text_param_name(h, i).decode()
            self.names.append(n)
            self.slots[n] = (self.query("param:" + n), self.query("numel:" + n))
        self.flat = torch.zeros(self.query("num_params"), dtype=torch.float32, device=device)
        self.workspace = torch.empty(self.query("workspace_bytes"), dtype=torch.uint8, device=device)
        self.loaded = False

    def query(self, key: str) -> int:
        v = _lib.load().m2t_text_query(self.handle, key.encode())
        if v < 0:
            raise KeyError(key)
        return int(v)

    def load(self, state: Dict[str, torch.Tensor]):
        """HF BertModel names (transformers 4.24), optionally behind the MedCLIP checkpoint's ``text_model.model.`` /
        ``text_model.`` prefixes (or ``bert.`` / ``model.``), plus ``projection_head.weight``; ``pooler.*``,
        ``embeddings.position_ids`` and anything else are ignored; a word-embedding table with fewer rows than the
        Bio_ClinicalBERT vocabulary fills the first rows."""
        seen = set()
        for k, v in state.items():
            for prefix in ("text_model.model.", "text_model.", "model.", "bert."):
                if k.startswith(prefix) and k[len(prefix):] in self.slots:
                    k = k[len(prefix):]
                    break
            if k not in self.slots:
                continue
            o, n = self.slots[k]
            if k == "embeddings.word_embeddings.weight" and v.dim() == 2 and v.shape[1] == 768 and v.numel() < n:
                self.flat[o:o + v.numel()].copy_(v.reshape(-1).to(self.flat))
            elif v.numel() != n:
                raise M2TError(f"shape mismatch for {k}: {tuple(v.shape)}")
            else:
                self.flat[o:o + n].copy_(v.reshape(-1).to(self.flat))
            seen.add(k)
        missing = [n for n in self.names if n not in seen]
        if missing:
            raise M2TError(f"missing text-tower weights: {missing[:5]} ... ({len(missing)})")
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().m2t_text_load_weights(self.handle, _lib.ptr(self.flat), _lib.ptr(self.workspace),
                                                         _lib.stream_ptr()), "m2t_text_load_weights")
        self.loaded = True

    def encode(self, input_ids, attention_mask) -> torch.Tensor:
        """input_ids, attention_mask: [n, len] integer arrays (host) -> [n,512] unit-norm embeddings on the device."""
        if not self.loaded:
            raise M2TError("TextEncoder: weights not loaded")
        ids = torch.as_tensor(input_ids, dtype=torch.int32).reshape(-1, torch.as_tensor(input_ids).shape[-1]).contiguous().cpu()
        mask = torch.as_tensor(attention_mask, dtype=torch.int32).reshape(ids.shape).contiguous().cpu()
        n, ln = ids.shape
        emb = torch.empty(n, 512, dtype=torch.float32, device=self.device)
        ip, mp = C.cast(ids.data_ptr(), C.POINTER(C.c_int)), C.cast(mask.data_ptr(), C.POINTER(C.c_int))
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().m2t_text_encode(self.handle, ip, mp, n, ln, _lib.ptr(emb), _lib.ptr(self.workspace),
                                                   _lib.stream_ptr()), "m2t_text_encode")
        return emb

    def __del__(self):
        try:
            if self.handle:
                _lib.load().m2t_text_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def hash_text_feature(caption: str) -> torch.Tensor:
    """Deterministic stand-in for the frozen text tower (NOT the MedCLIP embedding)."""
    seed = int.from_bytes(hashlib.sha256(caption.encode("utf-8")).digest()[:8], "little") % (2 ** 31)
    g = torch.Generator().manual_seed(seed)
    return torch.randn(512, generator=g)


class SemanticLoss(nn.Module):
    def __init__(self, criterion: str = "l1", N_patches: int = 3, device=None, compute_dtype: str = "fp32",
                 max_batch: int = 32, synthetic_text: bool = False):
        super().__init__()
        # synthetic_text=True (benchmarks / tests with no MedCLIP weights): a caption without an injected text feature
        # gets a deterministic hash embedding.  The default is to RAISE: a silent stand-in behind a drop-in surface
        # would log a wrong regulariser value (the reference loads the real text tower itself, losses.py:22-23).
        self.synthetic_text = bool(synthetic_text)
        self.device = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
        self.N_patches = int(N_patches)
        self.compute_dtype = compute_dtype
        self.max_batch = int(max_batch)
        self._enc: Optional[SwinEncoder] = None
        self._state: Optional[Dict[str, torch.Tensor]] = None
        self._text: Dict[str, torch.Tensor] = {}
        self._text_dev = None
        self._tenc: Optional[TextEncoder] = None
        self._text_state: Optional[Dict[str, torch.Tensor]] = None
        self._by_count: Dict[int, torch.Tensor] = {}       # text feature per token count (see _text_feature)
        self.tokenizer = None                               # callable(caption) -> {'token_type_ids': [..], 'attention_mask': [..]}

    # ---- injected constants -----------------------------------------------------------------
    def load_image_encoder(self, state_dict: Dict[str, torch.Tensor]):
        """HF swin-tiny state dict (4.24 or 5.x names) + 'projection_head.weight' [512,768]."""
        self._state = {k: v.detach() for k, v in state_dict.items()}
        if self._enc is not None:
            self._enc.load(self._state)

    def load_text_encoder(self, state_dict: Dict[str, torch.Tensor], tokenizer=None):
        """The MedCLIP text tower: HF BertModel names (4.24) + 'projection_head.weight' [512,768] (see TextEncoder.load),
        and the tokenizer the reference builds with ``MedCLIPProcessor()`` (losses.py:25,64): any callable
        ``tokenizer(caption) -> mapping with 'token_type_ids' and 'attention_mask'`` (one sequence).  The tokenizer stays
        on the host; its vocabulary file ships with the checkpoint, not with this build."""
        self._text_state = {k: v.detach() for k, v in state_dict.items()}
        if tokenizer is not None:
            self.tokenizer = tokenizer
        self._tenc = None
        self._by_count.clear()

    def load_medclip_state_dict(self, state_dict: Dict[str, torch.Tensor], tokenizer=None):
        """A whole MedCLIPModel checkpoint (``vision_model.*`` / ``text_model.*`` / ``logit_scale``, the file
        pretrained/medclip-vit/readme.md:1-5 points at): both towers at once."""
        vis = {k[len("vision_model."):] if k.startswith("vision_model.projection_head") else k: v
               for k, v in state_dict.items() if k.startswith("vision_model.")}
        txt = {k: v for k, v in state_dict.items() if k.startswith("text_model.")}
        self.load_image_encoder(vis)
        self.load_text_encoder(txt, tokenizer)

    def _text_encoder(self) -> TextEncoder:
        if self._tenc is None:
            if self.device.type != "cuda":
                raise M2TError("SemanticLoss (MI355X build) needs a HIP device; there is no CPU fallback")
            code = _lib.F32 if self.compute_dtype in ("fp32", "float32") else _lib.BF16
            self._tenc = TextEncoder(1, 128, code, self.device)
            self._tenc.load(self._text_state)
        return self._tenc

    def set_text_features(self, table: Dict[str, torch.Tensor]):
        self._text.update({k: v.detach().float().reshape(512).cpu() for k, v in table.items()})
        self._text_dev = None

    def _encoder(self) -> SwinEncoder:
        if self.device.type != "cuda":
            raise M2TError("SemanticLoss (MI355X build) needs a HIP device; there is no CPU fallback")
        if self._enc is None:
            code = _lib.F32 if self.compute_dtype in ("fp32", "float32") else _lib.BF16
            self._enc = SwinEncoder(2 * self.max_batch, code, self.device)
            if self._state is None:
                raise M2TError("SemanticLoss: call load_image_encoder(state_dict) first (MedCLIP weights are not vendored)")
            self._enc.load(self._state)
        return self._enc

    def _text_feature(self, caption: str) -> torch.Tensor:
        t = self._text.get(caption)
        if t is not None:
            return t
        if self._text_state is not None and self.tokenizer is not None:
            # losses.py:64-65: tokenizer(text=[caption]) then encode_text(outputs['token_type_ids'], outputs['attention_mask'])
            # -- the token-type ids (zeros) travel in the input_ids slot, so the feature is a function of the token count
            tok = self.tokenizer(caption)
            ids = [int(v) for v in torch.as_tensor(tok["token_type_ids"]).reshape(-1).tolist()]
            mask = [int(v) for v in torch.as_tensor(tok["attention_mask"]).reshape(-1).tolist()]
            key = len(ids) if (not any(ids) and all(mask)) else None
            t = self._by_count.get(key) if key is not None else None
            if t is None:
                t = self._text_encoder().encode([ids], [mask])[0].cpu()
                if key is not None:
                    self._by_count[key] = t
            self._text[caption] = t
            return t
        if not self.synthetic_text:
            raise M2TError(f"SemanticLoss: no text feature for caption {caption!r}: inject the MedCLIP text embeddings with "
                           "set_text_features({caption: tensor[512]}) (the text tower is not part of this build), or construct "
                           "SemanticLoss(synthetic_text=True) for a benchmark with stand-in embeddings")
        t = hash_text_feature(caption)
        self._text[caption] = t                    # deterministic: computed once per caption
        return t

    # ---- reference semantics ------------------------------------------------------------------
    def createNRandompatches(self, hs: int, ws: int, N: int, patch_size: int = 224):
        """Coordinates only (losses.py:29-40): x then y per patch, torch.randint on the global CPU RNG;
        `x` indexes rows (size(2)), `y` columns -- naming kept from the reference."""
        out = []
        for _ in range(N):
            xcoord = int(torch.randint(hs - patch_size, ()))
            ycoord = int(torch.randint(ws - patch_size, ()))
            out.append((xcoord, ycoord))
        return out

    def batch(self, sr: torch.Tensor, hr: torch.Tensor, captions: Iterable[str]) -> torch.Tensor:
        """Sum over the batch of loss_clip(sr[i], hr[i], captions[i]) (train.py:203-205, without the
        lambda_clip factor) -> Tensor[1]; also leaves the per-sample values in ``self.last_per_sample``."""
        captions = list(captions)
        B = sr.shape[0]
        if len(captions) != B or tuple(sr.shape) != tuple(hr.shape):
            raise M2TError("SemanticLoss.batch: need one caption per sample and sr/hr of equal shape")
        if sr.shape[1] != 3:
            sr, hr = sr.repeat(1, 3, 1, 1), hr.repeat(1, 3, 1, 1)         # losses.py:47-49
        enc = self._encoder()
        if 2 * B > enc.max_images:
            raise M2TError(f"batch {B} exceeds max_batch {self.max_batch}")
        hs, ws = sr.shape[2], sr.shape[3]
        with torch.no_grad():
            if self.N_patches > 1:
                last = []
                for _ in range(B):                                       # same RNG order as B sequential calls
                    last.append(self.createNRandompatches(hs, ws, self.N_patches - 1)[-1])
                crops = [(i, last[i][0], last[i][1]) for i in range(B)] + [(B + i, last[i][0], last[i][1]) for i in range(B)]
                emb = enc.encode_pair(sr.detach(), hr.detach(), crops)   # SR crops then HR crops, no torch.cat of the batches
            else:
                src = torch.cat((sr.detach().float(), hr.detach().float()), dim=0).contiguous()
                small = torch.empty(2 * B, 3, 224, 224, dtype=torch.float32, device=sr.device)
                with torch.cuda.device(sr.device):
                    _lib.check(_lib.load().m2t_bicubic_resize(_lib.ptr(src), _lib.ptr(small), 2 * B * 3, hs, ws, 224, 224,
                                                              _lib.stream_ptr()), "m2t_bicubic_resize")
                emb = enc.encode(small, [(i, 0, 0) for i in range(2 * B)])
            dev = sr.device
            key = tuple(captions)
            if self._text_dev is None or self._text_dev[0] != key or self._text_dev[1].device != dev:
                # the text features are constants of the frozen model: one upload per distinct caption list
                self._text_dev = (key, torch.stack([self._text_feature(c) for c in captions]).to(dev))
            text = self._text_dev[1]
            per = torch.empty(B, dtype=torch.float32, device=dev)
            tot = torch.empty(1, dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _lib.check(_lib.load().m2t_semantic_loss(_lib.ptr(emb), _lib.ptr(text), B, self.N_patches, _lib.ptr(per),
                                                         _lib.ptr(tot), _lib.stream_ptr()), "m2t_semantic_loss")
        self.last_per_sample = per
        self.last_embeddings = emb            # [2B,512]: SR rows then HR rows, unit norm (kept for inspection)
        return tot

    def __call__(self, x: torch.Tensor, y: torch.Tensor, batch_tokens: str) -> torch.Tensor:
        """Single sample, as called by train.py:205: x, y [3,Hs,Ws] (or [1,Hs,Ws]) -> Tensor[1]."""
        return self.batch(x.unsqueeze(0).to(self.device), y.unsqueeze(0).to(self.device), [batch_tokens])
