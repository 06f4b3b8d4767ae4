"""-m gpu: the MedCLIP text tower on the device (m2t_text_*, csrc/m2t_text.hip) against oracle/text_oracle.py (which is
checked against transformers.BertModel on the CPU): embeddings, the reference's input_ids quirk, checkpoint-layout
loading.  PARITY UNPINNED against the real package / weights (not vendored by the reference)."""
import pytest
import torch

from oracle import text_oracle as T

pytestmark = pytest.mark.gpu

VOCAB = 1200


def _encoder(dtype, params):
    from m2trans_amd import _lib
    from m2trans_amd.losses import TextEncoder
    enc = TextEncoder(4, 32, _lib.F32 if dtype == "fp32" else _lib.BF16, torch.device("cuda"))
    enc.load(params)
    return enc


@pytest.mark.parametrize("dtype,tol", [("fp32", 3e-5), ("bf16", 4e-2)])
def test_text_encode_matches_oracle(dtype, tol):
    p = T.closed_form_text_params(VOCAB)
    enc = _encoder(dtype, p)
    g = torch.Generator().manual_seed(11)
    ids = torch.randint(0, VOCAB, (3, 19), generator=g)
    mask = torch.ones(3, 19, dtype=torch.long)
    mask[2, 11:] = 0                                   # padded keys are masked; the token mean is NOT (package semantics)
    got = enc.encode(ids, mask).cpu()
    want = T.encode_text(ids, mask, p)
    assert got.shape == (3, 512)
    assert float((got.norm(dim=1) - 1).abs().max()) < 1e-4
    assert float((got - want).norm(dim=1).max()) < tol, float((got - want).norm(dim=1).max())


def test_semantic_loss_uses_the_token_count_quirk_and_checkpoint_prefixes():
    """losses.py:64-65: token_type_ids in the input_ids slot -> one feature per token count; weights arrive under the
    MedCLIP checkpoint's prefixes (text_model.model.*, text_model.projection_head.weight, pooler ignored)."""
    from m2trans_amd.losses import SemanticLoss
    p = T.closed_form_text_params(VOCAB)
    ck = {("text_model.projection_head.weight" if k == "projection_head.weight" else "text_model.model." + k): v for k, v in p.items()}
    ck["text_model.model.pooler.dense.weight"] = torch.zeros(768, 768)
    ck["text_model.model.embeddings.position_ids"] = torch.arange(512)[None]
    ck["logit_scale"] = torch.tensor(0.07)

    def tok(caption):                                  # stand-in tokenizer: [CLS] + words + [SEP]
        n = len(caption.split()) + 2
        return {"input_ids": list(range(5, 5 + n)), "token_type_ids": [0] * n, "attention_mask": [1] * n}

    sl = SemanticLoss(criterion="l1", N_patches=3, device="cuda", compute_dtype="fp32")
    sl.load_text_encoder(ck, tokenizer=tok)
    a = sl._text_feature("thyroid nodule with calcification")
    b = sl._text_feature("breast lesion irregular margin")          # same token count, different words
    c = sl._text_feature("normal liver")
    assert torch.equal(a, b)
    want = T.reference_text_feature(6, p)
    assert float((a - want).norm()) < 3e-5 and float((c - T.reference_text_feature(4, p)).norm()) < 3e-5
    assert float((a - c).norm()) > 1e-4
