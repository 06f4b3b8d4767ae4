"""CPU: the BERT / MedCLIP text-tower restatement (oracle/text_oracle.py) against transformers.BertModel with the same
random weights (the third-party implementation medclip wraps; 5.x here, the reference pins 4.24.0: same parameter
names and arithmetic for BERT).  The MedCLIP sub-path stays parity-unpinned: package and checkpoint are not vendored."""
import pytest
import torch

from oracle import text_oracle as T

VOCAB = 1000        # a small vocabulary keeps the CPU test light; the arithmetic does not depend on it


def test_bert_restatement_matches_transformers_bertmodel():
    transformers = pytest.importorskip("transformers")
    p = T.closed_form_text_params(VOCAB)
    cfg = transformers.BertConfig(vocab_size=VOCAB)
    model = transformers.BertModel(cfg).eval()
    sd = model.state_dict()
    ren = {k: v for k, v in p.items() if k in sd}
    missing = [k for k in sd if k not in ren and not k.startswith("pooler.") and "position_ids" not in k]
    assert not missing, missing[:10]
    model.load_state_dict(ren, strict=False)
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, VOCAB, (2, 17), generator=g)
    mask = torch.ones(2, 17, dtype=torch.long)
    mask[1, 12:] = 0                                                  # a padded sequence
    with torch.no_grad():
        want = model(input_ids=ids, attention_mask=mask, output_hidden_states=True).hidden_states
        got = T.bert_hidden_states(ids, mask, p)
    assert len(got) == len(want) == 13
    for l in (0, 1, 2, 12):
        a, b = got[l], want[l]
        # padded QUERY rows are garbage in both (they attend to the real keys only); compare what encode_text reads: all rows
        assert float((a - b).abs().max() / b.abs().max()) < 2e-5, l


def test_medclip_text_head_and_the_input_ids_quirk():
    p = T.closed_form_text_params(VOCAB)
    e = T.encode_text(torch.tensor([[5, 6, 7, 8]]), torch.ones(1, 4, dtype=torch.long), p)
    assert e.shape == (1, 512) and abs(float(e.norm()) - 1.0) < 1e-6
    # losses.py:65 passes token_type_ids (zeros) as input_ids: the feature depends on the token COUNT only
    f9 = T.reference_text_feature(9, p)
    same = T.encode_text(torch.zeros(1, 9, dtype=torch.long), torch.ones(1, 9, dtype=torch.long), p)[0]
    assert torch.equal(f9, same)
    assert float((f9 - T.reference_text_feature(10, p)).abs().max()) > 1e-4
    # explicit recomputation of the pooling order: token mean, then layer mean, then projection
    hs = T.bert_hidden_states(torch.zeros(1, 9, dtype=torch.long), torch.ones(1, 9, dtype=torch.long), p)
    pooled = (hs[1].mean(1) + hs[2].mean(1) + hs[12].mean(1)) / 3.0
    want = pooled @ p["projection_head.weight"].T
    want = want / want.norm()
    assert float((f9 - want[0]).abs().max()) < 1e-6


def test_text_param_inventory_is_bert_base():
    n = sum(int(torch.tensor(s).prod()) for k, s in T.text_param_shapes().items() if k != "projection_head.weight")
    # BertModel(vocab 28996) without its pooler: 108 310 272 - 590 592
    assert n == 108310272 - (768 * 768 + 768)
