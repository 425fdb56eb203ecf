"""Cross-checks of oracle/encoders_oracle.py (parity unpinned by the reference itself, see that file's header)."""
import numpy as np
import pytest
import torch

from oracle import encoders_oracle as E


def test_bert_restatement_matches_installed_transformers():
    transformers = pytest.importorskip("transformers")
    from mmgclip.networks.bert import BertConfigLite, _hf_layout, hf_config_dict
    torch.manual_seed(0)
    cfg = BertConfigLite(vocab_size=500, num_hidden_layers=2)
    mine = _hf_layout(cfg)
    hf = transformers.BertModel(transformers.BertConfig(**hf_config_dict(cfg), attn_implementation="eager"))
    hf.eval()
    missing, unexpected = hf.load_state_dict(mine.state_dict(), strict=False)
    assert not [k for k in missing if "position_ids" not in k], missing
    ids = torch.randint(1, 500, (3, 20))
    mask = torch.ones(3, 20, dtype=torch.long)
    mask[1, 12:] = 0
    mask[2, 5:] = 0
    with torch.no_grad():
        ref = hf(input_ids=ids, attention_mask=mask, token_type_ids=torch.zeros_like(ids))["last_hidden_state"]
        got = E.bert_forward(mine.state_dict(), ids, mask, None, heads=12)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)


def test_convnext_geometry_matches_notebook():
    """notebooks/clf_convnext_tiny_experimental.ipynb:641,682: [1,1,1906,818] -> features [1,768,59,25] -> avgpool [1,768,1,1]
    (checked on a proportionally smaller input to keep the CPU suite fast, plus the exact /32 floor rule)."""
    from mmgclip.networks.convnext import build_features
    torch.manual_seed(0)
    feats = build_features("tiny", in_chans=1)
    sd = {"features." + k: v for k, v in feats.state_dict().items()}
    x = torch.rand(1, 1, 1906 // 8 // 32 * 32 + 32, 128)
    with torch.no_grad():
        pooled, fmap = E.convnext_forward(sd, x)
    assert fmap.shape == (1, 768, x.shape[2] // 32, 4) and pooled.shape == (1, 768, 1, 1)
    assert (1906 // 32, 818 // 32) == (59, 25)
    assert sum(p.numel() for p in feats.parameters()) == 27_815_520   # ConvNeXt-T trunk with a 1-channel stem
