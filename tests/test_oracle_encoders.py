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


def test_resnet50_oracle_equals_the_torch_modules_of_the_same_tree():
    """oracle.resnet50_forward (functional, from a state dict) vs the nn.Conv2d / nn.BatchNorm2d modules of the torchvision-layout
    tree run directly in torch - train mode (batch statistics) and eval mode (running statistics), 4-D and the reference's 2-D
    input (mmgclip/networks/encoder.py:101-117)."""
    import torch.nn.functional as F
    from mmgclip.networks.resnet import _TorchvisionResNet
    torch.manual_seed(0)
    m = _TorchvisionResNet()
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for n_, b in m.named_buffers():
            if n_.endswith("running_mean"):
                b.copy_(0.1 * torch.randn(b.shape, generator=g))
            elif n_.endswith("running_var"):
                b.copy_(0.5 + torch.rand(b.shape, generator=g))

    def run(x):
        if x.dim() == 2:
            x = x.view(x.shape[0], 1, 1, x.shape[1]).repeat(1, 3, 1, 1)
        x = F.max_pool2d(F.relu(m.bn1(m.conv1(x))), 3, 2, 1)
        for li in range(4):
            for blk in getattr(m, f"layer{li + 1}"):
                y = F.relu(blk.bn1(blk.conv1(x)))
                y = F.relu(blk.bn2(blk.conv2(y)))
                y = blk.bn3(blk.conv3(y))
                x = F.relu(y + (blk.downsample(x) if blk.downsample is not None else x))
        return F.adaptive_avg_pool2d(x, 1).flatten(1)

    for x in (torch.rand(2, 3, 64, 64, generator=g), torch.rand(3, 96, generator=g)):
        sd = {k: v.clone() for k, v in m.state_dict().items()}      # (a train-mode run of the modules moves their running statistics)
        with torch.no_grad():
            m.eval()
            assert torch.allclose(E.resnet50_forward(sd, x, train_bn=False), run(x), rtol=1e-4, atol=1e-5)
            m.train()
            assert torch.allclose(E.resnet50_forward(sd, x, train_bn=True), run(x), rtol=1e-3, atol=1e-4)
