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


def test_bert_training_mode_dropout_positions_match_installed_transformers(monkeypatch):
    """The oracle's training-mode forward (counter-based masks, oracle/dropout_oracle.py) against HF BertModel in train() with
    torch's dropout function replaced by one that applies the SAME masks in call order: pins WHERE dropout acts (embeddings after
    LayerNorm, attention probabilities, the two dense outputs before their residual adds) to the third-party implementation the
    reference calls (mmgclip/networks/encoder.py:156 under ClassifierExperiment.py:97)."""
    transformers = pytest.importorskip("transformers")
    from mmgclip.networks.bert import BertConfigLite, _hf_layout, hf_config_dict
    from oracle import dropout_oracle as D
    torch.manual_seed(0)
    cfg = BertConfigLite(vocab_size=500, num_hidden_layers=2)
    mine = _hf_layout(cfg)
    hcfg = dict(hf_config_dict(cfg), hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    hf = transformers.BertModel(transformers.BertConfig(**hcfg, attn_implementation="eager"))
    hf.load_state_dict(mine.state_dict(), strict=False)
    hf.train()
    B, S, H, heads, seed = 3, 20, cfg.hidden_size, 12, 2024
    ids = torch.randint(1, 500, (B, S))
    mask = torch.ones(B, S, dtype=torch.long)
    mask[1, 12:] = 0
    mask[2, 5:] = 0
    calls = []

    def masked_dropout(x, p=0.5, training=True, inplace=False):
        n = len(calls)
        calls.append(tuple(x.shape))
        assert training and abs(p - 0.1) < 1e-9
        if n == 0:
            site = D.SITE_EMBEDDINGS
        else:
            layer, which = divmod(n - 1, 3)
            site = (D.site_attention_probs, D.site_attention_output, D.site_ffn_output)[which](layer)
        if x.dim() == 4:
            assert x.shape == (B, heads, S, S)
            keep = torch.from_numpy(D.attention_mask(B, heads, S, 0.1, seed, site))
        else:
            keep = torch.from_numpy(D.hidden_mask(B * S, H, 0.1, seed, site)).view(B, S, H)
        return x * keep.to(x.dtype) * (1.0 / (1.0 - float(torch.tensor(0.1, dtype=torch.float32))))

    monkeypatch.setattr(torch.nn.functional, "dropout", masked_dropout)
    with torch.no_grad():
        ref = hf(input_ids=ids, attention_mask=mask, token_type_ids=torch.zeros_like(ids))["last_hidden_state"]
    monkeypatch.undo()
    assert len(calls) == 1 + 3 * cfg.num_hidden_layers, calls
    with torch.no_grad():
        got = E.bert_forward(mine.state_dict(), ids, mask, None, heads=heads, dropout=(0.1, 0.1, seed))
        plain = E.bert_forward(mine.state_dict(), ids, mask, None, heads=heads)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)
    assert float((got - plain).abs().max()) > 0.1            # the masks do act
    # mask statistics and determinism of the generator itself
    m = D.hidden_mask(4096, 768, 0.1, seed, 7)
    assert abs(m.mean() - 0.9) < 2e-3 and (m == D.hidden_mask(4096, 768, 0.1, seed, 7)).all()
    assert (m != D.hidden_mask(4096, 768, 0.1, seed, 8)).mean() > 0.1 and (m != D.hidden_mask(4096, 768, 0.1, seed + 1, 7)).mean() > 0.1
    assert D.hidden_mask(64, 768, 0.0, seed, 1).all()


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


# ---- pins: oracle vs outputs of transformers' ConvNextModel / ViTModel / ResNetModel / BertModel (tests/golden/g5_*, g9_*) ----
import os                                                            # noqa: E402
import sys                                                           # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import recipes as R                                                  # noqa: E402


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _close(got, want, rtol, what=""):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)
    assert err <= rtol, f"{what}: max error {err:.3e} of the largest magnitude (allowed {rtol:.1e})"


@pytest.mark.parametrize("name", ["g5_convnext_tiny.npz", "g5_convnext_small.npz"])
def test_convnext_oracle_matches_third_party_golden(golden_dir, name):
    """forward (`features`, `avgpool`) and parameter gradients of oracle.convnext_forward == transformers ConvNextModel on the
    recipe weights, incl. the 3-channel odd-size net (77x50 -> stem floor(./4) -> ... the torch floor rule everywhere)."""
    from mmgclip.networks import convnext as CN
    g = _g(golden_dir, name)
    seed, depths, dims = int(g["seed"]), tuple(int(v) for v in g["depths"]), tuple(int(v) for v in g["dims"])
    n, cin, H, W = (int(v) for v in g["shape"])
    CN.CONFIGS["_golden"] = dict(depths=depths, dims=dims)
    feats = R.fill_(CN.build_features("_golden", cin), seed, "features.")
    sd = {"features." + k: v for k, v in feats.named_parameters()}
    img = R.structured_images(n, max(H, W), seed, cin)[:, :, :H, :W].contiguous()
    assert abs(float(img.double().sum()) - float(g["image_sum"])) < 1e-6 * float(g["image_sum"])      # the recipe reproduced
    pooled, fmap = E.convnext_forward(sd, img, depths=depths)
    assert tuple(fmap.shape) == tuple(g["fmap_shape"])
    _close(pooled.flatten(1).detach(), g["pooled"], 2e-5, "pooled")
    _close(fmap[:, :, 0, 0].detach(), g["fmap_first"], 2e-5, "fmap[0,0]")
    _close(fmap[:, :, -1, -1].detach(), g["fmap_last"], 2e-5, "fmap[-1,-1]")
    (pooled.flatten(1) * torch.from_numpy(g["gy"])).sum().backward()
    keys = [k[5:] for k in g.files if k.startswith("grad.")]
    assert len(keys) > 20
    for k in keys:
        _close(sd[k].grad, g["grad." + k], 2e-4, k)


def test_vit_oracle_matches_third_party_golden(golden_dir):
    from mmgclip.networks.vit import _tv_layout
    g = _g(golden_dir, "g5_vit_b16.npz")
    seed, size = int(g["seed"]), int(g["image_size"])
    tv = R.fill_(_tv_layout(size, 1, 768, 12, 3072, 16), seed)
    sd = dict(tv.named_parameters())
    img = R.structured_images(3, size, seed)
    cls = E.vit_forward(sd, img)
    _close(cls.detach(), g["cls"], 2e-5, "class token")
    (cls * torch.from_numpy(g["gy"])).sum().backward()
    for k in [k[5:] for k in g.files if k.startswith("grad.")]:
        _close(sd[k].grad.reshape(g["grad." + k].shape), g["grad." + k], 2e-4, k)


def test_resnet_oracle_matches_third_party_golden(golden_dir):
    from mmgclip.networks.resnet import _TorchvisionResNet
    g = _g(golden_dir, "g5_resnet50.npz")
    seed = int(g["seed"])
    sd = R.fill_(_TorchvisionResNet(), seed).state_dict()
    img = R.structured_images(4, 64, seed, in_chans=3) * 2.0 - 1.0
    with torch.no_grad():
        _close(E.resnet50_forward(sd, img, train_bn=False), g["pooled_eval"], 5e-5, "eval-mode batch norm")
        _close(E.resnet50_forward(sd, img, train_bn=True), g["pooled_train"], 5e-4, "train-mode batch norm")


def test_bert_oracle_matches_third_party_golden(golden_dir):
    """BERT-base (12 layers, vocab 28996) on ragged [4,77] and [2,256] batches, hidden state at [SEP] and [CLS]."""
    from mmgclip.networks.bert import BertConfigLite, _hf_layout
    g = _g(golden_dir, "g5_bert_base.npz")
    sd = R.fill_(_hf_layout(BertConfigLite()), int(g["seed"])).state_dict()
    for n, S in ((4, 77), (2, 256)):
        ids, mask, tt = R.ragged_tokens(n, S, int(g["seed"]))
        assert (ids.numpy() == g[f"ids_{S}"]).all() and (mask.numpy() == g[f"mask_{S}"]).all()
        with torch.no_grad():
            h = E.bert_forward(sd, ids, mask, tt)
        from oracle import clip_oracle as O
        _close(O.eos_pool(h, mask), g[f"eos_hidden_{S}"], 5e-5, f"[SEP] rows S={S}")
        _close(h[:, 0], g[f"cls_hidden_{S}"], 5e-5, f"[CLS] rows S={S}")


def c1_oracle_step(g, dtype=torch.float32):
    """BASELINE config C1 through the oracle, on the recipe weights / inputs of tests/golden/g9_c1_step_s*.npz.
    Returns (dict of outputs, dict name -> parameter with .grad)."""
    from mmgclip.networks import convnext as CN
    from mmgclip.networks.bert import BertConfigLite, _hf_layout
    from oracle import clip_oracle as O
    seed, S = int(g["seed"]), int(g["S"])
    csd = {"features." + k: v for k, v in R.fill_(CN.build_features("tiny", 1), seed, "features.").named_parameters()}
    bsd = dict(R.fill_(_hf_layout(BertConfigLite()), seed + 1).named_parameters())
    wi = R.seeded_tensor("image_projection_layer.layer.weight", (512, 768), seed + 2).requires_grad_(True)
    wt = R.seeded_tensor("text_projection_layer.layer.weight", (512, 768), seed + 2).requires_grad_(True)
    ls = torch.tensor(float(np.log(1 / 0.07)), requires_grad=True)
    img = R.structured_images(8, 224, seed)
    ids, mask, tt = R.ragged_tokens(8, S, seed)
    assert (ids.numpy() == g["ids"]).all()
    pooled, _ = E.convnext_forward(csd, img)
    pooled = pooled.flatten(1)
    pooled.retain_grad()
    tf = O.eos_pool(E.bert_forward(bsd, ids, mask, tt), mask)
    tf.retain_grad()
    out = O.forward_tail(O.linear_projection(pooled, wi), O.linear_projection(tf, wt), ls)
    loss, _ = O.clip_loss(out["logits_per_image"], out["logits_per_text"])
    loss.backward()
    out.update(loss=loss, pooled=pooled, text_features=tf)
    params = {"image." + k: v for k, v in csd.items()}
    params.update({"text." + k: v for k, v in bsd.items()})
    params.update(wi=wi, wt=wt, ls=ls)
    return out, params


@pytest.mark.parametrize("S", [77, 256])
def test_c1_step_oracle_matches_third_party_and_reference_golden(golden_dir, S):
    """SURVEY §8c / BASELINE C1: n = 8, 224x224, ConvNeXt-T + BERT-base + LinearProjectionLayer + CLIPLoss, forward and backward.
    Golden = transformers towers + the reference's own projection / loss classes (tests/golden/make_golden_encoders.py)."""
    g = _g(golden_dir, f"g9_c1_step_s{S}.npz")
    out, params = c1_oracle_step(g)
    _close(out["pooled"].detach(), g["pooled"], 2e-5, "image features")
    _close(out["text_features"].detach(), g["text_features"], 5e-5, "text features")
    _close(out["logits_per_image"].detach(), g["logits_per_image"], 2e-5, "logits_per_image")
    _close(out["logits_per_text"].detach(), g["logits_per_text"], 2e-5, "logits_per_text")
    assert abs(float(out["loss"]) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    assert np.ptp(g["logits_per_image"]) > 1.0                       # the structured inputs give the logits real spread
    _close(out["pooled"].grad, g["d_pooled"], 2e-4, "d image features")
    _close(out["text_features"].grad, g["d_text_features"], 2e-4, "d text features")
    _close(params["ls"].grad, g["d_logit_scale"], 2e-4, "d logit_scale")
    _close(params["wi"].grad[:16], g["d_image_projection_rows"], 2e-4, "d image projection")
    _close(params["wt"].grad[:16], g["d_text_projection_rows"], 2e-4, "d text projection")
    keys = [k[5:] for k in g.files if k.startswith("grad.") and not k.endswith(".rows")]
    assert len(keys) > 100
    for k in keys:
        _close(params[k].grad, g["grad." + k], 5e-4, k)
    rows = torch.from_numpy(g["word_rows"])
    _close(params["text.embeddings.word_embeddings.weight"].grad[rows], g["grad.text.embeddings.word_embeddings.weight.rows"], 5e-4, "word rows")
