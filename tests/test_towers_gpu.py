"""Encoder towers on the HIP kernels vs the fp32 CPU oracle (same state dict, same inputs).

Tolerances: activations/weights are bf16 on the device (8 bits of mantissa), the oracle is fp32.  The bars are ~1.5x what a run
measures (profiles/r02_measured_tolerances.jsonl, written by tests.conftest.measured): ConvNeXt features 0.6-1.0e-2 relative
(bar 1.5e-2), its parameter gradients <= 2.9e-2 / cosine >= 0.9996 (bars 4e-2 / 0.999); BERT hidden states 7e-3 (bar 1.5e-2),
gradients <= 8.2e-2 / >= 0.9966 (bars 0.12 / 0.995: the worst ones are the near-zero key projections)."""
import os

import numpy as np
import pytest
import torch

from oracle import encoders_oracle as E
from tests.conftest import measured

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().float().cpu().double().flatten(), b.detach().float().cpu().double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30)), float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))


def _randomize(module, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in module.named_parameters():
            if n.endswith("layer_scale"):
                p.copy_(0.3 + 0.7 * torch.rand(p.shape, generator=g))
            elif n.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif "LayerNorm.weight" in n or (p.dim() == 1 and n.endswith("weight")):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif p.dim() >= 2 and "embeddings" not in n:
                p.mul_(2.5)


@pytest.mark.parametrize("size,n,variant", [(64, 3, "tiny"), (96, 2, "tiny"), (64, 2, "base")])
def test_convnext_tower_forward_backward(dev, size, n, variant):
    """tiny: fused CNBlock kernels at C = 96 / 192 / 384 (+ GEMM pair at 768); base: C = 128 / 256 fused, 512 / 1024 GEMM pair."""
    from mmgclip.networks.encoder import ConvNextBaseEncoder, ConvNextTinyEncoder
    torch.manual_seed(0)
    tower = (ConvNextTinyEncoder if variant == "tiny" else ConvNextBaseEncoder)(micro_batch=2)
    _randomize(tower, 1)
    sd = {k[len("model."):]: v.clone() for k, v in tower.state_dict().items()}
    img = torch.rand(n, 1, size, size, generator=torch.Generator().manual_seed(2))
    wgt = torch.randn(n, tower.model_output_dimension, generator=torch.Generator().manual_seed(3))
    # oracle
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    pooled, _ = E.convnext_forward(osd, img, depths=(3, 3, 9, 3) if variant == "tiny" else (3, 3, 27, 3))
    (pooled.flatten(1) * wgt).sum().backward()
    # device
    tower = tower.to(dev)
    feat = tower(img.to(dev))
    r, c = _rel(feat, pooled.flatten(1))
    fr, fc = r, c
    assert r < 1.5e-2 and c > 0.9999, (r, c)
    (feat * wgt.to(dev)).sum().backward()
    worst = {}
    for name, p in tower.model.named_parameters():
        r, c = _rel(p.grad, osd[name].grad)
        worst[name] = (r, c)
    measured("convnext_tower_forward_backward", variant=variant, size=size, n=n, feat_rel=fr, feat_cos=fc,
             grad_rel_max=max(v[0] for v in worst.values()), grad_cos_min=min(v[1] for v in worst.values()))
    bad = {k: v for k, v in worst.items() if not (v[1] > 0.999 and v[0] < 4e-2)}
    assert not bad, f"{len(bad)} of {len(worst)} gradients off: {list(bad.items())[:8]}"


def test_convnext_stage1_on_chip_weight_gradient_backward_equals_the_gemm_path(dev, monkeypatch):
    """Round 3: the stage-1 blocks' backward runs on mmg_cnblock_bwdw (weight gradients accumulated on chip).  The same tower with
    MMG_BWDW=0 takes the round-2 path (fused data-path kernel + two weight-gradient GEMMs); every parameter gradient of the two runs must
    agree to bf16 rounding noise (training sizes always qualify: H, W multiples of 32 make the stage-1 row count a multiple of 64)."""
    from mmgclip.networks.encoder import ConvNextTinyEncoder
    img = torch.rand(3, 1, 96, 64, generator=torch.Generator().manual_seed(2))
    wgt = torch.randn(3, 768, generator=torch.Generator().manual_seed(3))
    grads = {}
    for knob in ("1", "0"):
        monkeypatch.setenv("MMG_BWDW", knob)
        torch.manual_seed(0)
        tower = ConvNextTinyEncoder(micro_batch=2)
        _randomize(tower, 1)
        tower = tower.to(dev)
        assert tower.bwdw == (knob == "1")
        feat = tower(img.to(dev))
        (feat * wgt.to(dev)).sum().backward()
        used = any(k.endswith(".bwdw") for k in tower._wc)
        grads[knob] = ({n: p.grad.detach().float().cpu().clone() for n, p in tower.model.named_parameters()}, used)
    assert grads["1"][1] and not grads["0"][1]                     # the knob really selects the path
    worst = max((_rel(grads["1"][0][n], g0) + (n,) for n, g0 in grads["0"][0].items()), key=lambda t: t[0])
    assert worst[0] < 2e-2 and min(_rel(grads["1"][0][n], g0)[1] for n, g0 in grads["0"][0].items()) > 0.9995, worst


def test_convnext_backward_on_the_saved_activation_equals_the_rebuilt_one(dev, monkeypatch):
    """Round 3: the stage-3/4 blocks (GEMM-pair backward) keep GELU(hidden) from the forward (mmg_cnblock_mlp_fwd's gact output / the first
    GEMM's own output) and their data-gradient GEMM runs epilogue 5 (GELU' only).  MMG_SAVE_GELU=0 is the round-2 form (epilogue 2 rebuilds
    the activation from the saved pre-activation): same forward bits, every parameter gradient equal to bf16 rounding noise."""
    from mmgclip.networks.encoder import ConvNextTinyEncoder
    img = torch.rand(3, 1, 64, 96, generator=torch.Generator().manual_seed(4))
    wgt = torch.randn(3, 768, generator=torch.Generator().manual_seed(5))
    runs = {}
    for knob in ("1", "0"):
        monkeypatch.setenv("MMG_SAVE_GELU", knob)
        torch.manual_seed(0)
        tower = ConvNextTinyEncoder(micro_batch=2)
        _randomize(tower, 1)
        tower = tower.to(dev)
        feat = tower(img.to(dev))
        assert tower.save_gelu == (knob == "1")
        (feat * wgt.to(dev)).sum().backward()
        runs[knob] = (feat.detach().float().cpu(), {n: p.grad.detach().float().cpu().clone() for n, p in tower.model.named_parameters()})
    assert torch.equal(runs["1"][0], runs["0"][0])
    worst = max((_rel(runs["1"][1][n], g0) + (n,) for n, g0 in runs["0"][1].items()), key=lambda t: t[0])
    assert worst[0] < 2e-2 and min(_rel(runs["1"][1][n], g0)[1] for n, g0 in runs["0"][1].items()) > 0.9995, worst


@pytest.mark.parametrize("variant,min_c", [("base", 1024), ("tiny", 768), ("base", 128), ("tiny", 384)])
def test_convnext_fp8_forward_matches_the_fp8_oracle(dev, variant, min_c, monkeypatch):
    """BASELINE config C5: e4m3 forward GEMMs (in every block with C % 128 == 0 and C >= min_c) against the oracle that rounds
    the same tensors to e4m3; the backward (bf16, straight through the rounding) against the oracle's STE gradients.
    An e4m3 step is 6-12 % of a value, so wherever the device's bf16 activations differ from the oracle's fp32 ones by a
    fraction of a percent some elements round to the other neighbour: per block that is a ~2 % perturbation, and it compounds
    through the stages.  Last-stage-only cases are therefore held to the bf16 tolerances; with e4m3 in (almost) every block the
    device must still be nearer to the e4m3 oracle than to the fp32 one.  Bit-level agreement is pinned block by block on
    identical inputs in tests/test_kernels_gpu.py::test_cnblock_fp8_forward_one_block."""
    from mmgclip.networks.encoder import ConvNextBaseEncoder, ConvNextTinyEncoder
    monkeypatch.setenv("MMG_FP8_BWD", "0")          # (this test: the bf16 backward of rounds 1 - 3; the 8-bit backward has its own below)
    torch.manual_seed(0)
    tower = (ConvNextTinyEncoder if variant == "tiny" else ConvNextBaseEncoder)(micro_batch=2, fp8=True)
    tower.fp8_min_channels = min_c
    last_stage_only = min_c == tower.dims[-1]
    _randomize(tower, 1)
    depths = (3, 3, 9, 3) if variant == "tiny" else (3, 3, 27, 3)
    sd = {k[len("model."):]: v.clone() for k, v in tower.state_dict().items()}
    img = torch.rand(2, 1, 64, 64, generator=torch.Generator().manual_seed(2))
    wgt = torch.randn(2, tower.model_output_dimension, generator=torch.Generator().manual_seed(3))
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    pooled, _ = E.convnext_forward(osd, img, depths=depths, fp8_min_channels=min_c)
    (pooled.flatten(1) * wgt).sum().backward()
    with torch.no_grad():
        pooled32, _ = E.convnext_forward(sd, img, depths=depths)
    tower = tower.to(dev)
    feat = tower(img.to(dev))
    n_fp8 = sum(k.endswith(".w1f8") for k in tower._wc)
    assert n_fp8 == sum(n for n, c in zip(depths, tower.dims) if c % 128 == 0 and c >= min_c)
    r8, c8 = _rel(feat, pooled.flatten(1))
    r32, _ = _rel(feat, pooled32.flatten(1))
    q_effect, _ = _rel(pooled.flatten(1), pooled32.flatten(1))
    print("fp8 tower", variant, min_c, "dev-vs-fp8-oracle", r8, c8, "dev-vs-fp32-oracle", r32, "oracle fp8-vs-fp32", q_effect)
    assert q_effect > 5e-3 and r8 < r32, (r8, r32, q_effect)       # the rounding is visible and the device follows it
    if last_stage_only:
        assert r8 < 4e-2 and c8 > 0.999, (r8, c8)
    else:
        assert r8 < 0.1 and c8 > 0.995, (r8, c8)
    (feat * wgt.to(dev)).sum().backward()
    bad, allg = {}, {}
    for name, p in tower.model.named_parameters():
        r, c = _rel(p.grad, osd[name].grad)
        allg[name] = (r, c)
        if not ((c > 0.995 and r < 0.11) if last_stage_only else (c > 0.975 and r < 0.25)):      # measured 0.997 / 0.074 and 0.984 / 0.18
            bad[name] = (r, c)
    measured("convnext_fp8_forward", variant=variant, min_c=min_c, feat_rel=r8, feat_cos=c8,
             grad_rel_max=max(v[0] for v in allg.values()), grad_cos_min=min(v[1] for v in allg.values()))
    assert not bad, f"{len(bad)} gradients off: {list(bad.items())[:8]}"


@pytest.mark.parametrize("variant,min_c", [("base", 1024), ("tiny", 384), ("base", 128)])
def test_convnext_fp8_backward_matches_the_fp8_oracle(dev, variant, min_c):
    """Round 4 (VERDICT r3 missing #2): the same blocks' BACKWARD in 8 bits - the incoming gradient cast to e5m2 with a per-tensor power-of-two
    scale, both data-gradient GEMMs on e5m2 x e4m3 operands (dh handed on in e5m2, written once), both weight-gradient GEMMs on the 8-bit operands
    (csrc/gemm_tn_fp8.hip) - against the oracle that rounds the same tensors at the same places (oracle.encoders_oracle.Fp8BlockMLP).  An e5m2 step
    is 25 % of a value: element by element the device's gradient bytes and the oracle's flip across rounding boundaries wherever their bf16 / fp32
    inputs differ by a fraction of a percent (the device's gradients are as far from the 8-bit oracle as that oracle is from the straight-through
    one: the rounding noise of two runs is uncorrelated); what must agree are the sums the GEMMs form."""
    from mmgclip.networks.encoder import ConvNextBaseEncoder, ConvNextTinyEncoder
    torch.manual_seed(0)
    tower = (ConvNextTinyEncoder if variant == "tiny" else ConvNextBaseEncoder)(micro_batch=2, fp8=True)
    tower.fp8_min_channels = min_c
    assert tower.fp8_bwd
    last_stage_only = min_c == tower.dims[-1]
    _randomize(tower, 1)
    depths = (3, 3, 9, 3) if variant == "tiny" else (3, 3, 27, 3)
    sd = {k[len("model."):]: v.clone() for k, v in tower.state_dict().items()}
    img = torch.rand(2, 1, 64, 64, generator=torch.Generator().manual_seed(2))
    wgt = torch.randn(2, tower.model_output_dimension, generator=torch.Generator().manual_seed(3))
    grads = {}
    for fb in (True, False):
        osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        pooled, _ = E.convnext_forward(osd, img, depths=depths, fp8_min_channels=min_c, fp8_backward=fb)
        (pooled.flatten(1) * wgt).sum().backward()
        grads[fb] = {k: v.grad for k, v in osd.items()}
    tower = tower.to(dev)
    feat = tower(img.to(dev))
    assert sum(k.endswith(".w2gt8") for k in tower._wc) == sum(n for n, c in zip(depths, tower.dims) if c % 128 == 0 and c >= min_c)
    r8, c8 = _rel(feat, pooled.flatten(1))
    (feat * wgt.to(dev)).sum().backward()
    bad, allg, nearer = {}, {}, 0
    for name, p in tower.model.named_parameters():
        r, c = _rel(p.grad, grads[True][name])
        r_ste, _ = _rel(p.grad, grads[False][name])
        allg[name] = (r, c)
        nearer += r <= r_ste
        # measured (profiles/r04_measured_tolerances.jsonl): last stage only 0.993 / 0.12; stages 3 - 4 of ConvNeXt-T 0.982 / 0.19; all 36 blocks of
        # ConvNeXt-B 0.956 / 0.30 - the size of the 8-bit oracle's own distance from the straight-through one (0.10 / 0.19 / 0.21)
        if not ((c > 0.99 and r < 0.15) if last_stage_only else (c > 0.94 and r < 0.35)):
            bad[name] = (r, c)
    q_effect = max(_rel(grads[True][n], grads[False][n])[0] for n in grads[True])
    print("fp8 backward", variant, min_c, "grad rel max", max(v[0] for v in allg.values()), "cos min", min(v[1] for v in allg.values()),
          "nearer to the 8-bit oracle than to the STE one:", nearer, "of", len(allg), "oracle 8-bit vs STE rel max", q_effect)
    measured("convnext_fp8_backward", variant=variant, min_c=min_c, feat_rel=r8, grad_rel_max=max(v[0] for v in allg.values()),
             grad_cos_min=min(v[1] for v in allg.values()), nearer_8bit_oracle=int(nearer), n_tensors=len(allg), oracle_8bit_vs_ste_rel_max=q_effect)
    assert not bad, f"{len(bad)} gradients off: {list(bad.items())[:8]}"


def test_convnext_frozen_makes_no_graph(dev):
    from mmgclip.networks.encoder import ConvNextTinyEncoder
    tower = ConvNextTinyEncoder(freeze=True).to(dev)
    out = tower(torch.rand(1, 1, 64, 64, device=dev))
    assert not out.requires_grad and out.shape == (1, 768)


def test_bert_tower_forward_backward(dev):
    from mmgclip.networks.bert import BertConfigLite
    from mmgclip.networks.encoder import BertEncoder
    from mmgclip.dataset.synthetic import synthetic_tokens
    torch.manual_seed(0)
    cfg = BertConfigLite(vocab_size=3000, num_hidden_layers=3)
    enc = BertEncoder(pretrained=None, random_init=True, freeze=False, config=cfg)
    _randomize(enc, 4)
    sd = {k[len("model."):]: v.clone() for k, v in enc.state_dict().items()}
    tok = synthetic_tokens(4, 77, 3000, torch.Generator().manual_seed(5))
    wgt = torch.randn(4 * 77, 768, generator=torch.Generator().manual_seed(6))
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = E.bert_forward(osd, tok["input_ids"], tok["attention_mask"], tok["token_type_ids"])
    valid = tok["attention_mask"].reshape(-1, 1).float()          # padded query rows are don't-care downstream
    (ref.reshape(-1, 768) * wgt * valid).sum().backward()
    enc = enc.to(dev)
    hid = enc.hidden_states({k: v.to(dev) for k, v in tok.items()})
    r, c = _rel(hid.float() * valid.to(dev), ref.reshape(-1, 768) * valid)
    hr, hc = r, c
    assert r < 1.5e-2 and c > 0.9999, (r, c)
    (hid.float() * (wgt * valid).to(dev)).sum().backward()
    bad, allg = {}, {}
    for name, p in enc.model.named_parameters():
        if name.startswith("pooler."):
            continue
        if name.endswith("attention.self.key.bias"):
            # softmax is invariant to a shift of all keys: the exact gradient is 0, both sides only hold rounding noise
            qb = enc.model.get_parameter(name.replace("key.bias", "query.bias")).grad
            assert p.grad.abs().max() < 0.25 * qb.abs().max()
            continue
        r, c = _rel(p.grad, osd[name].grad)
        allg[name] = (r, c)
        if not (c > 0.995 and r < 0.12):
            bad[name] = (r, c)
    measured("bert_tower_forward_backward", hidden_rel=hr, hidden_cos=hc, grad_rel_max=max(v[0] for v in allg.values()),
             grad_cos_min=min(v[1] for v in allg.values()), worst=max(allg, key=lambda k: allg[k][0]))
    assert not bad, f"{len(bad)} gradients off: {list(bad.items())[:8]}"


def test_bert_encoder_api_and_eos_pool(dev):
    """BertEncoder.forward returns [B,S,H] like encoder.py:156; frozen by default (encoder.py:141-142)."""
    from mmgclip.networks.bert import BertConfigLite, EosPool
    from mmgclip.networks.encoder import BertEncoder
    from mmgclip.dataset.synthetic import synthetic_tokens
    from oracle import clip_oracle as O
    enc = BertEncoder(pretrained="emilyalsentzer/Bio_ClinicalBERT", random_init=True,
                      config=BertConfigLite(vocab_size=3000, num_hidden_layers=2))
    assert all(not p.requires_grad for p in enc.parameters()) and enc.model_output_dimension == 768
    sd = {k[len("model."):]: v.clone() for k, v in enc.state_dict().items()}
    tok = synthetic_tokens(5, 40, 3000, torch.Generator().manual_seed(7))
    enc = enc.to(dev)
    dtok = {k: v.to(dev) for k, v in tok.items()}
    out = enc(dtok)
    assert out.shape == (5, 40, 768) and out.dtype == torch.float32
    ref = E.bert_forward(sd, tok["input_ids"], tok["attention_mask"], tok["token_type_ids"])
    pooled = EosPool.apply(enc.hidden_states(dtok), dtok["attention_mask"], 5, 40)
    rp = O.eos_pool(ref, tok["attention_mask"])
    r, c = _rel(pooled, rp)
    assert r < 3e-2 and c > 0.999
    with pytest.raises(OSError):
        BertEncoder(pretrained="emilyalsentzer/Bio_ClinicalBERT")


def test_vit_tower_forward_backward(dev):
    from mmgclip.networks.encoder import ViTB16Encoder
    torch.manual_seed(0)
    tower = ViTB16Encoder(image_size=96, layers=3, micro_batch=2)          # S = 37 tokens
    _randomize(tower, 8)
    with torch.no_grad():
        tower.model.class_token.normal_(std=0.5)
    sd = {k[len("model."):]: v.clone() for k, v in tower.state_dict().items()}
    img = torch.rand(3, 1, 96, 96, generator=torch.Generator().manual_seed(9))
    wgt = torch.randn(3, 768, generator=torch.Generator().manual_seed(10))
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = E.vit_forward(osd, img)
    (ref * wgt).sum().backward()
    tower = tower.to(dev)
    feat = tower(img.to(dev))
    r, c = _rel(feat, ref)
    assert r < 3e-2 and c > 0.999, (r, c)
    (feat * wgt.to(dev)).sum().backward()
    bad = {}
    for name, p in tower.model.named_parameters():
        if name.endswith("in_proj_bias"):
            continue        # its key third has an exactly-zero gradient (softmax shift invariance): noise only
        r, c = _rel(p.grad, osd[name].grad)
        if not (c > 0.99 and r < 0.12):
            bad[name] = (r, c)
    assert not bad, f"{len(bad)} gradients off: {list(bad.items())[:8]}"


def test_vit_gradient_checkpointing_equals_plain_backward(dev):
    """checkpoint=True keeps only the pixels of a micro-batch and re-runs its forward in the backward: same features and
    (deterministic kernels, one writer per row) the same gradients as the saving path."""
    from mmgclip.networks.encoder import ViTB16Encoder
    img = torch.rand(4, 1, 96, 96, generator=torch.Generator().manual_seed(1)).to(dev)
    wgt = torch.randn(4, 768, generator=torch.Generator().manual_seed(2)).to(dev)
    ref = ViTB16Encoder(image_size=96, layers=2, micro_batch=2)
    _randomize(ref, 3)
    out = {}
    for ck in (False, True):
        tower = ViTB16Encoder(image_size=96, layers=2, micro_batch=2, checkpoint=ck)
        tower.load_state_dict(ref.state_dict())
        tower = tower.to(dev)
        feat = tower(img)
        (feat * wgt).sum().backward()
        out[ck] = (feat.detach().clone(), {n: p.grad.detach().clone() for n, p in tower.model.named_parameters()})
    assert torch.equal(out[True][0], out[False][0])
    for n, gr in out[False][1].items():
        rel = float((out[True][1][n] - gr).norm() / (gr.norm() + 1e-30))
        assert rel < 1e-5, (n, rel)          # weight-gradient GEMMs accumulate with fp32 atomics: order noise only


def test_vit_tower_long_sequence_path(dev):
    """S = (272/16)^2 + 1 = 290 > 256 tokens: the tiled attention kernels carry the forward and the backward."""
    from mmgclip.networks.encoder import ViTB16Encoder
    torch.manual_seed(0)
    tower = ViTB16Encoder(image_size=272, layers=2, micro_batch=2)
    _randomize(tower, 12)
    sd = {k[len("model."):]: v.clone() for k, v in tower.state_dict().items()}
    img = torch.rand(2, 1, 272, 272, generator=torch.Generator().manual_seed(13))
    wgt = torch.randn(2, 768, generator=torch.Generator().manual_seed(14))
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = E.vit_forward(osd, img)
    (ref * wgt).sum().backward()
    tower = tower.to(dev)
    feat = tower(img.to(dev))
    r, c = _rel(feat, ref)
    assert r < 3e-2 and c > 0.999, (r, c)
    (feat * wgt.to(dev)).sum().backward()
    for name in ("conv_proj.weight", "encoder.layers.encoder_layer_0.self_attention.in_proj_weight",
                 "encoder.layers.encoder_layer_1.mlp.0.weight", "encoder.pos_embedding"):
        r, c = _rel(tower.model.get_parameter(name).grad, osd[name].grad)
        assert c > 0.99 and r < 0.12, (name, r, c)


def test_feature_extractor_full_resolution_geometry_and_file_format(dev, tmp_path):
    """Offline extraction (reference image_features.py:85-117): odd full-resolution sizes floor like torch's strided
    convolutions ([1,1,H,W] -> [1,768,H/32,W/32] -> [1,768,1,1], cf. 1906x818 -> 59x25 in the notebook), HF/torchvision
    layout state dict in, one `.pth` [1,768,1,1] fp32 file per image out."""
    import pandas as pd
    from PIL import Image
    from mmgclip.config import Config
    from mmgclip.networks.convnext import build_features
    from mmgclip.networks.image_features import ImageFeatureExtractor, load_image
    torch.manual_seed(0)
    feats = build_features("tiny", in_chans=1)
    _randomize(feats, 15)
    sd = {"features." + k: v for k, v in feats.state_dict().items()}
    ckpt = tmp_path / "convnext_tiny.pth"
    torch.save(sd, ckpt)
    h, w = 190, 117                                             # deliberately not multiples of 4 / 32
    rng = np.random.default_rng(0)
    paths = []
    for i in range(2):
        arr = (rng.random((h, w)) * 65535).astype(np.uint16)
        p = tmp_path / "2D_100micron" / "0" / f"p{i}" / f"img{i}.png"
        p.parent.mkdir(parents=True, exist_ok=True)
        Image.fromarray(arr).save(p)
        paths.append(str(p))
    cfg = Config.wrap({"networks": {"image_encoder": {"convnext_tiny_clf_path": str(ckpt)}}, "base": {"features_export_dir": str(tmp_path / "out")}})
    ex = ImageFeatureExtractor(config=cfg, dataset=pd.DataFrame({"image_path": paths + [str(tmp_path / "missing.png")]}))
    assert ex.image_encoder._tower.feature_map_shape(1906, 818) == (59, 25)
    ex.extract()
    assert (tmp_path / "out" / "failed.txt").exists()           # the missing file is logged, not fatal
    for i, pth in enumerate(paths):
        f = torch.load(tmp_path / "out" / "0" / f"p{i}" / f"img{i}.pth")
        assert f.shape == (1, 768, 1, 1) and f.dtype == torch.float32
        x = load_image(pth).unsqueeze(0)
        ref, fmap = E.convnext_forward(sd, x, scale16=True)
        assert fmap.shape[-2:] == (h // 32, w // 32)
        r, c = _rel(f, ref)
        assert r < 3e-2 and c > 0.999, (r, c)


def test_study_feature_extractor_pools_views(dev, tmp_path):
    """StudyFeatureExtractor (reference image_features.py:187-263): <=n views per exam, maxpool / avgpool / stack."""
    import pandas as pd
    from PIL import Image
    from mmgclip.config import Config
    from mmgclip.networks.convnext import build_features
    from mmgclip.networks.image_features import StudyFeatureExtractor, load_image
    torch.manual_seed(1)
    feats = build_features("tiny", in_chans=1)
    _randomize(feats, 16)
    sd = {"features." + k: v for k, v in feats.state_dict().items()}
    ckpt = tmp_path / "convnext_tiny.pth"
    torch.save(sd, ckpt)
    rng = np.random.default_rng(1)
    study = tmp_path / "2D_100micron" / "0" / "12" / "01234567" / "st01"
    study.mkdir(parents=True)
    for i, (h, w) in enumerate([(96, 64), (96, 64), (70, 90)]):
        Image.fromarray((rng.random((h, w)) * 255).astype(np.uint8)).save(study / f"v{i}.png")
    names = os.listdir(study)[:4]
    ref = torch.stack([E.convnext_forward(sd, load_image(str(study / n)).unsqueeze(0), scale16=True)[0].reshape(-1) for n in names])
    for method, want in (("maxpool", ref.max(0)[0]), ("avgpool", ref.mean(0)), ("stack", ref)):
        cfg = Config.wrap({"networks": {"image_encoder": {"convnext_tiny_clf_path": str(ckpt)}},
                           "base": {"features_export_dir": str(tmp_path / method)},
                           "dataset": {"config": {"n_images_per_study": 4, "concatenate_features_method": method}}})
        StudyFeatureExtractor(config=cfg, dataset=pd.DataFrame({"study_path": [str(study)]})).extract()
        assert not (tmp_path / method / "failed.txt").exists(), (tmp_path / method / "failed.txt").read_text()
        got = torch.load(tmp_path / method / "0" / "12" / "01234567" / "st01" / "01234567.pth")
        assert got.shape == want.shape
        r, c = _rel(got, want)
        assert r < 3e-2 and c > 0.999, (method, r, c)


def test_bert_encoder_loads_huggingface_directory(dev, tmp_path):
    """a `BertModel.save_pretrained` directory (config.json + model.safetensors) drops into BertEncoder
    (reference encoder.py:131-156 calls AutoModel.from_pretrained on a name or local path); output = HF last_hidden_state."""
    transformers = pytest.importorskip("transformers")
    torch.manual_seed(3)
    hf_cfg = transformers.BertConfig(vocab_size=300, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                                     intermediate_size=512, max_position_embeddings=64, hidden_dropout_prob=0.0,
                                     attention_probs_dropout_prob=0.0)
    hf = transformers.BertModel(hf_cfg, add_pooling_layer=False).eval()
    hf.save_pretrained(tmp_path / "bert")
    from mmgclip.networks.encoder import BertEncoder
    enc = BertEncoder(pretrained=str(tmp_path / "bert")).to(dev)
    assert enc.model_output_dimension == 128
    B, S = 4, 24
    ids = torch.randint(1, 300, (B, S))
    lens = torch.tensor([24, 9, 17, 3])
    mask = (torch.arange(S)[None] < lens[:, None]).long()
    ids = ids * mask
    with torch.no_grad():
        want = hf(input_ids=ids, attention_mask=mask, token_type_ids=torch.zeros_like(ids)).last_hidden_state
        got = enc({"input_ids": ids.to(dev), "attention_mask": mask.to(dev), "token_type_ids": torch.zeros_like(ids).to(dev)}).float().cpu()
    valid = mask.bool()
    r, c = _rel(got[valid], want[valid])
    assert r < 3e-2 and c > 0.999, (r, c)


def test_bert_packed_layout_equals_padded(dev):
    """The unpadded path (valid tokens only, mmg_attention_varlen_*) gives the padded path's hidden states on every valid row
    and the same parameter gradients; its padding rows are zeros."""
    from mmgclip.networks.bert import BertConfigLite
    from mmgclip.networks.encoder import BertEncoder
    from mmgclip.dataset.synthetic import synthetic_tokens
    torch.manual_seed(0)
    cfg = BertConfigLite(vocab_size=3000, num_hidden_layers=2)
    enc = BertEncoder(pretrained=None, random_init=True, freeze=False, config=cfg).to(dev)
    _randomize(enc, 4)
    tok = {k: v.to(dev) for k, v in synthetic_tokens(6, 77, 3000, torch.Generator().manual_seed(5)).items()}
    valid = tok["attention_mask"].reshape(-1, 1).bool()
    wgt = torch.randn(6 * 77, 768, generator=torch.Generator().manual_seed(6)).to(dev)
    outs, grads = [], []
    for packed in (False, True):
        enc.zero_grad(set_to_none=True)
        h = enc.hidden_states(tok, packed=packed)
        (h.float() * wgt * valid).sum().backward()
        outs.append(h.float())
        grads.append({n: p.grad.clone() for n, p in enc.model.named_parameters() if p.grad is not None})
    assert float(outs[1][~valid.expand_as(outs[1])].abs().max()) == 0.0
    r, c = _rel(outs[1] * valid, outs[0] * valid)
    assert r < 1e-2 and c > 0.9999, (r, c)
    for n, g0 in grads[0].items():
        if n.endswith("attention.self.key.bias") or n.startswith("pooler."):
            continue
        r, c = _rel(grads[1][n], g0)
        assert c > 0.999 and r < 3e-2, (n, r, c)


@pytest.mark.parametrize("shape", [(6, 3, 128, 160), (16, 768)])
def test_resnet50_encoder_forward_eval_and_train(dev, shape):
    """ResNet50Encoder (reference encoder.py:57-119) vs the fp32 restatement of torchvision's resnet50.  Eval mode (running
    statistics) is compared tightly.  Train mode chains 53 batch-statistics normalisations on a randomly initialised net: the
    bf16 rounding of the stored activations grows by ~1.35x per block (measured layer by layer; eval mode does not amplify), so
    the composed tower is only checked for gross agreement there - the train-mode arithmetic and its backward are checked
    block by block in the next test.  2-D input = the reference's 1 x L three-channel view."""
    from mmgclip.networks.encoder import ResNet50Encoder
    torch.manual_seed(0)
    enc = ResNet50Encoder(pretrained=False)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for n_, b in enc.model.named_buffers():
            if n_.endswith("running_mean"):
                b.copy_(0.1 * torch.randn(b.shape, generator=g))
            elif n_.endswith("running_var"):
                b.copy_(0.5 + torch.rand(b.shape, generator=g))
    assert enc.model_output_dimension == 2048
    assert all(p.requires_grad == n_.startswith("layer4.") for n_, p in enc.model.named_parameters())
    sd = {k: v.clone() for k, v in enc.model.state_dict().items()}
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(2)) if len(shape) == 4 else \
        torch.randn(*shape, generator=torch.Generator().manual_seed(2)).abs()
    with torch.no_grad():
        ref_eval = E.resnet50_forward(sd, x, train_bn=False)
        ref_train = E.resnet50_forward(sd, x, train_bn=True, storage_bf16=True)
    enc = enc.to(dev)
    enc.eval()
    with torch.no_grad():
        out = enc(x.to(dev))
    assert out.shape == (shape[0], 2048) and not out.requires_grad
    r, c = _rel(out, ref_eval)
    assert r < 3e-2 and c > 0.999, ("eval", r, c)
    enc.train()
    feat = enc(x.to(dev))
    r, c = _rel(feat, ref_train)
    assert c > 0.97, ("train", r, c)
    feat.sum().backward()
    for n_, p in enc.model.named_parameters():
        assert (p.grad is not None) == n_.startswith("layer4."), n_
        assert p.grad is None or torch.isfinite(p.grad).all()
    # running statistics moved like torch's (momentum 0.1) in train mode, and only there
    assert not torch.allclose(enc.model.bn1.running_mean.cpu(), sd["bn1.running_mean"])
    assert int(enc.model.bn1.num_batches_tracked) == 1


@pytest.mark.parametrize("bi", [0, 1])
def test_resnet50_layer4_block_train_mode_forward_backward(dev, bi):
    """One layer4 bottleneck (bi = 0: stride 2 + down-sampling shortcut; bi = 1: identity shortcut) in training mode against fp32
    torch autograd of the same block on the same bf16 input: output, running statistics, every parameter gradient and (identity
    block) the gradient w.r.t. the input."""
    import torch.nn.functional as F
    from mmgclip.networks.encoder import ResNet50Encoder
    torch.manual_seed(0)
    enc = ResNet50Encoder(pretrained=False)
    blk = enc.model.layer4[bi]
    g = torch.Generator().manual_seed(3 + bi)
    with torch.no_grad():
        for p in blk.parameters():
            if p.dim() == 1:
                p.copy_(1.0 + 0.3 * torch.randn(p.shape, generator=g))
        for n_, p in blk.named_parameters():
            if n_.endswith("bias"):
                p.copy_(0.2 * torch.randn(p.shape, generator=g))
    n, H, W = 4, 8, 8
    cin, width, s = blk.conv1.in_channels, blk.conv1.out_channels, blk.stride
    x = torch.randn(n, cin, H, W, generator=g).abs().to(torch.bfloat16).float()
    Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
    dout = torch.randn(n, 4 * width, Ho, Wo, generator=g).to(torch.bfloat16).float()
    # fp32 reference (weights rounded to bf16 like the device operands)
    prm = {k: v.detach().clone().requires_grad_(True) for k, v in blk.named_parameters()}
    q = lambda t: t.to(torch.bfloat16).float()      # noqa: E731
    xr = x.clone().requires_grad_(True)

    def bnf(t, k):
        return F.batch_norm(t, None, None, prm[k + ".weight"], prm[k + ".bias"], True, 0.1, 1e-5)
    y = F.relu(bnf(F.conv2d(xr, prm["conv1.weight"] + (q(prm["conv1.weight"]) - prm["conv1.weight"]).detach()), "bn1"))
    y = F.relu(bnf(F.conv2d(y, prm["conv2.weight"] + (q(prm["conv2.weight"]) - prm["conv2.weight"]).detach(), stride=s, padding=1), "bn2"))
    y = bnf(F.conv2d(y, prm["conv3.weight"] + (q(prm["conv3.weight"]) - prm["conv3.weight"]).detach()), "bn3")
    if bi == 0:
        idn = bnf(F.conv2d(xr, prm["downsample.0.weight"] + (q(prm["downsample.0.weight"]) - prm["downsample.0.weight"]).detach(), stride=s),
                  "downsample.1")
    else:
        idn = xr
    ref = F.relu(y + idn)
    ref.backward(dout)
    # device
    enc = enc.to(dev).train()
    enc._materialize(dev)
    enc._refresh_working_copies()
    enc._arena.prepare_grads()
    rows = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).to(torch.bfloat16).to(dev).contiguous()    # noqa: E731
    out, ho, wo, sv = enc._block_fwd(rows(x), n, H, W, blk, f"3.{bi}.", True)
    assert (ho, wo) == (Ho, Wo)
    r, c = _rel(out, rows(ref.detach()))
    assert r < 2e-2 and c > 0.9995, ("out", r, c)
    dx = enc._block_bwd(rows(dout), sv, blk, f"3.{bi}.", need_dx=(bi > 0))
    if bi > 0:
        r, c = _rel(dx, rows(xr.grad))
        assert r < 0.12 and c > 0.995, ("dx", r, c)
    bad = {}
    for k, p in blk.named_parameters():
        r, c = _rel(p.grad, prm[k].grad)
        if not (c > 0.99 and r < 0.15):          # three bf16-stored gradient stages deep for conv1 / bn1
            bad[k] = (r, c)
    assert not bad, bad


def test_convnext_gradient_checkpointing_equals_plain_backward(dev, monkeypatch):
    """checkpoint=True keeps only the pixels and re-runs each micro-batch's forward before its backward: same features, same
    gradients (up to the order of fp32 atomics), activation memory of one micro-batch.  Round 4: the LAST micro-batch keeps its
    activations and its backward runs first (MMG_CKPT_KEEP_LAST, default on) - one recomputation less, the same peak."""
    from mmgclip.networks.encoder import ConvNextTinyEncoder
    torch.manual_seed(0)
    img = torch.rand(8, 1, 256, 256, generator=torch.Generator().manual_seed(1)).to(dev)
    wgt = torch.randn(8, 768, generator=torch.Generator().manual_seed(2)).to(dev)
    ref = ConvNextTinyEncoder(micro_batch=2)
    state = {k: v.clone() for k, v in ref.state_dict().items()}
    out = {}
    for ck in (False, "0", "1"):               # plain; checkpointed with every micro-batch recomputed; ... with the last one kept
        monkeypatch.setenv("MMG_CKPT_KEEP_LAST", ck or "1")
        tower = ConvNextTinyEncoder(micro_batch=2, checkpoint=bool(ck))
        tower.load_state_dict(state)
        tower = tower.to(dev)
        tower(img[:2]).sum().backward()                   # materialise arenas / working copies before measuring
        tower.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        feat = tower(img)
        held = torch.cuda.memory_allocated() - base         # activations kept between forward and backward
        (feat * wgt).sum().backward()
        torch.cuda.synchronize()
        out[ck] = (feat.detach().clone(), {n: p.grad.detach().clone() for n, p in tower.model.named_parameters()}, held,
                   torch.cuda.max_memory_allocated() - base)
        del tower, feat
    for ck in ("0", "1"):
        assert _rel(out[ck][0], out[False][0])[0] < 1e-5
        for n, g in out[ck][1].items():
            assert _rel(g, out[False][1][n])[0] < 2e-3, (ck, n)
    assert out["0"][2] < 0.25 * out[False][2], (out["0"][2], out[False][2])            # (pixels only)
    assert out["1"][2] < 0.30 * out[False][2], (out["1"][2], out[False][2])            # + one of the four micro-batches
    assert out["1"][3] <= 1.02 * out["0"][3], (out["1"][3], out["0"][3])               # the peak (one micro-batch in its backward) is the same
