#!/usr/bin/env python3
"""Encoder pins (G5, G9): outputs of THIRD-PARTY implementations of the towers, on seeded weights and inputs.

The reference's encoder arithmetic lives in torchvision 0.14.1 (ConvNeXt inside a TorchScript archive, ResNet-50) and
transformers 4.41.0 (BertModel) - neither in the reference tree nor, for torchvision, installed here.
The transformers package of this image (5.15) ships independent implementations of all four architectures:
`ConvNextModel`, `ViTModel`, `ResNetModel`, `BertModel`.  This script builds the weights with tests/golden/recipes.py in
the torchvision / HF layout the repo uses, maps them into those models, runs them in fp32 on the CPU (forward and
autograd backward) and stores inputs-by-seed + expected outputs:

    g5_convnext_tiny.npz   ConvNextModel, depths 3/3/9/3, dims 96..768, 1 input channel, 2 x 96 x 64 pixels
    g5_convnext_small.npz  a 2/2/2/2-deep 32..256 net on 3 channels with odd sizes (floor rule of the stride convolutions)
    g5_vit_b16.npz         ViTModel, 12 layers x 768, patch 16, 64 x 64 pixels (S = 17)
    g5_resnet50.npz        ResNetModel (bottleneck, v1.5 strides), train-mode and eval-mode batch norm
    g5_bert_base.npz       BertModel, 12 layers, [4,77] and [2,256] ragged batches: last_hidden_state at the [SEP] rows
    g9_c1_step_s77.npz / g9_c1_step_s256.npz
                           BASELINE config C1 end to end (n = 8, 224 x 224, ConvNeXt-T + BERT-base): third-party towers ->
                           the REFERENCE's own LinearProjectionLayer and CLIPLoss (loaded by file path like make_golden.py)
                           -> logits, loss and a gradient subset.

Run in the build container:  python tests/golden/make_golden.py   (calls this file)   or   python tests/golden/make_golden_encoders.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "mmg-clip_amd"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import recipes as R                                   # noqa: E402


def t2n(t):
    return t.detach().cpu().numpy()


# ---- weight mapping: torchvision layout (the repo's) -> transformers models ---------------------------------------------
def convnext_to_hf(sd, depths):
    """torchvision `features.*` keys -> transformers ConvNextModel keys."""
    out = {"embeddings.patch_embeddings.weight": sd["features.0.0.weight"], "embeddings.patch_embeddings.bias": sd["features.0.0.bias"],
           "embeddings.layernorm.weight": sd["features.0.1.weight"], "embeddings.layernorm.bias": sd["features.0.1.bias"]}
    for si in range(4):
        st = 1 + 2 * si
        for bi in range(depths[si]):
            a, b = f"features.{st}.{bi}.", f"encoder.stages.{si}.layers.{bi}."
            out[b + "layer_scale_parameter"] = sd[a + "layer_scale"].reshape(-1)
            for tv, hf in (("block.0", "dwconv"), ("block.2", "layernorm"), ("block.3", "pwconv1"), ("block.5", "pwconv2")):
                out[b + hf + ".weight"], out[b + hf + ".bias"] = sd[a + tv + ".weight"], sd[a + tv + ".bias"]
        if si > 0:
            a, b = f"features.{st - 1}.", f"encoder.stages.{si}.downsampling_layer."
            for j in (0, 1):
                out[b + f"{j}.weight"], out[b + f"{j}.bias"] = sd[a + f"{j}.weight"], sd[a + f"{j}.bias"]
    return out


def vit_to_hf(sd, layers, hidden):
    out = {"embeddings.cls_token": sd["class_token"], "embeddings.position_embeddings": sd["encoder.pos_embedding"],
           "embeddings.patch_embeddings.projection.weight": sd["conv_proj.weight"],
           "embeddings.patch_embeddings.projection.bias": sd["conv_proj.bias"],
           "layernorm.weight": sd["encoder.ln.weight"], "layernorm.bias": sd["encoder.ln.bias"]}
    for i in range(layers):
        a, b = f"encoder.layers.encoder_layer_{i}.", f"layers.{i}."
        w, bias = sd[a + "self_attention.in_proj_weight"], sd[a + "self_attention.in_proj_bias"]
        for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
            out[b + f"attention.{nm}.weight"] = w[j * hidden:(j + 1) * hidden]
            out[b + f"attention.{nm}.bias"] = bias[j * hidden:(j + 1) * hidden]
        out[b + "attention.o_proj.weight"], out[b + "attention.o_proj.bias"] = sd[a + "self_attention.out_proj.weight"], sd[a + "self_attention.out_proj.bias"]
        for tv, hf in (("ln_1", "layernorm_before"), ("ln_2", "layernorm_after"), ("mlp.0", "mlp.fc1"), ("mlp.3", "mlp.fc2")):
            out[b + hf + ".weight"], out[b + hf + ".bias"] = sd[a + tv + ".weight"], sd[a + tv + ".bias"]
    return out


def resnet_to_hf(sd):
    out = {}

    def bn(dst, src):
        for k in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
            out[dst + "." + k] = sd[src + "." + k]

    out["embedder.embedder.convolution.weight"] = sd["conv1.weight"]
    bn("embedder.embedder.normalization", "bn1")
    for li, blocks in enumerate((3, 4, 6, 3)):
        for bi in range(blocks):
            a, b = f"layer{li + 1}.{bi}.", f"encoder.stages.{li}.layers.{bi}."
            for j in range(3):
                out[b + f"layer.{j}.convolution.weight"] = sd[a + f"conv{j + 1}.weight"]
                bn(b + f"layer.{j}.normalization", a + f"bn{j + 1}")
            if bi == 0:
                out[b + "shortcut.convolution.weight"] = sd[a + "downsample.0.weight"]
                bn(b + "shortcut.normalization", a + "downsample.1")
    return out


def load_exact(model, mapped):
    """Load the mapped weights; returns the model keys that got none (position-id buffers and poolers never do)."""
    missing, unexpected = model.load_state_dict(mapped, strict=False)
    assert not unexpected, unexpected
    return [k for k in missing if "position_ids" not in k and not k.startswith("pooler.")]


def scale16(x):
    """mmgclip/networks/image_features.py:95-99 (the reference feeds 16-bit-scaled pixels)."""
    return (65535.0 * x - 32767.5) / 32767.5


# ---- G5: one tower at a time -------------------------------------------------------------------------------------------
def hf_convnext(depths, dims, in_chans):
    from transformers import ConvNextConfig, ConvNextModel
    return ConvNextModel(ConvNextConfig(num_channels=in_chans, depths=list(depths), hidden_sizes=list(dims), layer_norm_eps=1e-6,
                                        drop_path_rate=0.0)).eval()


def golden_convnext(name, depths, dims, in_chans, n, H, W, seed):
    from mmgclip.networks import convnext as CN
    CN.CONFIGS["_golden"] = dict(depths=tuple(depths), dims=tuple(dims))
    feats = R.fill_(CN.build_features("_golden", in_chans), seed, "features.")
    sd = {"features." + k: v for k, v in feats.state_dict().items()}
    hf = hf_convnext(depths, dims, in_chans)
    assert not [k for k in load_exact(hf, convnext_to_hf(sd, depths)) if not k.startswith("layernorm.")]
    img = R.structured_images(n, max(H, W), seed, in_chans)[:, :, :H, :W].contiguous()
    fmap = hf(pixel_values=scale16(img)).last_hidden_state                 # = torchvision `features`
    pooled = fmap.mean((2, 3))                                                # = torchvision `avgpool` (flattened)
    gy = R.seeded_tensor("gy", pooled.shape, seed + 1)
    (pooled * gy).sum().backward()
    grads = {k: p.grad for k, p in hf.named_parameters() if p.grad is not None}
    inv = {hfk: tvk for tvk, hfk in _convnext_key_pairs(depths)}
    out = dict(seed=seed, depths=np.array(depths), dims=np.array(dims), in_chans=in_chans, shape=np.array([n, in_chans, H, W]),
               image_sum=float(img.double().sum()), pooled=t2n(pooled), fmap_shape=np.array(fmap.shape),
               fmap_first=t2n(fmap[:, :, 0, 0]), fmap_last=t2n(fmap[:, :, -1, -1]), gy=t2n(gy))
    for hfk, g in grads.items():                                              # gradient subset: every tensor of <= 1024 numbers
        if hfk in inv and g.numel() <= 1024:
            out["grad." + inv[hfk]] = t2n(g.reshape(sd[inv[hfk]].shape))
    np.savez_compressed(os.path.join(HERE, name), **out)


def _convnext_key_pairs(depths):
    """(torchvision key, transformers key) for every parameter."""
    pairs = [("features.0.0.weight", "embeddings.patch_embeddings.weight"), ("features.0.0.bias", "embeddings.patch_embeddings.bias"),
             ("features.0.1.weight", "embeddings.layernorm.weight"), ("features.0.1.bias", "embeddings.layernorm.bias")]
    for si in range(4):
        st = 1 + 2 * si
        for bi in range(depths[si]):
            a, b = f"features.{st}.{bi}.", f"encoder.stages.{si}.layers.{bi}."
            pairs.append((a + "layer_scale", b + "layer_scale_parameter"))
            for tv, hf in (("block.0", "dwconv"), ("block.2", "layernorm"), ("block.3", "pwconv1"), ("block.5", "pwconv2")):
                pairs += [(a + tv + ".weight", b + hf + ".weight"), (a + tv + ".bias", b + hf + ".bias")]
        if si > 0:
            for j in (0, 1):
                pairs += [(f"features.{st - 1}.{j}.weight", f"encoder.stages.{si}.downsampling_layer.{j}.weight"),
                          (f"features.{st - 1}.{j}.bias", f"encoder.stages.{si}.downsampling_layer.{j}.bias")]
    return pairs


def golden_vit(seed=31):
    from transformers import ViTConfig, ViTModel
    from mmgclip.networks.vit import _tv_layout
    size, layers, hidden = 64, 12, 768
    tv = R.fill_(_tv_layout(size, 1, hidden, layers, 3072, 16), seed)
    sd = tv.state_dict()
    hf = ViTModel(ViTConfig(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=12, intermediate_size=3072,
                            image_size=size, patch_size=16, num_channels=1, layer_norm_eps=1e-6, hidden_dropout_prob=0.0,
                            attention_probs_dropout_prob=0.0, attn_implementation="eager"), add_pooling_layer=False).eval()
    assert not load_exact(hf, vit_to_hf(sd, layers, hidden))
    img = R.structured_images(3, size, seed)
    cls = hf(pixel_values=scale16(img)).last_hidden_state[:, 0]
    gy = R.seeded_tensor("gy", cls.shape, seed + 1)
    (cls * gy).sum().backward()
    g = dict(hf.named_parameters())
    np.savez_compressed(os.path.join(HERE, "g5_vit_b16.npz"), seed=seed, image_size=size, image_sum=float(img.double().sum()),
                        cls=t2n(cls), gy=t2n(gy),
                        **{"grad.class_token": t2n(g["embeddings.cls_token"].grad), "grad.encoder.ln.weight": t2n(g["layernorm.weight"].grad),
                           "grad.encoder.layers.encoder_layer_0.ln_1.weight": t2n(g["layers.0.layernorm_before.weight"].grad),
                           "grad.encoder.layers.encoder_layer_11.mlp.3.bias": t2n(g["layers.11.mlp.fc2.bias"].grad),
                           "grad.encoder.layers.encoder_layer_5.self_attention.in_proj_bias":
                               t2n(torch.cat([g[f"layers.5.attention.{n}.bias"].grad for n in ("q_proj", "k_proj", "v_proj")])),
                           "grad.conv_proj.bias": t2n(g["embeddings.patch_embeddings.projection.bias"].grad)})


def golden_resnet(seed=41):
    from transformers import ResNetConfig, ResNetModel
    from mmgclip.networks.resnet import _TorchvisionResNet
    tv = R.fill_(_TorchvisionResNet(), seed)
    sd = {k: v.clone() for k, v in tv.state_dict().items()}
    img = R.structured_images(4, 64, seed, in_chans=3) * 2.0 - 1.0
    out = dict(seed=seed, image_sum=float(img.double().sum()))
    for mode in ("eval", "train"):
        hf = ResNetModel(ResNetConfig())
        assert not load_exact(hf, resnet_to_hf(sd))
        hf.train(mode == "train")
        with torch.no_grad():
            out["pooled_" + mode] = t2n(hf(pixel_values=img).pooler_output.flatten(1))
    np.savez_compressed(os.path.join(HERE, "g5_resnet50.npz"), **out)


def hf_bert(sd, cfg):
    import transformers
    from mmgclip.networks.bert import hf_config_dict
    hf = transformers.BertModel(transformers.BertConfig(**hf_config_dict(cfg), attn_implementation="eager"), add_pooling_layer=False).eval()
    missing, unexpected = hf.load_state_dict({k: v for k, v in sd.items() if not k.startswith("pooler.")}, strict=False)
    assert not unexpected and not [k for k in missing if "position_ids" not in k], (missing, unexpected)
    return hf


def golden_bert(seed=51):
    from mmgclip.networks.bert import BertConfigLite, _hf_layout
    cfg = BertConfigLite()
    sd = R.fill_(_hf_layout(cfg), seed).state_dict()
    hf = hf_bert(sd, cfg)
    out = dict(seed=seed)
    for n, S in ((4, 77), (2, 256)):
        ids, mask, tt = R.ragged_tokens(n, S, seed)
        with torch.no_grad():
            h = hf(input_ids=ids, attention_mask=mask, token_type_ids=tt).last_hidden_state
        eos = mask.sum(-1) - 1
        out[f"ids_{S}"], out[f"mask_{S}"] = t2n(ids), t2n(mask)
        out[f"eos_hidden_{S}"] = t2n(h[torch.arange(n), eos])
        out[f"cls_hidden_{S}"] = t2n(h[:, 0])
    np.savez_compressed(os.path.join(HERE, "g5_bert_base.npz"), **out)


# ---- G9: BASELINE config C1, end to end ---------------------------------------------------------------------------------
def golden_c1(proj_mod, losses_mod, S, seed=61, min_len=None, tag=""):
    from mmgclip.networks import convnext as CN
    from mmgclip.networks.bert import BertConfigLite, _hf_layout
    n, size = 8, 224
    depths, dims = CN.CONFIGS["tiny"]["depths"], CN.CONFIGS["tiny"]["dims"]
    feats = R.fill_(CN.build_features("tiny", 1), seed, "features.")
    csd = {"features." + k: v for k, v in feats.state_dict().items()}
    cnx = hf_convnext(depths, dims, 1)
    assert not [k for k in load_exact(cnx, convnext_to_hf(csd, depths)) if not k.startswith("layernorm.")]   # final LN: unused
    cfg = BertConfigLite()
    bsd = R.fill_(_hf_layout(cfg), seed + 1).state_dict()
    bert = hf_bert(bsd, cfg)
    torch.manual_seed(0)
    pi = proj_mod.LinearProjectionLayer(embedding_dim=768, projection_dim=512, dropout=0.5)
    pt = proj_mod.LinearProjectionLayer(embedding_dim=768, projection_dim=512, dropout=0.5)
    with torch.no_grad():
        pi.layer.weight.copy_(R.seeded_tensor("image_projection_layer.layer.weight", (512, 768), seed + 2))
        pt.layer.weight.copy_(R.seeded_tensor("text_projection_layer.layer.weight", (512, 768), seed + 2))
    img = R.structured_images(n, size, seed)
    ids, mask, tt = R.ragged_tokens(n, S, seed, min_len=min_len)
    ls = torch.tensor(float(np.log(1 / 0.07)), requires_grad=True)

    pooled = cnx(pixel_values=scale16(img)).last_hidden_state.mean((2, 3))                 # encoder.py:53-54
    pooled.retain_grad()
    hidden = bert(input_ids=ids, attention_mask=mask, token_type_ids=tt).last_hidden_state   # encoder.py:156
    tf = hidden[torch.arange(n), mask.sum(-1) - 1]                                          # mmgclip_model.py:110-111
    tf.retain_grad()
    ip, tp = pi(pooled.flatten(1)), pt(tf)                                                  # mmgclip_model.py:124-125
    ie = ip / ip.norm(dim=1, keepdim=True)                                                  # :128-129
    te = tp / tp.norm(dim=1, keepdim=True)
    s = ls.exp()                                                                            # :132
    li, lt = s * ie @ te.t(), s * te @ ie.t()                                               # :135-136
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        loss, labels = losses_mod.CLIPLoss()(logits_per_image=li, logits_per_text=lt)
    finally:
        torch.Tensor.cuda = orig_cuda
    loss.backward()

    inv = {hfk: tvk for tvk, hfk in _convnext_key_pairs(depths)}
    out = dict(seed=seed, S=S, min_len=-1 if min_len is None else int(min_len), image_sum=float(img.double().sum()), ids=t2n(ids), mask=t2n(mask),
               pooled=t2n(pooled), text_features=t2n(tf), image_embeddings=t2n(ie), text_embeddings=t2n(te),
               logits_per_image=t2n(li), logits_per_text=t2n(lt), loss=t2n(loss), labels=t2n(labels),
               d_pooled=t2n(pooled.grad), d_text_features=t2n(tf.grad), d_logit_scale=t2n(ls.grad),
               d_image_projection_rows=t2n(pi.layer.weight.grad[:16]), d_text_projection_rows=t2n(pt.layer.weight.grad[:16]))
    for hfk, p in cnx.named_parameters():                                   # small ConvNeXt tensors (<= 1536 numbers)
        if p.grad is not None and hfk in inv and p.numel() <= 1536:
            out["grad.image." + inv[hfk]] = t2n(p.grad.reshape(csd[inv[hfk]].shape))
    for k, p in bert.named_parameters():                                    # every 768-wide BERT vector (LayerNorms, biases)
        if p.grad is not None and p.dim() == 1 and p.numel() == 768:
            out["grad.text." + k] = t2n(p.grad)
    used = torch.unique(ids[:, :8])[:24]                                    # some word-embedding rows (also [CLS]/[SEP]/pad)
    out["word_rows"] = t2n(used)
    out["grad.text.embeddings.word_embeddings.weight.rows"] = t2n(bert.embeddings.word_embeddings.weight.grad[used])
    np.savez_compressed(os.path.join(HERE, f"g9_c1_step_s{S}{tag}.npz"), **out)


def main(proj_mod=None, losses_mod=None):
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    if proj_mod is not None and "--only-c1" in sys.argv:      # round 4: the two first C1 fixtures alone (they predated the `min_len` key)
        golden_c1(proj_mod, losses_mod, 77)
        golden_c1(proj_mod, losses_mod, 256)
        return
    golden_convnext("g5_convnext_tiny.npz", (3, 3, 9, 3), (96, 192, 384, 768), 1, 2, 96, 64, seed=11)
    golden_convnext("g5_convnext_small.npz", (2, 2, 2, 2), (32, 64, 128, 256), 3, 2, 77, 50, seed=21)
    golden_vit()
    golden_resnet()
    golden_bert()
    if proj_mod is not None and "--only-c1b" in sys.argv:     # round 3: the second C1 batch alone, the other fixtures untouched
        golden_c1(proj_mod, losses_mod, 256, seed=71, min_len=128, tag="_seed71_long")
        return
    if proj_mod is not None:
        golden_c1(proj_mod, losses_mod, 77)
        golden_c1(proj_mod, losses_mod, 256)
        # a second batch (other weights, other images, long prompts only: 128..256 tokens): is the 1e-3 loss bar typical or lucky?
        golden_c1(proj_mod, losses_mod, 256, seed=71, min_len=128, tag="_seed71_long")
    print("encoder golden vectors written to", HERE)


if __name__ == "__main__":
    import make_golden as MG
    MG.install_cos_sim_stub()
    main(MG.load_by_path("ref_projection", "mmgclip/networks/projection.py"), MG.load_by_path("ref_losses", "mmgclip/loss/losses.py"))
