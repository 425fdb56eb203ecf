"""CPU oracle for the encoder towers — TEST INFRASTRUCTURE ONLY (see oracle/clip_oracle.py header).

Pinning.  The arithmetic of the towers lives in third-party packages that are not in the reference tree — torchvision
0.14.1 `ConvNeXt` (inside a TorchScript archive the repo does not ship; torchvision is not installed here) and `resnet50`,
transformers 4.41.0 `BertModel` (5.15 installed) — and the reference holds no golden output for any of them,
so the reference ITSELF cannot pin this file.  It is pinned against independent third-party implementations instead:
tests/golden/make_golden_encoders.py runs transformers' `ConvNextModel`, `ViTModel`, `ResNetModel` and `BertModel` on seeded
weights (tests/golden/recipes.py) at the full architectures (ConvNeXt-T 3/3/9/3, ViT-B/16, ResNet-50, BERT-base 12 layers) and
stores their outputs and autograd gradients (tests/golden/g5_*.npz); g9_c1_step_s{77,256}.npz is BASELINE config C1 end to end
(those towers -> the reference's own LinearProjectionLayer and CLIPLoss).  tests/test_oracle_encoders.py checks every function
below against them (forward 2e-5, gradients 2e-4...5e-4 of the largest magnitude).  What stays unpinned: torchvision's own
rounding order (same formulas, different kernels) and the private pretrained weights.

The restatements follow the published module definitions and the call order at the reference's call sites:
    mmgclip/networks/encoder.py:40-55      ConvNextTiny.forward = model.features(x) -> model.avgpool(x)
    mmgclip/networks/image_features.py:95-99   x*65535 ; (x - 32767.5)/32767.5
    mmgclip/networks/encoder.py:101-117    ResNet50Encoder.forward
    mmgclip/networks/encoder.py:146-156    BertEncoder.forward = model(**x)['last_hidden_state']
    notebooks/clf_convnext_tiny_experimental.ipynb cell 3   (module tree)   notebooks/bert_experimental.ipynb:609-624 (config)
"""
import math

import torch
import torch.nn.functional as F


def q_e4m3(t):
    """Round to OCP e4m3 (round-to-nearest-even, saturating at +-448) and back to fp32: the value an e4m3 byte holds."""
    return t.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32)


def q_e4m3_weight(w):
    """Per-tensor power-of-two scaling as csrc/fp8_ops.hip: scale = 2^floor(log2(448 / max|w|)); returns the dequantised weight
    (exact in fp32: the scale is a power of two)."""
    amax = float(w.detach().abs().max())
    scale = 1.0
    if 0.0 < amax < 3.0e38:
        scale = 2.0 ** math.floor(math.log2(448.0 / amax))
        if amax * scale > 448.0:
            scale *= 0.5
    return q_e4m3(w * scale) / scale


def convnext_forward(sd, images, depths=(3, 3, 9, 3), scale16=True, prefix="features.", fp8_min_channels=None, fp8_backward=False):
    """torchvision ConvNeXt `features` + `avgpool` from a state dict (fp32).  images [n,Cin,H,W]; returns [n,C,1,1].
    fp8_min_channels: blocks with C % 128 == 0 and C >= that value run their two Linear layers on e4m3 operands (LayerNorm and
    GELU outputs rounded to e4m3 unscaled, weights per-tensor scaled), fp32 accumulation - BASELINE config C5's forward; the
    rounding is a straight-through identity for autograd, as the bf16 backward of the build treats it; fp8_backward=True: those blocks' backward in
    8 bits as well (Fp8BlockMLP below: e5m2 gradients, the build's round-4 default)."""
    x = images
    if scale16:
        x = 65535.0 * x
        x = (x - 32767.5) / 32767.5
    g = lambda k: sd[prefix + k]                                           # noqa: E731

    def ln2d(x, k):      # LayerNorm2d: permute to NHWC, layer_norm over C, permute back
        x = x.permute(0, 2, 3, 1)
        x = F.layer_norm(x, (x.shape[-1],), g(k + ".weight"), g(k + ".bias"), 1e-6)
        return x.permute(0, 3, 1, 2)

    x = F.conv2d(x, g("0.0.weight"), g("0.0.bias"), stride=4)
    x = ln2d(x, "0.1")
    for si in range(4):
        st = 1 + 2 * si
        for bi in range(depths[si]):
            k = f"{st}.{bi}."
            C = x.shape[1]
            y = F.conv2d(x, g(k + "block.0.weight"), g(k + "block.0.bias"), padding=3, groups=C)
            y = y.permute(0, 2, 3, 1)
            y = F.layer_norm(y, (C,), g(k + "block.2.weight"), g(k + "block.2.bias"), 1e-6)
            if fp8_backward and fp8_min_channels is not None and C % 128 == 0 and C >= fp8_min_channels:
                y = Fp8BlockMLP.apply(y, g(k + "block.3.weight"), g(k + "block.3.bias"), g(k + "block.5.weight"), g(k + "block.5.bias"),
                                      g(k + "layer_scale").reshape(C))
                x = x + y.permute(0, 3, 1, 2)
                continue
            if fp8_min_channels is not None and C % 128 == 0 and C >= fp8_min_channels:
                ste = lambda t, q: t + (q - t).detach()                    # noqa: E731  value of q, gradient of t
                w1, w2 = g(k + "block.3.weight"), g(k + "block.5.weight")
                y = F.linear(ste(y, q_e4m3(y)), ste(w1, q_e4m3_weight(w1)), g(k + "block.3.bias"))
                y = F.gelu(y)
                y = F.linear(ste(y, q_e4m3(y)), ste(w2, q_e4m3_weight(w2)), g(k + "block.5.bias"))
            else:
                y = F.linear(y, g(k + "block.3.weight"), g(k + "block.3.bias"))
                y = F.gelu(y)
                y = F.linear(y, g(k + "block.5.weight"), g(k + "block.5.bias"))
            y = y.permute(0, 3, 1, 2)
            x = x + g(k + "layer_scale") * y             # stochastic depth p = 0
        if si < 3:
            x = ln2d(x, f"{st + 1}.0")
            x = F.conv2d(x, g(f"{st + 1}.1.weight"), g(f"{st + 1}.1.bias"), stride=2)
    fmap = x
    return F.adaptive_avg_pool2d(x, 1), fmap


def bert_forward(sd, ids, attention_mask=None, token_type_ids=None, heads=12, eps=1e-12, prefix="", dropout=None):
    """HF BertModel.last_hidden_state from a state dict, fp32.  dropout=None: eval mode.  dropout=(p_hidden, p_attention, seed):
    training mode with the counter-based masks of oracle/dropout_oracle.py at HF's four dropout positions (BertEmbeddings after
    its LayerNorm; BertSelfAttention on the probabilities; BertSelfOutput and BertOutput on the dense output before the residual
    add) - the reference reaches them through model.train() (mmgclip/experiments/ClassifierExperiment.py:97)."""
    g = lambda k: sd[prefix + k]                                           # noqa: E731
    B, S = ids.shape
    if token_type_ids is None:
        token_type_ids = torch.zeros_like(ids)
    x = g("embeddings.word_embeddings.weight")[ids] + g("embeddings.token_type_embeddings.weight")[token_type_ids] \
        + g("embeddings.position_embeddings.weight")[:S][None]
    H = x.shape[-1]
    x = F.layer_norm(x, (H,), g("embeddings.LayerNorm.weight"), g("embeddings.LayerNorm.bias"), eps)
    if dropout is not None:
        from . import dropout_oracle as D
        p_h, p_a, seed = dropout

        def drop_hidden(t, site):
            if p_h <= 0:
                return t
            keep = torch.from_numpy(D.hidden_mask(B * S, H, p_h, seed, site)).view(B, S, H)
            return t * keep.to(t.dtype) * (1.0 / (1.0 - float(torch.tensor(p_h, dtype=torch.float32))))

        x = drop_hidden(x, D.SITE_EMBEDDINGS)
    add = None
    if attention_mask is not None:
        add = (1.0 - attention_mask.to(x.dtype))[:, None, None, :] * torch.finfo(x.dtype).min
    i = 0
    while prefix + f"encoder.layer.{i}.attention.self.query.weight" in sd:
        p = f"encoder.layer.{i}."
        lin = lambda t, k: F.linear(t, g(p + k + ".weight"), g(p + k + ".bias"))   # noqa: E731
        split = lambda t: t.view(B, S, heads, H // heads).permute(0, 2, 1, 3)        # noqa: E731
        q, k, v = split(lin(x, "attention.self.query")), split(lin(x, "attention.self.key")), split(lin(x, "attention.self.value"))
        s = q @ k.transpose(-1, -2) / (H // heads) ** 0.5
        if add is not None:
            s = s + add
        prob = s.softmax(-1)
        if dropout is not None and p_a > 0:
            keep = torch.from_numpy(D.attention_mask(B, heads, S, p_a, seed, D.site_attention_probs(i)))
            prob = prob * keep.to(prob.dtype) * (1.0 / (1.0 - float(torch.tensor(p_a, dtype=torch.float32))))
        ctx = (prob @ v).permute(0, 2, 1, 3).reshape(B, S, H)
        ao = lin(ctx, "attention.output.dense")
        if dropout is not None:
            ao = drop_hidden(ao, D.site_attention_output(i))
        a = F.layer_norm(ao + x, (H,), g(p + "attention.output.LayerNorm.weight"), g(p + "attention.output.LayerNorm.bias"), eps)
        f = lin(F.gelu(lin(a, "intermediate.dense")), "output.dense")
        if dropout is not None:
            f = drop_hidden(f, D.site_ffn_output(i))
        x = F.layer_norm(f + a, (H,), g(p + "output.LayerNorm.weight"), g(p + "output.LayerNorm.bias"), eps)
        i += 1
    return x


def vit_forward(sd, images, heads=12, patch=16, scale16=True, prefix=""):
    """torchvision VisionTransformer (vit_b_16 layout) up to the class token after encoder.ln, fp32.  Not in the reference
    (BASELINE config C4); pinned only by the published torchvision definition: conv_proj patches -> [cls; tokens] +
    pos_embedding -> N x {x + MHA(ln_1(x)); x + MLP(ln_2(x))} -> encoder.ln -> x[:, 0]."""
    g = lambda k: sd[prefix + k]                                           # noqa: E731
    x = images
    if scale16:
        x = (65535.0 * x - 32767.5) / 32767.5
    x = F.conv2d(x, g("conv_proj.weight"), g("conv_proj.bias"), stride=patch)
    B, H = x.shape[0], x.shape[1]
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([g("class_token").expand(B, -1, -1), x], dim=1) + g("encoder.pos_embedding")
    S = x.shape[1]
    i = 0
    while prefix + f"encoder.layers.encoder_layer_{i}.ln_1.weight" in sd:
        p = f"encoder.layers.encoder_layer_{i}."
        y = F.layer_norm(x, (H,), g(p + "ln_1.weight"), g(p + "ln_1.bias"), 1e-6)
        qkv = F.linear(y, g(p + "self_attention.in_proj_weight"), g(p + "self_attention.in_proj_bias"))
        q, k, v = (t.view(B, S, heads, H // heads).permute(0, 2, 1, 3) for t in qkv.chunk(3, dim=-1))
        a = ((q @ k.transpose(-1, -2)) / (H // heads) ** 0.5).softmax(-1) @ v
        a = a.permute(0, 2, 1, 3).reshape(B, S, H)
        x = x + F.linear(a, g(p + "self_attention.out_proj.weight"), g(p + "self_attention.out_proj.bias"))
        z = F.layer_norm(x, (H,), g(p + "ln_2.weight"), g(p + "ln_2.bias"), 1e-6)
        x = x + F.linear(F.gelu(F.linear(z, g(p + "mlp.0.weight"), g(p + "mlp.0.bias"))), g(p + "mlp.3.weight"), g(p + "mlp.3.bias"))
        i += 1
    x = F.layer_norm(x, (H,), g("encoder.ln.weight"), g("encoder.ln.bias"), 1e-6)
    return x[:, 0]


def resnet50_forward(sd, images, train_bn=True, prefix="", storage_bf16=False):
    """torchvision resnet50 without `fc`, as the reference's ResNet50Encoder.forward runs it (mmgclip/networks/encoder.py:
    101-117), from a state dict (fp32).  train_bn=True: batch statistics in every BatchNorm (the reference calls model.train()
    on the whole model; frozen layers included).  storage_bf16=True rounds every stored activation and every weight to bf16 (fp32
    arithmetic in between), i.e. the storage format of the device path: 53 batch-statistics normalisations in a row amplify
    that rounding, so the tight comparison of the device tower is against this variant.  Returns [n, 2048]."""
    q = (lambda t: t.to(torch.bfloat16).float()) if storage_bf16 else (lambda t: t)          # noqa: E731
    g = lambda k: sd[prefix + k]                                           # noqa: E731
    conv = lambda x, k, **kw: q(F.conv2d(x, q(g(k)), **kw))                # noqa: E731

    def bn(x, k, res=None, relu=True):
        if train_bn:
            y = F.batch_norm(x, None, None, g(k + ".weight"), g(k + ".bias"), True, 0.1, 1e-5)
        else:
            y = F.batch_norm(x, g(k + ".running_mean"), g(k + ".running_var"), g(k + ".weight"), g(k + ".bias"), False, 0.1, 1e-5)
        if res is not None:
            y = y + res
        return q(F.relu(y) if relu else y)

    x = images
    if x.dim() == 2:
        x = x.view(x.shape[0], 1, 1, x.shape[1]).repeat(1, 3, 1, 1)
    x = bn(conv(q(x), "conv1.weight", stride=2, padding=3), "bn1")
    x = F.max_pool2d(x, 3, 2, 1)
    for li, (blocks, stride) in enumerate(((3, 1), (4, 2), (6, 2), (3, 2))):
        for bi in range(blocks):
            k = f"layer{li + 1}.{bi}."
            s = stride if bi == 0 else 1
            y = bn(conv(x, k + "conv1.weight"), k + "bn1")
            y = bn(conv(y, k + "conv2.weight", stride=s, padding=1), k + "bn2")
            y = conv(y, k + "conv3.weight")
            idn = bn(conv(x, k + "downsample.0.weight", stride=s), k + "downsample.1", relu=False) if bi == 0 else x
            x = bn(y, k + "bn3", res=idn)
    return F.adaptive_avg_pool2d(x, 1).flatten(1)


# ---- the 8-bit backward of config C5's blocks (round 4) -------------------------------------------------------------------------------------------
def q_e5m2(t):
    """Round to OCP e5m2 (round-to-nearest-even, saturating at +-57344) and back to fp32."""
    return t.clamp(-57344.0, 57344.0).to(torch.float8_e5m2).to(torch.float32)


def e5m2_scale(t):
    """Per-tensor power-of-two gradient scale as csrc/fp8_ops.hip (quantize_e5m2_kernel): 2^floor(log2(16384 / max|t|)), 1 for an all-zero tensor."""
    amax = float(t.detach().abs().max())
    scale = 1.0
    if 0.0 < amax < 3.0e38:
        scale = 2.0 ** math.floor(math.log2(16384.0 / amax))
        if amax * scale > 16384.0:
            scale *= 0.5
    return scale


class Fp8BlockMLP(torch.autograd.Function):
    """gamma * ( GELU( q(ln) q(W1)^T + b1 ) -> q -> q(W2)^T + b2 ) of a CNBlock (LayerNorm output `ln` [.., C]) with the arithmetic of the
    build's fp8 path in BOTH directions (mmgclip/networks/convnext.py, the `.w2gt8` branch):
      forward   e4m3 operands, fp32 accumulation (as convnext_forward's straight-through form);
      backward  dy8 = e5m2(dy s) with s = e5m2_scale(dy);  dh8 = e5m2((dy8 q(gamma W2)) GELU'(h)) at the SAME scale;
                d ln = (dh8 / s) q(W1);  dW2 = gamma (dy8^T g8) / s,  db2 = gamma colsum(dy) (from dy itself),  dW1 = (dh8^T ln8) / s,
                db1 = colsum(dh8) / s,  d gamma = sum over rows of dy * (the forward's MLP output)."""

    @staticmethod
    def forward(ctx, ln, w1, b1, w2, b2, gamma):
        ln8 = q_e4m3(ln)
        h = F.linear(ln8, q_e4m3_weight(w1), b1)
        g8 = q_e4m3(F.gelu(h))
        y = F.linear(g8, q_e4m3_weight(w2), b2)
        ctx.save_for_backward(ln8, h, g8, y, w1, w2, gamma)
        return gamma * y

    @staticmethod
    def backward(ctx, dy):
        ln8, h, g8, y, w1, w2, gamma = ctx.saved_tensors
        C = dy.shape[-1]
        dyf, lnf, hf, gf = dy.reshape(-1, C), ln8.reshape(-1, C), h.reshape(-1, 4 * C), g8.reshape(-1, 4 * C)
        s = e5m2_scale(dyf)
        dy8 = q_e5m2(dyf * s)                                            # (values at scale s)
        w2gt = q_e4m3_weight((w2 * gamma.reshape(C, 1)).t().contiguous())  # [4C, C]
        dgelu = 0.5 * (1 + torch.erf(hf / math.sqrt(2))) + hf * torch.exp(-0.5 * hf * hf) / math.sqrt(2 * math.pi)
        dh8 = q_e5m2((dy8 @ w2gt.t()) * dgelu)                           # still at scale s
        w1t = q_e4m3_weight(w1.t().contiguous())                         # [C, 4C]
        dln = (dh8 @ w1t.t()) / s
        dw2 = gamma.reshape(C, 1) * (dy8.t() @ gf) / s
        db2 = gamma * dyf.sum(0)
        dw1 = (dh8.t() @ lnf) / s
        db1 = dh8.sum(0) / s
        dgamma = (dyf * y.reshape(-1, C)).sum(0)
        return dln.reshape(ln8.shape), dw1, db1, dw2, db2, dgamma
