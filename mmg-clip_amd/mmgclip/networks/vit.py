"""ViT-B/16 image tower on the HIP kernels (forward + backward), state-dict compatible with torchvision `vit_b_16`.

Not in the reference (its image encoders are ConvNeXt-T / ResNet-50, mmgclip/networks/encoder.py:15,57): BASELINE config C4
asks for it (SURVEY.md §7 decision 2: CLS token + learned absolute positions sized to the configured image, pre-LN
ViT-B/16, 768 features).  Layout follows torchvision.models.VisionTransformer: `conv_proj`, `class_token`,
`encoder.pos_embedding`, `encoder.layers.encoder_layer_{i}.{ln_1, self_attention.{in_proj_weight,in_proj_bias,out_proj},
ln_2, mlp.{0,3}}`, `encoder.ln`; LayerNorm eps 1e-6, erf-GELU, no dropout.  The oracle is oracle/encoders_oracle.vit_forward.

All layers reuse the BERT/ConvNeXt kernels: the 16x16/16 patch convolution is `mmg_patchify` + GEMM, attention is
`mmg_attention_fwd/bwd` without a key mask (whole sequence in LDS) up to S = 256 tokens (224x224 -> 197) and the
flash-style tiled `mmg_attention_long_fwd/bwd` beyond (1024x1024 -> S = 4097, BASELINE config C4).
"""
import os

import torch
import torch.nn as nn

from .. import _hip
from .. import kernels as K
from .. import linalg as L
from .._hip import call, ptr, stream
from ..params import ParamArena, backward_finished, note_forward, stream_anchor

LN_EPS = 1e-6


def _tv_layout(image_size, in_chans, hidden, layers, mlp_dim, patch):
    m = nn.Module()
    m.conv_proj = nn.Conv2d(in_chans, hidden, kernel_size=patch, stride=patch)
    m.class_token = nn.Parameter(torch.zeros(1, 1, hidden))
    m.encoder = nn.Module()
    seq = (image_size // patch) ** 2 + 1
    m.encoder.pos_embedding = nn.Parameter(torch.empty(1, seq, hidden).normal_(std=0.02))
    m.encoder.layers = nn.Module()
    for i in range(layers):
        blk = nn.Module()
        blk.ln_1 = nn.LayerNorm(hidden, eps=LN_EPS)
        blk.self_attention = nn.Module()
        blk.self_attention.in_proj_weight = nn.Parameter(torch.empty(3 * hidden, hidden))
        blk.self_attention.in_proj_bias = nn.Parameter(torch.zeros(3 * hidden))
        blk.self_attention.out_proj = nn.Linear(hidden, hidden)
        blk.ln_2 = nn.LayerNorm(hidden, eps=LN_EPS)
        blk.mlp = nn.Sequential(nn.Linear(hidden, mlp_dim), nn.GELU(), nn.Identity(), nn.Linear(mlp_dim, hidden), nn.Identity())
        nn.init.xavier_uniform_(blk.self_attention.in_proj_weight)
        nn.init.xavier_uniform_(blk.mlp[0].weight)
        nn.init.xavier_uniform_(blk.mlp[3].weight)
        nn.init.normal_(blk.mlp[0].bias, std=1e-6)
        nn.init.normal_(blk.mlp[3].bias, std=1e-6)
        setattr(m.encoder.layers, f"encoder_layer_{i}", blk)
    m.encoder.ln = nn.LayerNorm(hidden, eps=LN_EPS)
    fan_in = in_chans * patch * patch
    nn.init.trunc_normal_(m.conv_proj.weight, std=(1.0 / fan_in) ** 0.5)
    nn.init.zeros_(m.conv_proj.bias)
    return m


class ViTTower(nn.Module):
    """pixels fp32 [n, Cin, H, W] -> class-token features [n, hidden]."""

    def __init__(self, image_size=224, in_chans=1, hidden=768, layers=12, heads=12, mlp_dim=3072, patch=16, scale16=True,
                 micro_batch=256, checkpoint=False):
        super().__init__()
        self.checkpoint = checkpoint        # keep only each micro-batch's pixels, re-run its forward inside its backward
        assert hidden == 64 * heads, "the attention kernel is specialised for head_dim 64"
        self.image_size, self.in_chans, self.hidden, self.layers, self.heads = image_size, in_chans, hidden, layers, heads
        self.mlp_dim, self.patch, self.scale16, self.micro_batch = mlp_dim, patch, scale16, micro_batch
        self.seq = (image_size // patch) ** 2 + 1
        self.model = _tv_layout(image_size, in_chans, hidden, layers, mlp_dim, patch)
        self.model_output_dimension = hidden
        self.kp = (patch * patch * in_chans + 31) // 32 * 32
        self._arena = self._wc = self._wc_version = self._anchor = None
        self.post_backward_hook = None

    def _blk(self, i):
        return getattr(self.model.encoder.layers, f"encoder_layer_{i}")

    def _materialize(self, device):
        if self._arena is not None and self._arena.device == device and self._arena.is_bound():
            return
        self._arena = ParamArena(list(self.model.named_parameters()), device)
        self._wc_version = None
        self._anchor = torch.zeros(1, device=device, requires_grad=True)

    @property
    def arena(self):
        return self._arena

    def _refresh_working_copies(self):
        A = self._arena
        v = A.version()
        if self._wc_version == v:
            return
        wc = {}
        cw = self.model.conv_proj.weight.data
        w = torch.zeros(cw.shape[0], self.kp, device=cw.device)
        w[:, :self.patch * self.patch * self.in_chans] = cw.permute(0, 2, 3, 1).reshape(cw.shape[0], -1)
        wc["conv"] = K.cast_bf16(w)
        wc["cls"] = K.cast_bf16(self.model.class_token.data.reshape(-1))
        wc["pos"] = K.cast_bf16(self.model.encoder.pos_embedding.data.reshape(self.seq, self.hidden))
        for i in range(self.layers):
            b = self._blk(i)
            for tag, wt in (("qkv", b.self_attention.in_proj_weight.data), ("o", b.self_attention.out_proj.weight.data),
                            ("f1", b.mlp[0].weight.data), ("f2", b.mlp[3].weight.data)):
                wc[f"{i}.{tag}"] = K.cast_bf16(wt)
                wc[f"{i}.{tag}t"] = K.transpose_cast_bf16(wt)
        self._wc, self._wc_version = wc, v

    def _forward_mb(self, img, save):
        wc, H, S, heads = self._wc, self.hidden, self.seq, self.heads
        B = img.shape[0]
        p0 = K.patchify(img, self.patch, self.kp, self.scale16)
        tok = L.gemm_nt(p0, wc["conv"], bias=self.model.conv_proj.bias.data)
        x = torch.empty(B * S, H, device=img.device, dtype=torch.bfloat16)
        call("mmg_vit_assemble_fwd", ptr(tok), ptr(wc["cls"]), ptr(wc["pos"]), ptr(x), B, S, H, stream())
        saved = {"p0": p0, "layers": [], "B": B} if save else None
        del tok
        for i in range(self.layers):
            b = self._blk(i)
            y, m1, r1 = K.layernorm_fwd(x, b.ln_1.weight.data, b.ln_1.bias.data, LN_EPS, want_stats=save)
            qkv = L.gemm_nt(y, wc[f"{i}.qkv"], bias=b.self_attention.in_proj_bias.data)
            ctx, lse = K.attention_fwd(qkv, None, B, S, heads, want_lse=save)
            x1 = L.gemm_nt(ctx, wc[f"{i}.o"], bias=b.self_attention.out_proj.bias.data, residual=x)
            z, m2, r2 = K.layernorm_fwd(x1, b.ln_2.weight.data, b.ln_2.bias.data, LN_EPS, want_stats=save)
            hpre = torch.empty(B * S, self.mlp_dim, device=img.device, dtype=torch.bfloat16) if save else None
            g = L.gemm_nt(z, wc[f"{i}.f1"], bias=b.mlp[0].bias.data, epi=L.EPI_GELU, aux_out=hpre)
            x2 = L.gemm_nt(g, wc[f"{i}.f2"], bias=b.mlp[3].bias.data, residual=x1)
            if save:
                saved["layers"].append((x, m1, r1, qkv, ctx, lse, x1, m2, r2, hpre))
            del y, g, z
            x = x2
        ln = self.model.encoder.ln
        out, mf, rf = K.layernorm_fwd(x, ln.weight.data, ln.bias.data, LN_EPS, want_stats=save)
        idx = torch.zeros(B, device=img.device, dtype=torch.int32)
        feat = torch.empty(B, H, device=img.device, dtype=torch.float32)
        call("mmg_gather_rows_fwd", ptr(out), ptr(idx), ptr(feat), B, S, H, stream())
        if save:
            saved["final"] = (x, mf, rf, idx)
        return feat, saved

    def _backward_mb(self, dfeat, saved):
        wc, A, H, S, heads = self._wc, self._arena, self.hidden, self.seq, self.heads
        B = saved["B"]
        x, mf, rf, idx = saved["final"]
        ln = self.model.encoder.ln
        dout = K.eos_pool_bwd(dfeat, idx, B, S)
        dx = K.layernorm_bwd(dout, x, mf, rf, ln.weight.data, A.g("encoder.ln.weight"), A.g("encoder.ln.bias"))
        del dout
        for i in range(self.layers - 1, -1, -1):
            b = self._blk(i)
            p = f"encoder.layers.encoder_layer_{i}."
            x, m1, r1, qkv, ctx, lse, x1, m2, r2, hpre = saved["layers"][i]
            # MLP branch: x2 = x1 + W2 gelu(W1 LN2(x1) + b1) + b2
            g = torch.empty_like(hpre)
            dh = L.gemm_nt(dx, wc[f"{i}.f2t"], epi=L.EPI_DGELU, aux_in=hpre, aux_out=g)
            L.gemm_tn_acc(dx, g, A.g(p + "mlp.3.weight"), colsum=A.g(p + "mlp.3.bias"))
            del g
            z, _, _ = K.layernorm_fwd(x1, b.ln_2.weight.data, b.ln_2.bias.data, LN_EPS, want_stats=False)
            L.gemm_tn_acc(dh, z, A.g(p + "mlp.0.weight"), colsum=A.g(p + "mlp.0.bias"))
            del z
            dz = L.gemm_nt(dh, wc[f"{i}.f1t"])
            del dh
            dx1 = K.layernorm_bwd(dz, x1, m2, r2, b.ln_2.weight.data, A.g(p + "ln_2.weight"), A.g(p + "ln_2.bias"),
                                  add=dx)                          # + residual path, fused in the LN-backward kernel
            del dz
            # attention branch: x1 = x + Wo attn(Wqkv LN1(x) + b) + bo
            L.gemm_tn_acc(dx1, ctx, A.g(p + "self_attention.out_proj.weight"), colsum=A.g(p + "self_attention.out_proj.bias"))
            dctx = L.gemm_nt(dx1, wc[f"{i}.ot"])
            dqkv = K.attention_bwd(qkv, None, ctx, lse, dctx, B, S, heads)
            del dctx
            y, _, _ = K.layernorm_fwd(x, b.ln_1.weight.data, b.ln_1.bias.data, LN_EPS, want_stats=False)
            L.gemm_tn_acc(dqkv, y, A.g(p + "self_attention.in_proj_weight"), colsum=A.g(p + "self_attention.in_proj_bias"))
            del y
            dy = L.gemm_nt(dqkv, wc[f"{i}.qkvt"])
            del dqkv
            dx = K.layernorm_bwd(dy, x, m1, r1, b.ln_1.weight.data, A.g(p + "ln_1.weight"), A.g(p + "ln_1.bias"), add=dx1)
            del dy, dx1
            saved["layers"][i] = None
        dtok = torch.empty(B * (S - 1), H, device=dx.device, dtype=torch.bfloat16)
        call("mmg_vit_assemble_bwd", ptr(dx), ptr(dtok), ptr(A.g("encoder.pos_embedding")), ptr(A.g("class_token")), B, S, H, stream())
        tmp = torch.zeros(H, self.kp, device=dx.device, dtype=torch.float32)
        L.gemm_tn_acc(dtok, saved["p0"], tmp, colsum=A.g("conv_proj.bias"))
        kk = self.patch * self.patch * self.in_chans
        src = tmp[:, :kk].contiguous()
        call("mmg_grad_relayout", ptr(src), ptr(A.g("conv_proj.weight")), 0, H, self.in_chans, self.patch, self.patch, kk, stream())

    def forward(self, images):
        _hip.require_gpu(images)
        if images.shape[-1] != self.image_size or images.shape[-2] != self.image_size:
            raise ValueError(f"ViT was built for {self.image_size}x{self.image_size} inputs (learned positions), got {tuple(images.shape)}")
        self._materialize(images.device)
        needs_grad = torch.is_grad_enabled() and self._arena.any_trainable()
        note_forward(self, needs_grad)
        return _ViTFn.apply(self, images.float().contiguous(), stream_anchor(self, self._anchor.device) if needs_grad else None)


class _ViTFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tower, images, anchor):
        tower._refresh_working_copies()
        save = anchor is not None
        ckpt = save and tower.checkpoint
        feats, saved = [], []
        for i in range(0, images.shape[0], tower.micro_batch):
            # (the last micro-batch keeps its activations and is the first one the backward takes: one recomputation less, still one
            #  micro-batch of activations alive at a time - MMG_CKPT_KEEP_LAST=0 recomputes all)
            keep = ckpt and i + tower.micro_batch >= images.shape[0] and os.environ.get("MMG_CKPT_KEEP_LAST", "1") != "0"
            ft, sv = tower._forward_mb(images[i:i + tower.micro_batch], save and (not ckpt or keep))
            feats.append(ft)
            saved.append({"recompute": images[i:i + tower.micro_batch], "B": ft.shape[0]} if (ckpt and not keep) else sv)
        ctx.tower, ctx.saved_mb, ctx.reverse = tower, saved if save else None, ckpt
        return torch.cat(feats, 0) if len(feats) > 1 else feats[0]

    @staticmethod
    def backward(ctx, dfeat):
        tower = ctx.tower
        tower._arena.prepare_grads()
        dfeat = dfeat.float().contiguous()
        sizes = [sv["B"] for sv in ctx.saved_mb]
        starts = [sum(sizes[:k]) for k in range(len(sizes))]
        order = list(range(len(sizes)))
        if ctx.reverse:
            order.reverse()
        for k in order:
            sv = ctx.saved_mb[k]
            if "recompute" in sv:
                _, sv = tower._forward_mb(sv["recompute"], True)
            tower._backward_mb(dfeat[starts[k]:starts[k] + sizes[k]].contiguous(), sv)
            ctx.saved_mb[k] = None
        ctx.saved_mb = None
        backward_finished(tower)
        return None, None, None
