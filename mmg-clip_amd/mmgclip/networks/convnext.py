"""ConvNeXt image tower on the HIP kernels: forward + backward over NHWC bf16 activations.

What the reference has (path:line in the reference tree): `ConvNextTiny.forward` = `model.features(x)` then
`model.avgpool(x)` on a torchvision-layout ConvNeXt-T TorchScript archive (mmgclip/networks/encoder.py:40-55), fed by
`(x*65535 - 32767.5)/32767.5` (mmgclip/networks/image_features.py:95-99); module tree printed in
notebooks/clf_convnext_tiny_experimental.ipynb cell 3 (Conv2dNormActivation(Conv2d 4x4/4, LayerNorm2d), CNBlock(dwconv7,
Permute, LayerNorm, Linear C->4C, GELU, Linear 4C->C, Permute) * layer_scale + residual, depths 3/3/9/3, dims
96/192/384/768; downsample = LayerNorm2d + Conv2d 2x2/2).  The reference never trains it; the backward here is new
capability (north star) and is checked against oracle/encoders_oracle.py.

Parameter names/shapes follow torchvision (`features.{i}.{j}.block.{k}.weight`, `layer_scale` ...) so a torchvision
state dict loads unchanged.  Stochastic depth is not applied (p = 0).

Data layout on the MI355X: activations are bf16 [n*H*W, C] (NHWC flattened), so every pointwise Linear is a plain
row-major GEMM and LayerNorm reads contiguous rows; the 2x2/4x4 stride=kernel convolutions become GEMMs on patchified
rows (the LayerNorm kernel writes the patchified layout directly).  Blocks with C <= 384 run LayerNorm + Linear + GELU +
Linear + layer scale + residual as one fused launch (csrc/cnblock_mlp.hip); for C <= 192 the backward recomputes the hidden
row on chip.  Saved for backward per block and pixel: block input (C), depthwise output (C) and - only where the backward
does not recompute it - the pre-GELU hidden (4C) in bf16 + LN statistics; LN output and GELU output are rebuilt.
"""
import os

import torch
import torch.nn as nn

from .. import _hip
from .. import kernels as K
from .. import linalg as L
from ..params import ParamArena, backward_finished, last_backward, note_forward, stream_anchor

CONFIGS = {
    "tiny": dict(depths=(3, 3, 9, 3), dims=(96, 192, 384, 768)),
    "small": dict(depths=(3, 3, 27, 3), dims=(96, 192, 384, 768)),
    "base": dict(depths=(3, 3, 27, 3), dims=(128, 256, 512, 1024)),
}
LN_EPS = 1e-6


class LayerNorm2d(nn.LayerNorm):
    """Parameter container with torchvision's name; the arithmetic runs in mmg_layernorm_fwd."""


class CNBlock(nn.Module):
    def __init__(self, dim, layer_scale=1e-6):
        super().__init__()
        self.block = nn.Sequential(
            nn.Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim, bias=True),
            nn.Identity(),                      # Permute([0, 2, 3, 1])
            nn.LayerNorm(dim, eps=LN_EPS),
            nn.Linear(dim, 4 * dim, bias=True),
            nn.GELU(),
            nn.Linear(4 * dim, dim, bias=True),
            nn.Identity(),                      # Permute([0, 3, 1, 2])
        )
        self.layer_scale = nn.Parameter(torch.ones(dim, 1, 1) * layer_scale)


def build_features(variant="tiny", in_chans=1):
    cfg = CONFIGS[variant]
    dims, depths = cfg["dims"], cfg["depths"]
    layers = [nn.Sequential(nn.Conv2d(in_chans, dims[0], kernel_size=4, stride=4, bias=True), LayerNorm2d(dims[0], eps=LN_EPS))]
    for i, (d, n) in enumerate(zip(dims, depths)):
        layers.append(nn.Sequential(*[CNBlock(d) for _ in range(n)]))
        if i < 3:
            layers.append(nn.Sequential(LayerNorm2d(d, eps=LN_EPS), nn.Conv2d(d, dims[i + 1], kernel_size=2, stride=2)))
    feats = nn.Sequential(*layers)
    for m in feats.modules():                       # torchvision ConvNeXt init
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
    return feats


class _TorchvisionLayout(nn.Module):
    """`.features` / `.avgpool` holder so state-dict keys read `model.features...` like the reference archive."""

    def __init__(self, variant, in_chans):
        super().__init__()
        self.features = build_features(variant, in_chans)
        self.avgpool = nn.AdaptiveAvgPool2d(1)


class ConvNextTower(nn.Module):
    """pixels fp32 [n, Cin, H, W] in [0,1] (scale16=True applies the reference's 16-bit scaling) -> features [n, dims[-1]]."""

    def __init__(self, variant="tiny", in_chans=1, scale16=True, micro_batch=64, fused_mlp=None, checkpoint=False, fp8=False,
                 fp8_min_channels=None):
        super().__init__()
        self.variant, self.in_chans, self.scale16, self.micro_batch = variant, in_chans, scale16, micro_batch
        # narrow stages (C <= 256) run the CNBlock MLP as one fused launch; MMG_FUSED_MLP=0 keeps the GEMM pair
        self.fused_mlp = (os.environ.get("MMG_FUSED_MLP", "1") != "0") if fused_mlp is None else bool(fused_mlp)
        self.fused_bwd_saved_h = os.environ.get("MMG_FUSED_MLP_BWD_SAVED_H", "0") == "1"
        self.bwdw = os.environ.get("MMG_BWDW", "1") != "0"        # on-chip weight-gradient backward where supported (csrc/cnblock_bwdw.hip)
        # blocks whose backward is the GEMM pair keep their LayerNorm output ([M,C] bf16) instead of recomputing it - decided per forward:
        # only while those copies stay below 4 % of the device memory (C2: 8.4 GB on; ConvNeXt-B at 256 images without checkpointing:
        # 31 GB on top of 267 GiB of activations, off).  MMG_SAVE_LN=0 / 1 force it.
        self.save_ln_mode = os.environ.get("MMG_SAVE_LN", "auto")
        self.save_ln = self.save_ln_mode == "1"
        # the same blocks also keep GELU(hidden) ([M,4C] bf16) from the forward, so that the backward's data-gradient GEMM applies GELU' only
        # (its epilogue is VALU-bound and half of it rebuilt that activation) - decided per forward like save_ln, while those copies stay
        # below 15 % of the device memory (C2: 33.8 GB on; ConvNeXt-B without checkpointing: 131 GB, off).  MMG_SAVE_GELU=0 / 1 force it.
        self.save_gelu_mode = os.environ.get("MMG_SAVE_GELU", "auto")
        self.save_gelu = self.save_gelu_mode == "1"
        # with GELU(h) kept, keep GELU'(h) instead of h as the second 4C-wide tensor (round 4; MMG_SAVE_DGELU=0: h, and the polynomial in the backward)
        self.save_dgelu = os.environ.get("MMG_SAVE_DGELU", "1") != "0"
        self.checkpoint = checkpoint        # recompute each micro-batch's forward in the backward (north-star config C5)
        # fp8 (config C5): the two pointwise GEMMs of every block with C % 128 == 0 and C >= fp8_min_channels run their FORWARD
        # on e4m3 operands (LayerNorm / GELU outputs cast unscaled, weights with a per-tensor power-of-two scale); the backward
        # stays bf16 on the saved pre-activation.  Default 512: the stages whose blocks are GEMM pairs anyway.
        self.fp8 = bool(fp8)
        # round 4: the same blocks' BACKWARD in 8 bits too (MMG_FP8_BWD=0: the bf16 backward of rounds 1 - 3): the incoming gradient is cast to e5m2
        # with a per-tensor power-of-two scale, both data-gradient GEMMs run on e5m2 x e4m3 operands (dh handed on in e5m2, written once), both
        # weight-gradient GEMMs on the 8-bit operands the forward / data path already hold (csrc/gemm_tn_fp8.hip)
        self.fp8_bwd = os.environ.get("MMG_FP8_BWD", "1") != "0"
        self.fp8_bwd_now = False             # decided per forward (_decide_save_ln): the 8-bit operands are kept only while they fit
        self.fp8_delayed = os.environ.get("MMG_FP8_DELAYED", "1") != "0"      # gradient scale from the previous quantisation of the same tensor role
        self._e5m2_state = {}
        # (round 4, with the 8-bit backward: 256 - same-box A/B of `bench.py --variant base --fp8 --checkpoint`: 925 ms/step from C = 256, 939 from 512,
        #  936 from 128: at C = 256 the GEMM pair's 4C-wide tensors are 8-bit in both directions now; rounds 1 - 3, forward only: no difference, 512)
        self.fp8_min_channels = int(os.environ.get("MMG_FP8_MIN_C", "256" if self.fp8_bwd else "512")) if fp8_min_channels is None else int(fp8_min_channels)
        self.dims, self.depths = CONFIGS[variant]["dims"], CONFIGS[variant]["depths"]
        self.model = _TorchvisionLayout(variant, in_chans)
        self.model_output_dimension = self.dims[-1]
        self.kp = (16 * in_chans + 31) // 32 * 32          # stem GEMM K padded to the MFMA k-step
        self._arena = None
        self._wc = None
        self._wc_version = None
        self._anchor = None
        self.post_backward_hook = None      # called with the arena once this tower's gradients are complete

    # ---- parameter plumbing ------------------------------------------------------------------------------
    def _materialize(self, device):
        if self._arena is not None and self._arena.device == device and self._arena.is_bound():
            return
        self._arena = ParamArena(list(self.model.named_parameters()), device)
        self._wc_version = None
        self._anchor = torch.zeros(1, device=device, requires_grad=True)

    @property
    def arena(self):
        return self._arena

    def _fp8_block(self, C):
        return self.fp8 and C % 128 == 0 and C >= self.fp8_min_channels

    def _refresh_working_copies(self):
        """bf16 / transposed / tap-major copies the kernels read; rebuilt only when a parameter changed."""
        A = self._arena
        v = A.version()
        if self._wc_version == v:
            return
        f = self.model.features
        wc = {}
        stem = f[0][0].weight.data                                   # [C0, Cin, 4, 4] -> [(kh,kw,ci)] padded
        w = torch.zeros(stem.shape[0], self.kp, device=stem.device)
        w[:, :16 * self.in_chans] = stem.permute(0, 2, 3, 1).reshape(stem.shape[0], -1)
        wc["stem.w"] = K.cast_bf16(w)
        for si in range(4):
            for bi, blk in enumerate(f[1 + 2 * si]):
                C = self.dims[si]
                key = f"{si}.{bi}"
                wc[key + ".w49"] = blk.block[0].weight.data.reshape(C, 49).t().contiguous()
                wc[key + ".w1"] = K.cast_bf16(blk.block[3].weight.data)                      # [4C, C]
                wc[key + ".w1t"] = K.transpose_cast_bf16(blk.block[3].weight.data)           # [C, 4C]
                wc[key + ".w2"] = K.cast_bf16(blk.block[5].weight.data)                      # [C, 4C]
                wc[key + ".w2gt"] = K.transpose_cast_bf16(blk.block[5].weight.data,          # [4C, C] * gamma
                                                          blk.layer_scale.data.reshape(C))
                if self._fp8_block(C):                                                       # e4m3 bytes + (scale, 1/scale)
                    wc[key + ".w1f8"], wc[key + ".s1"] = K.quantize_e4m3(blk.block[3].weight.data)
                    wc[key + ".w2f8"], wc[key + ".s2"] = K.quantize_e4m3(blk.block[5].weight.data)
                    if self.fp8_bwd:         # the data-gradient GEMMs' weights: (gamma W2)^T [4C, C] and W1^T [C, 4C], e4m3 + (scale, 1/scale)
                        wc[key + ".w2gt8"], wc[key + ".s2gt"] = K.quantize_e4m3(
                            (blk.block[5].weight.data * blk.layer_scale.data.reshape(C, 1)).t().contiguous())
                        wc[key + ".w1t8"], wc[key + ".s1t"] = K.quantize_e4m3(blk.block[3].weight.data.t().contiguous())
                elif self.fused_mlp and K.cnblock_supported(C):                              # packed LDS images
                    wc[key + ".mlp"] = K.cnblock_pack(blk.block[3].weight.data, blk.block[5].weight.data)
                    mode = K.cnblock_bwd_mode(C)     # 1: hidden row recomputed; 2: reads the forward's saved pre-activation
                    if mode == 2 and not self.fused_bwd_saved_h:   # (C=384: slower than the GEMM pair so far)
                        mode = 0
                    if mode:
                        wc[key + (".mlpb" if mode == 1 else ".mlpb2")] = K.cnblock_pack(
                            blk.block[3].weight.data, blk.block[5].weight.data, blk.layer_scale.data.reshape(C), backward=mode)
                    # round 3: backward with the weight gradients accumulated on chip (C = 96: nothing 4C-wide reaches HBM)
                    if mode == 1 and self.bwdw and K.cnblock_bwdw_supported(C, 64):
                        wc[key + ".bwdw"] = K.cnblock_bwdw_pack(blk.block[3].weight.data, blk.block[5].weight.data, blk.block[2].weight.data,
                                                                blk.block[2].bias.data, blk.layer_scale.data.reshape(C), blk.block[3].bias.data)
            if si < 3:
                conv = f[2 + 2 * si][1].weight.data                                          # [2C, C, 2, 2]
                wds = conv.permute(0, 2, 3, 1).reshape(conv.shape[0], -1).contiguous()       # [(kh,kw,ci)]
                wc[f"ds{si}.w"] = K.cast_bf16(wds)
                wc[f"ds{si}.wt"] = K.transpose_cast_bf16(wds)
        self._wc, self._wc_version = wc, v

    def _decide_save_ln(self, n_alive, H, W, device, ckpt=False):
        """n_alive = images whose saved tensors are alive at once (the whole batch; one micro-batch under checkpointing).
        ckpt: under checkpointing the saved tensors of ONE micro-batch are all the activation memory there is, so the optional copies may take a
        larger share of the device (round 4, ConvNeXt-B in micro-batches of 128: 264 against 249 pairs/s with them, peak 238 GiB)."""
        extra, hh, ww = 0, H // 4, W // 4
        for si in range(4):
            C = self.dims[si]
            if not (K.cnblock_supported(C) and K.cnblock_bwd_mode(C) == 1) and not (K.cnblock_bwd_mode(C) == 2 and self.fused_bwd_saved_h):
                extra += self.depths[si] * n_alive * hh * ww * C * 2
            hh, ww = hh // 2, ww // 2
        total = torch.cuda.get_device_properties(device).total_memory
        # 8-bit backward: its forward keeps the e4m3 LayerNorm output and activation (5 C bytes per row and block on top of the bf16 side output)
        # - only while that stays below 15 % of the device memory (C5, checkpointed micro-batches of 64: 23 GB, on; ConvNeXt-B at 256 images
        # without checkpointing: 93 GB on top of 267 GiB, off - that forward then saves what rounds 1 - 3 saved and its backward runs in bf16)
        keep8, hh, ww = 0, H // 4, W // 4
        for si in range(4):
            C = self.dims[si]
            if self.fp8 and self.fp8_bwd and C % 128 == 0 and C >= self.fp8_min_channels:
                keep8 += self.depths[si] * n_alive * hh * ww * C * 5
            hh, ww = hh // 2, ww // 2
        mode8 = os.environ.get("MMG_FP8_BWD", "auto")
        self.fp8_bwd_now = self.fp8 and self.fp8_bwd and (mode8 == "1" or keep8 <= (0.20 if ckpt else 0.15) * total)
        if self.save_ln_mode in ("0", "1"):
            ln = self.save_ln_mode == "1"
        else:
            ln = extra <= (0.10 if ckpt else 0.04) * total
        if self.save_gelu_mode in ("0", "1"):
            self.save_gelu = self.save_gelu_mode == "1"
        else:
            self.save_gelu = 4 * extra <= (0.30 if ckpt else 0.15) * total            # ([M,4C] against [M,C])
        return ln

    # ---- forward / backward over one micro-batch -----------------------------------------------------------
    def _forward_mb(self, img, save):
        f, wc = self.model.features, self._wc
        n, _, H, W = img.shape
        h, w_ = H // 4, W // 4
        saved = {}
        p0 = K.patchify(img, 4, self.kp, self.scale16)
        s0 = L.gemm_nt(p0, wc["stem.w"], bias=f[0][0].bias.data)
        x, mean, rstd = K.layernorm_fwd(s0, f[0][1].weight.data, f[0][1].bias.data, LN_EPS, want_stats=save)
        if save:
            saved["stem"] = (p0, s0, mean, rstd)
        for si in range(4):
            C = self.dims[si]
            for bi, blk in enumerate(f[1 + 2 * si]):
                key = f"{si}.{bi}"
                d = K.dwconv7(x, wc[key + ".w49"], blk.block[0].bias.data, n, h, w_, C)
                # LN + Linear + GELU + Linear + layer scale + residual in one launch (C = 512, ConvNeXt-B stage 3: only when nothing
                # is saved for a backward - with the 4C-wide pre-activation store it is no faster than the GEMM pair)
                # MMG_MLP_FUSED_SAVE_MAXC (A/B knob, round 4): widest block whose SAVING forward stays on the fused kernel - at C = 384 that kernel
                # stores three 4C- / C-wide streams beside its output and the GEMM pair (256 x 256 and 256 x 192 tiles) is within reach of it
                if key + ".mlp" in wc and (C <= int(os.environ.get("MMG_MLP_FUSED_SAVE_MAXC", "384")) or not save):
                    keep = save and key + ".mlpb" not in wc      # the fused backward recomputes the hidden row
                    # a GEMM-pair backward (C = 384 by default) also gets the LayerNorm output from the forward's registers: one
                    # [M,C] store instead of a LayerNorm pass over d in the backward
                    keep_ln = keep and self.save_ln and key + ".mlpb2" not in wc
                    # ... and (save_gelu) GELU(hidden) as the second GEMM consumed it: that backward's data-gradient GEMM then applies GELU' only
                    keep_g = keep and self.save_gelu and key + ".mlpb2" not in wc
                    # round 4: with GELU(hidden) kept, the second saved 4C-wide tensor is GELU'(hidden) instead of the hidden itself (same bytes): the
                    # backward's data-gradient GEMM then multiplies by it (NT epilogue 7) instead of evaluating the polynomial per element
                    outs = K.cnblock_mlp_fwd(d, blk.block[2].weight.data, blk.block[2].bias.data, LN_EPS, wc[key + ".mlp"],
                                             blk.block[3].bias.data, blk.block[5].bias.data, blk.layer_scale.data.reshape(C), x,
                                             want_hpre=keep, want_stats=keep, want_xln=keep_ln, want_gact=keep_g,
                                             hpre_kind=1 if (keep_g and self.save_dgelu) else 0)
                    xn, hpre, mean, rstd = outs[:4]
                    ln = outs[4] if keep_ln else None
                    gact = outs[-1] if keep_g else None
                    if save:
                        saved[key] = (x, d, mean, rstd, hpre, ln, gact)
                    x = xn
                    continue
                hpre = torch.empty(x.shape[0], 4 * C, device=x.device, dtype=torch.bfloat16) if save else None
                if key + ".w1f8" in wc:      # e4m3 operands, fp32 accumulate; the saved pre-activation stays bf16
                    ln, mean, rstd = K.layernorm_fwd_fp8(d, blk.block[2].weight.data, blk.block[2].bias.data, LN_EPS, want_stats=save)
                    # (8-bit backward: the side output is GELU'(h) - its data-gradient GEMM multiplies by it - and the e4m3 LayerNorm output /
                    #  activation are kept: they ARE the weight-gradient GEMMs' operands)
                    g = L.gemm_nt_fp8(ln, wc[key + ".w1f8"], bias=blk.block[3].bias.data, aux_out=hpre,
                                      epi=L.EPI_GELU_DAUX if (save and self.fp8_bwd_now and key + ".w2gt8" in wc) else L.EPI_GELU,
                                      out_kind=L.OUT_E4M3, alpha_dev=wc[key + ".s1"][1:])
                    xn = L.gemm_nt_fp8(g, wc[key + ".w2f8"], bias=blk.block[5].bias.data, colscale=blk.layer_scale.data.reshape(C),
                                       residual=x, alpha_dev=wc[key + ".s2"][1:])
                else:
                    ln, mean, rstd = K.layernorm_fwd(d, blk.block[2].weight.data, blk.block[2].bias.data, LN_EPS, want_stats=save)
                    g = L.gemm_nt(ln, wc[key + ".w1"], bias=blk.block[3].bias.data, aux_out=hpre,
                                  epi=L.EPI_GELU_DAUX if (save and self.save_gelu and self.save_dgelu) else L.EPI_GELU)   # (hpre = GELU'(h) then)
                    xn = L.gemm_nt(g, wc[key + ".w2"], bias=blk.block[5].bias.data, colscale=blk.layer_scale.data.reshape(C),
                                   residual=x)
                if save:                         # (an e4m3 LayerNorm output / activation is not what the bf16 backward reads: those are recomputed)
                    if self.fp8_bwd_now and key + ".w2gt8" in wc:
                        saved[key] = (x, d, mean, rstd, hpre, ln, g)         # (uint8 tensors: the 8-bit backward below)
                    else:
                        saved[key] = (x, d, mean, rstd, hpre, ln if (self.save_ln and ln.dtype == torch.bfloat16) else None,
                                      g if (self.save_gelu and g.dtype == torch.bfloat16) else None)
                del ln, g
                x = xn
            if si < 3:
                lnm = f[2 + 2 * si][0]
                ld, mean, rstd = K.layernorm_fwd(x, lnm.weight.data, lnm.bias.data, LN_EPS, patch_hw=(h, w_), want_stats=save)
                xn = L.gemm_nt(ld, wc[f"ds{si}.w"], bias=f[2 + 2 * si][1].bias.data)
                if save:
                    saved[f"ds{si}"] = (x, mean, rstd, ld)
                x = xn
                h, w_ = h // 2, w_ // 2
        feat = K.avgpool_fwd(x, n, h * w_, self.dims[-1])
        saved["shape"] = (n, H, W)
        return feat, saved

    def _backward_mb(self, dfeat, saved, tmp, final=False, announce=False):
        """final: last micro-batch of this backward - the GEMM-shaped temporaries of a stage are folded into the torch-layout
        gradients as soon as the stage has passed; announce: it is also the tower's last backward of the step, so the stage's
        gradients are complete and are handed to the gradient all-reduce (ParamArena.mark_ready) while the earlier, larger
        feature maps are still in backward."""
        f, wc, A = self.model.features, self._wc, self._arena
        n, H, W = saved["shape"]
        h, w_ = H // 32, W // 32
        gname = lambda mod, leaf: A.g(self._pname[id(mod)] + "." + leaf)      # noqa: E731
        dx = K.avgpool_bwd(dfeat, n, h * w_, self.dims[-1])
        for si in range(3, -1, -1):
            C = self.dims[si]
            if si < 3:
                x, mean, rstd, ld = saved[f"ds{si}"]
                conv, lnm = f[2 + 2 * si][1], f[2 + 2 * si][0]
                L.gemm_tn_acc(dx, ld, tmp[f"ds{si}.dw"], colsum=gname(conv, "bias"))
                dld = L.gemm_nt(dx, wc[f"ds{si}.wt"])
                h, w_ = h * 2, w_ * 2
                dx = K.layernorm_bwd(dld, x, mean, rstd, lnm.weight.data, gname(lnm, "weight"), gname(lnm, "bias"),
                                     patch_hw=(h, w_))
                del dld
            for bi in range(self.depths[si] - 1, -1, -1):
                blk = f[1 + 2 * si][bi]
                key = f"{si}.{bi}"
                x, d, mean, rstd, hpre, ln_saved, g_saved = saved[key]
                if hpre is None and key + ".bwdw" in wc and K.cnblock_bwdw_supported(C, d.shape[0]):
                    # stage 1: data path AND both weight gradients in two launches that read dx, d and write dd - the g / dh tensors
                    # ([M,4C] each) of the path below and its two weight-gradient GEMMs do not exist
                    packed, b1f = wc[key + ".bwdw"]
                    dd = K.cnblock_bwdw(dx, d, blk.block[2].weight.data, blk.block[2].bias.data, LN_EPS, packed, b1f,
                                        gname(blk.block[3], "weight"), gname(blk.block[3], "bias"), tmp[key + ".dw2raw"], tmp[key + ".db2raw"],
                                        gname(blk.block[2], "weight"), gname(blk.block[2], "bias"))
                    dln = None
                elif hpre is None or key + ".mlpb2" in wc:   # fused data path (hidden row recomputed on chip / read back)
                    # C <= 128: the LayerNorm backward rides in the epilogue (`dd` comes back instead of d LN-out); wider
                    # blocks have no registers left for it
                    fuse_ln = C <= 128
                    dh, g, ln, dln, mean, rstd = K.cnblock_mlp_bwd(
                        dx, d, blk.block[2].weight.data, blk.block[2].bias.data, LN_EPS,
                        wc[key + (".mlpb" if hpre is None else ".mlpb2")], blk.block[3].bias.data, hpre,
                        ln_grads=(gname(blk.block[2], "weight"), gname(blk.block[2], "bias")) if fuse_ln else None)
                    if fuse_ln:
                        dd, dln = dln, None
                    L.gemm_tn_acc(dx, g, tmp[key + ".dw2raw"], colsum=tmp[key + ".db2raw"])
                    del g
                    L.gemm_tn_acc(dh, ln, gname(blk.block[3], "weight"), colsum=gname(blk.block[3], "bias"))
                    del ln, dh
                elif g_saved is not None and g_saved.dtype == torch.uint8:
                    # 8-bit backward (config C5): hpre holds GELU'(h) (bf16), ln_saved / g_saved the e4m3 operands of the forward GEMMs
                    # (delayed scaling from the second use on: the scale of this block's gradient comes from its previous quantisation - one pass)
                    # (+ the bias gradient of the second Linear = column sums of the bf16 gradient itself, in the same pass over it)
                    dy8, sdy = K.quantize_e5m2(dx, self._e5m2_state.setdefault(key, {}) if self.fp8_delayed else None, colsum=tmp[key + ".db2raw"])
                    dh8 = L.gemm_nt_fp8_bwd(dy8, wc[key + ".w2gt8"], aux_in=hpre, epi=L.EPI_MUL_AUX, out_kind=L.OUT_E5M2,
                                            alpha_dev=wc[key + ".s2gt"][1:])               # e5m2 at dy's scale: (acc / s_w) * GELU'
                    L.gemm_tn_fp8_acc(dy8, g_saved, tmp[key + ".dw2raw"], alpha_dev=sdy[1:])
                    L.gemm_tn_fp8_acc(dh8, ln_saved, gname(blk.block[3], "weight"), alpha_dev=sdy[1:], colsum=gname(blk.block[3], "bias"))
                    dln = L.gemm_nt_fp8_bwd(dh8, wc[key + ".w1t8"], alpha_dev=wc[key + ".s1t"][1:], alpha_dev2=sdy[1:])
                    del dh8, dy8
                else:
                    if g_saved is not None:                # the forward kept GELU(h): GELU' only (half the epilogue's arithmetic and stores)
                        g = g_saved
                        # ... and (save_dgelu) GELU'(h) in place of h: one multiply per element
                        dh = L.gemm_nt(dx, wc[key + ".w2gt"], epi=L.EPI_MUL_AUX if self.save_dgelu else L.EPI_DGELU_ONLY, aux_in=hpre)
                    else:
                        g = torch.empty_like(hpre)         # GELU(hpre), rebuilt by the same epilogue that applies GELU'
                        dh = L.gemm_nt(dx, wc[key + ".w2gt"], epi=L.EPI_DGELU, aux_in=hpre, aux_out=g)
                    L.gemm_tn_acc(dx, g, tmp[key + ".dw2raw"], colsum=tmp[key + ".db2raw"])
                    del g
                    ln = ln_saved if ln_saved is not None else \
                        K.layernorm_fwd(d, blk.block[2].weight.data, blk.block[2].bias.data, LN_EPS, want_stats=False)[0]
                    L.gemm_tn_acc(dh, ln, gname(blk.block[3], "weight"), colsum=gname(blk.block[3], "bias"))
                    del ln
                    dln = L.gemm_nt(dh, wc[key + ".w1t"])
                    del dh
                if dln is not None:
                    dd = K.layernorm_bwd(dln, d, mean, rstd, blk.block[2].weight.data, gname(blk.block[2], "weight"),
                                         gname(blk.block[2], "bias"))
                del dln
                K.dwconv7_wgrad(x, dd, tmp[key + ".dw49"], gname(blk.block[0], "bias"), n, h, w_, C)
                dx = K.dwconv7(dd, wc[key + ".w49"], None, n, h, w_, C, add=dx, flip=True)
                del dd
            if final:
                self._finalize_stage(tmp, si)
                if announce:
                    A.mark_ready((f"features.{1 + 2 * si}.",) + ((f"features.{2 + 2 * si}.",) if si < 3 else ()))
        p0, s0, mean, rstd = saved["stem"]
        ds0 = K.layernorm_bwd(dx, s0, mean, rstd, f[0][1].weight.data, gname(f[0][1], "weight"), gname(f[0][1], "bias"))
        L.gemm_tn_acc(ds0, p0, tmp["stem.dw"], colsum=gname(f[0][0], "bias"))
        if final:
            self._finalize_stem(tmp)

    def _alloc_tmp(self, device):
        z = lambda *s: torch.zeros(*s, device=device, dtype=torch.float32)   # noqa: E731
        tmp = {"stem.dw": z(self.dims[0], self.kp)}
        for si in range(4):
            C = self.dims[si]
            for bi in range(self.depths[si]):
                key = f"{si}.{bi}"
                tmp[key + ".dw2raw"], tmp[key + ".db2raw"], tmp[key + ".dw49"] = z(C, 4 * C), z(C), z(49, C)
            if si < 3:
                tmp[f"ds{si}.dw"] = z(self.dims[si + 1], 4 * C)
        return tmp

    def _finalize_stem(self, tmp):
        """Fold the GEMM-shaped stem temporary into the torch-layout gradient."""
        f, A = self.model.features, self._arena
        from .._hip import call, ptr, stream
        stem = f[0][0]
        # the stem temp is [C0, Kp] with zero-padded tail columns: relayout the first 16*Cin columns
        kk = 16 * self.in_chans
        src = tmp["stem.dw"][:, :kk].contiguous()
        call("mmg_grad_relayout", ptr(src), ptr(A.g(self._pname[id(stem)] + ".weight")), 0, self.dims[0], self.in_chans, 4, 4, kk,
             stream())

    def _finalize_stage(self, tmp, si):
        """Fold stage si's GEMM-shaped temporaries (and those of the downsample layer behind it) into the torch-layout gradients
        (layer scale, conv layouts)."""
        f, A = self.model.features, self._arena
        from .._hip import call, ptr, stream
        gname = lambda mod, leaf: A.g(self._pname[id(mod)] + "." + leaf)      # noqa: E731
        C = self.dims[si]
        for bi, blk in enumerate(f[1 + 2 * si]):
            key = f"{si}.{bi}"
            call("mmg_layerscale_finalize", ptr(blk.block[5].weight.data), ptr(blk.block[5].bias.data),
                 ptr(blk.layer_scale.data), ptr(tmp[key + ".dw2raw"]), ptr(tmp[key + ".db2raw"]),
                 ptr(gname(blk.block[5], "weight")), ptr(gname(blk.block[5], "bias")),
                 ptr(A.g(self._pname[id(blk)] + ".layer_scale")), C, 4 * C, stream())
            call("mmg_grad_relayout", ptr(tmp[key + ".dw49"]), ptr(gname(blk.block[0], "weight")), 1, C, 1, 7, 7, C,
                 stream())
        if si < 3:
            conv = f[2 + 2 * si][1]
            call("mmg_grad_relayout", ptr(tmp[f"ds{si}.dw"]), ptr(gname(conv, "weight")), 0, self.dims[si + 1], C, 2, 2,
                 4 * C, stream())

    # ---- public -----------------------------------------------------------------------------------------------
    def feature_map_shape(self, H, W):
        """(h, w) of the last stage for an H x W input (floor at every stride, e.g. 1906 x 818 -> 59 x 25)."""
        h, w = H // 4, W // 4
        for _ in range(3):
            h, w = h // 2, w // 2
        return h, w

    def forward(self, images):
        _hip.require_gpu(images)
        if images.shape[-2] < 32 or images.shape[-1] < 32:
            raise ValueError(f"ConvNeXt needs at least 32x32 pixels, got {tuple(images.shape)}")
        self._materialize(images.device)
        self._pname = {id(m): "features." + n for n, m in self.model.features.named_modules()}
        needs_grad = torch.is_grad_enabled() and self._arena.any_trainable()
        if needs_grad and (images.shape[-2] % 32 or images.shape[-1] % 32):
            raise NotImplementedError("training the ConvNeXt tower needs H and W to be multiples of 32 (inference accepts any "
                                      "size >= 32: strided layers drop the remainder exactly as torch's convolutions do)")
        note_forward(self, needs_grad)
        return _ConvNextFn.apply(self, images.float().contiguous(), stream_anchor(self, self._anchor.device) if needs_grad else None)


class _ConvNextFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tower, images, anchor):
        tower._refresh_working_copies()
        save = anchor is not None
        feats, saved = [], []
        mb = tower.micro_batch
        ckpt = save and tower.checkpoint
        if save:
            tower.save_ln = tower._decide_save_ln(min(mb, images.shape[0]) if ckpt else images.shape[0], images.shape[-2], images.shape[-1],
                                                  images.device, ckpt=ckpt and mb < images.shape[0])
        for i in range(0, images.shape[0], mb):
            # gradient checkpointing at micro-batch granularity: keep only the pixels, re-run the micro-batch's forward (with its
            # activations saved) right before its backward - activation memory becomes one micro-batch instead of the whole batch.
            # The LAST micro-batch keeps its activations (round 4): the backward starts with it (reverse order), so still only one
            # micro-batch's activations are alive at any time, and one of the n recomputations is not run (MMG_CKPT_KEEP_LAST=0: all are).
            keep = ckpt and i + mb >= images.shape[0] and os.environ.get("MMG_CKPT_KEEP_LAST", "1") != "0"
            ft, sv = tower._forward_mb(images[i:i + mb], save and (not ckpt or keep))
            feats.append(ft)
            saved.append({"recompute": images[i:i + mb]} if (ckpt and not keep) else sv)
        ctx.tower, ctx.saved_mb, ctx.reverse = tower, saved if save else None, ckpt
        return torch.cat(feats, 0) if len(feats) > 1 else feats[0]

    @staticmethod
    def backward(ctx, dfeat):
        tower = ctx.tower
        tower._arena.prepare_grads()
        tmp = tower._alloc_tmp(dfeat.device)
        dfeat = dfeat.float().contiguous()
        sizes = [sv["recompute"].shape[0] if "recompute" in sv else sv["shape"][0] for sv in ctx.saved_mb]
        starts = [sum(sizes[:k]) for k in range(len(sizes))]
        order = list(range(len(sizes)))
        if ctx.reverse:                      # checkpointing: the micro-batch whose activations were kept (the last one) first
            order.reverse()
        for pos, k in enumerate(order):
            sv = ctx.saved_mb[k]
            if "recompute" in sv:
                _, sv = tower._forward_mb(sv["recompute"], True)
            i, n = starts[k], sizes[k]
            final = pos == len(order) - 1
            tower._backward_mb(dfeat[i:i + n].contiguous(), sv, tmp, final=final, announce=final and last_backward(tower))
            sv.clear()
            ctx.saved_mb[k] = None
        ctx.saved_mb = None
        backward_finished(tower)
        return None, None, None
