"""ResNet-50 image tower on the HIP kernels (reference: mmgclip/networks/encoder.py:57-119).

The reference wraps torchvision's `resnet50(pretrained=True)` without its `fc`, freezes every parameter except `layer4`
(:77-89) and runs conv1 -> bn1 -> relu -> maxpool -> layer1..4 -> avgpool -> flatten (:105-117); a 2-D input [B, L] (the
precomputed feature vector that `MMGCLIP.encode_images` hands over) is viewed as a 1 x L image and repeated to 3 channels
(:101-103).  Module tree and state-dict keys follow torchvision (conv1, bn1, layer{1..4}.{i}.conv{1,2,3}, bn{1,2,3},
downsample.{0,1}), so a torchvision checkpoint loads unchanged (no network here: random initialisation otherwise).

Data layout: NHWC bf16 rows [n*H*W, C]; a 1x1 convolution is a plain GEMM on those rows, k x k convolutions go through an
explicit column matrix (mmg_im2col_nhwc) into the same MFMA GEMM; BatchNorm (+ shortcut add + ReLU) is one streaming pass
over per-channel statistics.  BatchNorm follows the module's mode exactly like torch: batch statistics and running-average
updates under .train() - also in the frozen layers, as in the reference, which calls model.train() on the whole model - and
running statistics under .eval().  Only layer4 has a backward (its input is not differentiated: layer3 is frozen)."""
import torch
import torch.nn as nn

from .. import _hip
from .. import kernels as K
from .. import linalg as L
from ..params import ParamArena, backward_finished, note_forward, stream_anchor
from .._hip import call, ptr, stream

BN_EPS, BN_MOMENTUM = 1e-5, 0.1
LAYERS = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))       # (width, blocks, stride of the first block)


class Bottleneck(nn.Module):
    """Parameter container with torchvision's names; the arithmetic runs in ResNetTower."""

    def __init__(self, cin, width, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, 4 * width, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(4 * width)
        self.downsample = nn.Sequential(nn.Conv2d(cin, 4 * width, 1, stride=stride, bias=False), nn.BatchNorm2d(4 * width)) \
            if downsample else None
        self.stride = stride


class _TorchvisionResNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for li, (width, blocks, stride) in enumerate(LAYERS):
            mods = []
            for b in range(blocks):
                mods.append(Bottleneck(cin, width, stride if b == 0 else 1, downsample=(b == 0)))
                cin = 4 * width
            setattr(self, f"layer{li + 1}", nn.Sequential(*mods))
        for m in self.modules():                        # torchvision's initialisation
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")


class ResNetTower(nn.Module):
    def __init__(self):
        super().__init__()
        self.model = _TorchvisionResNet()
        self.model_output_dimension = 2048
        for p in self.model.parameters():               # encoder.py:77-89
            p.requires_grad = False
        for p in self.model.layer4.parameters():
            p.requires_grad = True
        self._arena = None
        self._wc, self._wc_version = None, None
        self._anchor = None
        self.post_backward_hook = None

    # ---- parameters -----------------------------------------------------------------------------------------------------
    @property
    def arena(self):
        return self._arena

    def _materialize(self, device):
        if self._arena is not None and self._arena.device == device and self._arena.is_bound():
            return
        named = [("layer4." + n, p) for n, p in self.model.layer4.named_parameters()]
        self._arena = ParamArena(named, device)           # the trainable part: one buffer, one AdamW launch, one all-reduce
        self._wc_version = None
        self._anchor = torch.zeros(1, device=device, requires_grad=True)

    @staticmethod
    def _w2d(conv, cin_pad=None):
        """[Cout, Cin, kh, kw] -> [Cout, (kh, kw, ci)] fp32 (input channels zero-padded to cin_pad), K padded to 32."""
        w = conv.weight.data
        co, ci, kh, kw = w.shape
        if cin_pad and cin_pad > ci:
            w = torch.cat([w, torch.zeros(co, cin_pad - ci, kh, kw, device=w.device)], 1)
            ci = cin_pad
        w = w.permute(0, 2, 3, 1).reshape(co, kh * kw * ci)
        kp = (w.shape[1] + 31) // 32 * 32
        if kp != w.shape[1]:
            w = torch.cat([w, torch.zeros(co, kp - w.shape[1], device=w.device)], 1)
        return w.contiguous()

    def _refresh_working_copies(self):
        v = (self._arena.version(), sum(p._version for p in self.model.parameters()))
        if self._wc_version == v:
            return
        wc = {"conv1": K.cast_bf16(self._w2d(self.model.conv1, 8))}
        for li in range(4):
            for bi, blk in enumerate(getattr(self.model, f"layer{li + 1}")):
                key = f"{li}.{bi}."
                for name in ("conv1", "conv2", "conv3"):
                    w2 = self._w2d(getattr(blk, name))
                    wc[key + name] = K.cast_bf16(w2)
                    if li == 3:                           # data-gradient operands (layer4 only)
                        wc[key + name + "t"] = K.transpose_cast_bf16(w2)
                if blk.downsample is not None:
                    wc[key + "ds"] = K.cast_bf16(self._w2d(blk.downsample[0]))
        self._wc, self._wc_version = wc, v

    # ---- pieces -----------------------------------------------------------------------------------------------------------
    def _bn(self, x, bn, residual=None, relu=True):
        train = bn.training
        if train and bn.num_batches_tracked is not None:
            bn.num_batches_tracked += 1
        return K.batchnorm_fwd(x, bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var, train, bn.eps, bn.momentum,
                               residual=residual, relu=relu)

    def _block_fwd(self, x, n, H, W, blk, key, save):
        wc = self._wc
        cin, width, s = blk.conv1.in_channels, blk.conv1.out_channels, blk.stride
        x1 = L.gemm_nt(x, wc[key + "conv1"])
        a1, m1, r1 = self._bn(x1, blk.bn1)
        Ho, Wo = K.conv_out_hw(H, W, 3, s, 1)
        col2 = K.im2col(a1, n, H, W, width, 3, s, 1, 9 * width)
        x2 = L.gemm_nt(col2, wc[key + "conv2"])
        del col2
        a2, m2, r2 = self._bn(x2, blk.bn2)
        x3 = L.gemm_nt(a2, wc[key + "conv3"])
        if blk.downsample is not None:
            xs = x if s == 1 else K.im2col(x, n, H, W, cin, 1, s, 0, cin)
            xd = L.gemm_nt(xs, wc[key + "ds"])
            idn, md, rd = self._bn(xd, blk.downsample[1], relu=False)
        else:
            xs = xd = md = rd = None
            idn = x
        out, m3, r3 = self._bn(x3, blk.bn3, residual=idn, relu=True)
        saved = dict(x=x, xs=xs, x1=x1, a1=a1, x2=x2, a2=a2, x3=x3, xd=xd, out=out, st=(m1, r1, m2, r2, m3, r3, md, rd),
                     geom=(n, H, W, Ho, Wo)) if save else None
        return out, Ho, Wo, saved

    def _block_bwd(self, dout, sv, blk, key, need_dx):
        wc, A = self._wc, self._arena
        g = lambda name: A.g("layer4." + key.split(".", 1)[1] + name)      # noqa: E731  (key = "3.<bi>.")
        n, H, W, Ho, Wo = sv["geom"]
        m1, r1, m2, r2, m3, r3, md, rd = sv["st"]
        cin, width, s = blk.conv1.in_channels, blk.conv1.out_channels, blk.stride
        # bn3 (+ shortcut add + ReLU)
        dx3, dres = K.batchnorm_bwd(dout, sv["x3"], sv["out"], m3, r3, blk.bn3.weight.data, g("bn3.weight"), g("bn3.bias"), want_dres=True)
        L.gemm_tn_acc(dx3, sv["a2"], g("conv3.weight").view(4 * width, width))
        da2 = L.gemm_nt(dx3, wc[key + "conv3t"])
        del dx3
        dx2, _ = K.batchnorm_bwd(da2, sv["x2"], sv["a2"], m2, r2, blk.bn2.weight.data, g("bn2.weight"), g("bn2.bias"))
        del da2
        col2 = K.im2col(sv["a1"], n, H, W, width, 3, s, 1, 9 * width)
        tmp = torch.zeros(width, 9 * width, device=dout.device, dtype=torch.float32)
        L.gemm_tn_acc(dx2, col2, tmp)
        del col2
        call("mmg_grad_relayout", ptr(tmp), ptr(g("conv2.weight")), 0, width, width, 3, 3, 9 * width, stream())
        dcol = L.gemm_nt(dx2, wc[key + "conv2t"])
        del dx2
        da1 = K.col2im(dcol, n, H, W, width, 3, s, 1)
        del dcol
        dx1, _ = K.batchnorm_bwd(da1, sv["x1"], sv["a1"], m1, r1, blk.bn1.weight.data, g("bn1.weight"), g("bn1.bias"))
        del da1
        L.gemm_tn_acc(dx1, sv["x"], g("conv1.weight").view(width, cin))
        dx = None
        if blk.downsample is not None:
            dxd, _ = K.batchnorm_bwd(dres, sv["xd"], None, md, rd, blk.downsample[1].weight.data, g("downsample.1.weight"),
                                     g("downsample.1.bias"))
            L.gemm_tn_acc(dxd, sv["xs"], g("downsample.0.weight").view(4 * width, cin))
            if need_dx:
                raise NotImplementedError("gradient w.r.t. the input of a down-sampling block is not needed (layer3 is frozen)")
        elif need_dx:
            dx = L.gemm_nt(dx1, wc[key + "conv1t"], residual=dres)      # main path + identity shortcut
        return dx

    # ---- forward / backward -----------------------------------------------------------------------------------------------
    def _forward_impl(self, x, save):
        m, wc = self.model, self._wc
        n, _, H, W = x.shape
        xin = torch.zeros(n, H, W, 8, device=x.device, dtype=torch.bfloat16)      # NHWC, channels padded 3 -> 8 (layout plumbing)
        xin[..., :3] = x.permute(0, 2, 3, 1)
        col = K.im2col(xin.view(-1, 8), n, H, W, 8, 7, 2, 3, wc["conv1"].shape[1])
        H, W = K.conv_out_hw(H, W, 7, 2, 3)
        h = L.gemm_nt(col, wc["conv1"])
        del col
        h, _, _ = self._bn(h, m.bn1)
        h2 = K.maxpool3x3s2(h, n, H, W, 64)
        H, W = K.conv_out_hw(H, W, 3, 2, 1)
        h = h2
        saved = []
        for li in range(4):
            for bi, blk in enumerate(getattr(m, f"layer{li + 1}")):
                h, H, W, sv = self._block_fwd(h, n, H, W, blk, f"{li}.{bi}.", save and li == 3)
                if sv is not None:
                    saved.append(sv)
        feat = K.avgpool_fwd(h, n, H * W, 2048)
        return feat, (saved, n, H * W)

    def forward(self, x):
        _hip.require_gpu(x)
        if x.dim() == 2:                                   # encoder.py:101-103
            x = x.view(x.shape[0], 1, 1, x.shape[1]).repeat(1, 3, 1, 1)
        self._materialize(x.device)
        needs_grad = torch.is_grad_enabled() and self.training and self._arena.any_trainable()
        note_forward(self, needs_grad)
        return _ResNetFn.apply(self, x.float().contiguous(), stream_anchor(self, self._anchor.device) if needs_grad else None)


class _ResNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tower, x, anchor):
        tower._refresh_working_copies()
        save = anchor is not None
        feat, state = tower._forward_impl(x, save)
        ctx.tower, ctx.state = tower, state if save else None
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        tower = ctx.tower
        saved, n, hw = ctx.state
        tower._arena.prepare_grads()
        dh = K.avgpool_bwd(dfeat.float().contiguous(), n, hw, 2048)
        blocks = list(tower.model.layer4)
        for bi in range(len(blocks) - 1, -1, -1):
            dh = tower._block_bwd(dh, saved[bi], blocks[bi], f"3.{bi}.", need_dx=bi > 0)
        ctx.state = None
        backward_finished(tower)
        return None, None, None
