"""Autograd building blocks for the projection heads: Linear (+ReLU/GELU), LayerNorm, Dropout on the HIP kernels.

fp32 tensors at the boundary (the heads are the reference's only trainable weights and feed the fp32 contrastive
head); the matrix products run on the bf16 MFMA path with fp32 accumulation.
"""
import torch

from . import _hip
from . import kernels as K
from . import linalg as L
from ._hip import call, ptr, stream

_ACT = {None: L.EPI_NONE, "relu": L.EPI_RELU, "gelu": L.EPI_GELU}


class LinearAct(torch.autograd.Function):
    """y = act(x @ W.T + b) — nn.Linear (+ ReLU / GELU) of mmgclip/networks/projection.py:33,55-61,94-96."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        _hip.require_gpu(x, weight)
        x16 = K.cast_bf16(x.float().contiguous())
        w16 = K.cast_bf16(weight.detach().float().contiguous())
        b = bias.detach().float().contiguous() if bias is not None else None
        pre = torch.empty(x.shape[0], weight.shape[0], device=x.device, dtype=torch.bfloat16) if act else None
        y = L.gemm_nt(x16, w16, bias=b, epi=_ACT[act], aux_out=pre, out_dtype=torch.float32)
        ctx.save_for_backward(x16, weight, pre if act else torch.empty(0))
        ctx.act, ctx.has_bias = act, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x16, weight, pre = ctx.saved_tensors
        dy16 = K.cast_bf16(dy.float().contiguous())
        if ctx.act:                                                  # dpre = dy * act'(pre)
            gated = torch.empty_like(dy16)
            call("mmg_act_grad_bf16", ptr(dy16), ptr(pre), ptr(gated), dy16.numel(), 0 if ctx.act == "gelu" else 1, stream())
            dy16 = gated
        wt16 = K.transpose_cast_bf16(weight.detach().float().contiguous())      # [K, N]
        dx = L.gemm_nt(dy16, wt16, out_dtype=torch.float32)
        dw = torch.zeros(weight.shape, device=dy.device, dtype=torch.float32)
        L.gemm_tn_acc(dy16, x16, dw)
        db = None
        if ctx.has_bias:
            db = torch.zeros(weight.shape[0], device=dy.device, dtype=torch.float32)
            L.colsum_acc(dy16, db)
        return dx, dw, db, None


def linear(x, weight, bias=None, act=None):
    return LinearAct.apply(x, weight, bias, act)


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        _hip.require_gpu(x)
        x16 = K.cast_bf16(x.float().contiguous())
        w, b = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        y16, mean, rstd = K.layernorm_fwd(x16, w, b, eps)
        ctx.save_for_backward(x16, mean, rstd, w)
        return K.cast_f32(y16)

    @staticmethod
    def backward(ctx, dy):
        x16, mean, rstd, w = ctx.saved_tensors
        dg, db = torch.zeros_like(w), torch.zeros_like(w)
        dx16 = K.layernorm_bwd(K.cast_bf16(dy.float().contiguous()), x16, mean, rstd, w, dg, db)
        return K.cast_f32(dx16), dg, db, None


def layer_norm(x, weight, bias, eps=1e-5):
    return LayerNormFn.apply(x, weight, bias, eps)


class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        x = x.float().contiguous()
        y = torch.empty_like(x)
        keep = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
        call("mmg_dropout_fwd", ptr(x), ptr(y), ptr(keep), x.numel(), float(p), int(seed), stream())
        ctx.save_for_backward(keep)
        ctx.p = p
        return y

    @staticmethod
    def backward(ctx, dy):
        (keep,) = ctx.saved_tensors
        dy = dy.float().contiguous()
        dx = torch.empty_like(dy)
        call("mmg_dropout_bwd", ptr(dy), ptr(keep), ptr(dx), dy.numel(), float(ctx.p), stream())
        return dx, None, None


_dropout_calls = [0]


def dropout(x, p, training):
    if not training or p <= 0.0:
        return x
    _dropout_calls[0] += 1
    seed = (torch.initial_seed() * 1000003 + _dropout_calls[0]) & 0x7FFFFFFFFFFFFFFF
    return DropoutFn.apply(x, p, seed)
