"""Projection heads with the reference's class names, constructor signatures and state-dict keys
(mmgclip/networks/projection.py:4-33, 36-61, 85-101); the arithmetic runs on the HIP kernels (mmgclip/ops.py)."""
import torch
from torch import nn

from .. import ops


class LinearProjectionLayer(nn.Module):
    """Bias-free linear map; `dropout` is accepted and ignored, as in the reference (projection.py:15-17)."""

    def __init__(self, embedding_dim, projection_dim=512, dropout=0):
        super().__init__()
        self.layer = nn.Linear(embedding_dim, projection_dim, bias=False)

    def forward(self, x):
        return ops.linear(x, self.layer.weight)


class MultiLinearHead(nn.Module):
    """Linear(+bias) chain with ReLU + Dropout between layers, none after the last (projection.py:37-61)."""

    def __init__(self, embedding_dim, projection_dim=[], dropout=0.5):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.projection_dim = projection_dim
        dims = [embedding_dim] + list(projection_dim)
        self.layers = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)])
        self.dropout = nn.Dropout(dropout)
        self.relu = nn.ReLU()

    def forward(self, x):
        last = len(self.layers) - 1
        for i, layer in enumerate(self.layers):
            x = ops.linear(x, layer.weight, layer.bias, act=None if i == last else "relu")
            if i < last:
                x = ops.dropout(x, self.dropout.p, self.training)
        return x


class MLPProjectionHead(nn.Module):
    """p = Linear(x); LayerNorm(Dropout(Linear(GELU(p))) + p)  (projection.py:86-101)."""

    def __init__(self, embedding_dim, projection_dim, dropout=0.5):
        super().__init__()
        self.projection = nn.Linear(embedding_dim, projection_dim)
        self.gelu = nn.GELU()
        self.fc = nn.Linear(projection_dim, projection_dim)
        self.dropout = nn.Dropout(dropout)
        self.layer_norm = nn.LayerNorm(projection_dim)

    def forward(self, x):
        projected = ops.linear(x, self.projection.weight, self.projection.bias)
        h = _gelu(projected)
        h = ops.linear(h, self.fc.weight, self.fc.bias)
        h = ops.dropout(h, self.dropout.p, self.training)
        h = h + projected
        return ops.layer_norm(h, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps)


class _Gelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        from .. import kernels as K
        x16 = K.cast_bf16(x.float().contiguous())
        ctx.save_for_backward(x16)
        return K.cast_f32(K.gelu(x16))

    @staticmethod
    def backward(ctx, dy):
        from .. import kernels as K
        from .._hip import call, ptr, stream
        (x16,) = ctx.saved_tensors
        dy16 = K.cast_bf16(dy.float().contiguous())
        out = torch.empty_like(dy16)
        call("mmg_act_grad_bf16", ptr(dy16), ptr(x16), ptr(out), dy16.numel(), 0, stream())
        return K.cast_f32(out)


def _gelu(x):
    return _Gelu.apply(x)
