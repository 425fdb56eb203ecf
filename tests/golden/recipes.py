"""Seeded weight / input recipes shared by the golden-vector generator and the tests (test infrastructure).

The encoder fixtures (g5_*, g9_*) hold only OUTPUTS of third-party implementations (transformers' ConvNextModel, ViTModel,
ResNetModel, BertModel) plus a seed: the weights (up to 108 M numbers) are re-created on the test side from the same
recipe.  numpy's PCG64 stream is used (stable across numpy versions by policy) and every tensor gets its own stream keyed by
(seed, crc32(name)), so the values do not depend on iteration order or on which other tensors exist.

The init is deliberately NOT the training init: layer scales of 0.1-0.3 (torchvision starts at 1e-6, which would switch every
block off), fan-in scaled weights (O(1) activations, non-uniform attention), non-trivial LayerNorm / BatchNorm affine
parameters and running statistics - every term of every formula then matters to the result.
"""
import zlib

import numpy as np
import torch


def _rng(seed, name):
    return np.random.Generator(np.random.PCG64([int(seed), zlib.crc32(name.encode())]))


def seeded_tensor(name, shape, seed):
    """The recipe: value of tensor `name` (a state-dict key) with `shape` under `seed`, fp32."""
    r = _rng(seed, name)
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    n = lambda s: r.standard_normal(shape, dtype=np.float32) * np.float32(s)      # noqa: E731
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if leaf in ("layer_scale", "layer_scale_parameter"):
        v = r.uniform(0.1, 0.3, shape).astype(np.float32)
    elif leaf == "running_mean":
        v = n(0.1)
    elif leaf == "running_var":
        v = r.uniform(0.5, 1.5, shape).astype(np.float32)
    elif leaf in ("class_token", "cls_token", "pos_embedding", "position_embeddings") or "embeddings." in name and len(shape) == 2:
        v = n(0.05)
    elif len(shape) == 1 and leaf == "weight":                   # LayerNorm / BatchNorm gamma
        v = 1.0 + n(0.1)
    elif len(shape) == 1:                                        # every bias (incl. in_proj_bias)
        v = n(0.05)
    else:                                                        # Linear / Conv / in_proj_weight: fan-in scaled
        fan_in = int(np.prod(shape[1:]))
        v = n(1.0 / np.sqrt(fan_in))
    return torch.from_numpy(np.ascontiguousarray(v))


def fill_(module, seed, prefix=""):
    """Overwrite every parameter and buffer of `module` with the recipe (keys as in module.state_dict())."""
    with torch.no_grad():
        for k, t in module.state_dict().items():
            t.copy_(seeded_tensor(prefix + k, t.shape, seed))
    return module


def structured_images(n, size, seed, in_chans=1):
    """[n, in_chans, size, size] in [0,1): per-image smooth pattern (own frequency, phase, brightness) + noise, so that
    different images give DIFFERENT features (i.i.d. noise images all look alike to an encoder: cosine ~ 1, logits flat)."""
    r = _rng(seed, f"images{n}x{size}")
    yy, xx = np.meshgrid(np.linspace(0, 1, size, dtype=np.float32), np.linspace(0, 1, size, dtype=np.float32), indexing="ij")
    out = np.empty((n, in_chans, size, size), np.float32)
    for i in range(n):
        fx, fy = r.uniform(0.5, 6.0, 2)
        ph = r.uniform(0, 2 * np.pi)
        bright, amp = r.uniform(0.2, 0.8), r.uniform(0.1, 0.2)
        pat = bright + amp * np.sin(2 * np.pi * (fx * xx + fy * yy) + ph)
        for c in range(in_chans):
            noise = r.random((size, size), dtype=np.float32)
            out[i, c] = np.clip(0.8 * pat + 0.2 * noise, 0.0, 0.999)
    return torch.from_numpy(out)


def ragged_tokens(n, S, seed, vocab_size=28996, min_len=None):
    """input_ids / attention_mask / token_type_ids int64 [n,S] with the reference's collate layout
    (mmgclip/dataset/dataset.py:347: padding='max_length'): [CLS]=101 ... [SEP]=102, pad 0, lengths spread over 8..S."""
    r = _rng(seed, f"tokens{n}x{S}")
    lo = min(8, S) if min_len is None else int(min_len)           # min_len: long prompts only (the second C1 batch, round 3)
    lens = r.integers(lo, S + 1, n)
    lens[0], lens[-1] = S, lo                        # always cover the full-length and the shortest case
    ids = r.integers(1000, vocab_size, (n, S))
    mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.int64)
    ids = ids * mask
    ids[:, 0] = 101
    ids[np.arange(n), lens - 1] = 102
    return (torch.from_numpy(ids.astype(np.int64)), torch.from_numpy(mask), torch.zeros(n, S, dtype=torch.long))
