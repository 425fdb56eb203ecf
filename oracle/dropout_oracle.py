"""TEST INFRASTRUCTURE - numpy restatement of the counter-based dropout mask of mmg-clip_amd/csrc/dropout.h.

Only tests/ may import this.  The reference's dropout is torch.nn.Dropout inside Hugging Face BertModel, live because
ClassifierExperiment.train() calls model.train() (/root/reference/mmgclip/experiments/ClassifierExperiment.py:97,
/root/reference/mmgclip/networks/encoder.py:156; p = 0.1 per notebooks/bert_experimental.ipynb:609-624).  torch's random stream
is not reproducible outside torch, so parity is defined GIVEN the mask: the HIP kernels and this file derive the same mask from
(seed, site, element index); the BERT oracle (oracle/encoders_oracle.py: bert_forward(..., dropout=...)) applies it at HF's
four dropout positions.  Parity of the mask generator itself is unpinned by construction (there is no reference mask to pin to);
what is pinned is where the masks are applied (HF module order) - see tests/test_oracle_encoders.py.
"""
import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def _u32(x):
    return np.asarray(x, dtype=np.uint64) & _M32


def fmix32(x):
    x = _u32(x)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & _M32
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & _M32
    x ^= x >> np.uint64(16)
    return x


def drop_key(seed, site):
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    lo, hi = seed & 0xFFFFFFFF, seed >> 32
    return int(fmix32(np.uint64(((int(site) ^ hi) + lo) & 0xFFFFFFFF)))


def drop_threshold(p):
    t = float(np.float32(p)) * 4294967296.0
    return 0 if t <= 0 else (4294967295 if t >= 4294967295.0 else int(t))


def keep_mask(index, p, seed, site, key=None):
    """Boolean array: element `index` (any integer array, taken modulo 2^32) survives dropout (`key`: an array of per-element
    keys replacing drop_key(seed, site))."""
    key = np.uint64(drop_key(seed, site)) if key is None else _u32(key)
    bits = fmix32((_u32(index) * np.uint64(0x9E3779B1) + key) & _M32)
    return bits >= np.uint64(drop_threshold(p))


def hidden_mask(n_tokens, C, p, seed, site):
    """[n_tokens, C] keep mask of a hidden-state dropout site (index = token * C + column, token = b * S + s)."""
    idx = np.arange(n_tokens, dtype=np.uint64)[:, None] * np.uint64(C) + np.arange(C, dtype=np.uint64)[None, :]
    return keep_mask(idx, p, seed, site)


def attention_mask(B, heads, S, p, seed, site, first_sequence=0):
    """[B, heads, S(query), S(key)] keep mask of the attention-probability dropout of sequences first_sequence .. first_sequence + B:
    index = q * 512 + k under the per-(sequence, head) key fmix32(key + bh * 0xB5297A4D), bh = sequence * heads + head."""
    bh = np.arange(B * heads, dtype=np.uint64)[:, None, None] + np.uint64(first_sequence * heads)
    key_bh = fmix32((np.uint64(drop_key(seed, site)) + bh * np.uint64(0xB5297A4D)) & _M32)
    q = np.arange(S, dtype=np.uint64)[None, :, None]
    k = np.arange(S, dtype=np.uint64)[None, None, :]
    return keep_mask(q * np.uint64(512) + k + np.uint64(0) * bh, p, seed, site, key=key_bh).reshape(B, heads, S, S)


# dropout sites of the text tower (mmg-clip_amd/mmgclip/networks/bert.py)
SITE_EMBEDDINGS = 0


def site_attention_probs(layer):
    return 4 * layer + 1


def site_attention_output(layer):
    return 4 * layer + 2


def site_ffn_output(layer):
    return 4 * layer + 3
