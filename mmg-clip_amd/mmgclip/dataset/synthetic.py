"""Synthetic stand-in for the reference's datasets (mmgclip/dataset/dataset.py needs proprietary data): emits batches
with the collate_fn's keys and dtypes (dataset.py:343-351 / :548-561) from seeded generators.

    image_features  fp32 [n,1,768,1,1]  |N(1, 0.4)|   (pre-extracted mode)        or
    image           fp32 [n,Cin,H,W]    U[0,1)        (pixel encoders)
    text_tokens     {input_ids, token_type_ids, attention_mask} int64 [n,S]: [CLS]=101 ... [SEP]=102, body U{1000..V-1},
                    pad 0, lengths U{8..S}
    image_label     int64 [n,1], image_id list[str], image_description list[str], prompt_labels list[dict]
"""
import torch


class TokenBatch(dict):
    """dict with `.to(device)` acting in place like transformers.BatchEncoding (mmgclip_model.py:106)."""

    def to(self, device):
        for k in list(self.keys()):
            self[k] = self[k].to(device)
        return self


def synthetic_tokens(n, S, vocab_size=28996, generator=None, fixed_length=None):
    g = generator
    lens = torch.randint(min(8, S), S + 1, (n,), generator=g) if fixed_length is None else torch.full((n,), fixed_length)
    ids = torch.randint(1000, vocab_size, (n, S), generator=g)
    pos = torch.arange(S)[None, :]
    mask = (pos < lens[:, None]).long()
    ids = ids * mask
    ids[:, 0] = 101
    ids[torch.arange(n), lens - 1] = 102
    return TokenBatch(input_ids=ids, token_type_ids=torch.zeros(n, S, dtype=torch.long), attention_mask=mask)


def synthetic_batch(n, S=77, image_size=None, in_chans=1, feature_dim=768, vocab_size=28996, seed=42, with_impression=False):
    g = torch.Generator().manual_seed(seed)
    batch = {}
    if image_size is None:
        batch["image_features"] = (1.0 + 0.4 * torch.randn(n, 1, feature_dim, 1, 1, generator=g)).abs()
    else:
        batch["image"] = torch.rand(n, in_chans, image_size, image_size, generator=g)
    batch["text_tokens"] = synthetic_tokens(n, S, vocab_size, g)
    if with_impression:
        batch["image_impression_tokens"] = synthetic_tokens(n, S, vocab_size, g)
    batch["image_label"] = torch.randint(0, 2, (n, 1), generator=g)
    batch["image_id"] = [f"synthetic_{seed}_{i}" for i in range(n)]
    batch["image_description"] = ["synthetic"] * n
    shapes = torch.randint(0, 4, (n,), generator=g)
    birads = torch.randint(-1, 7, (n,), generator=g)
    batch["prompt_labels"] = [{"BenignMalignantDatasetLabels": int(batch["image_label"][i, 0]), "MassShapeLabels": int(shapes[i]),
                               "BIRADS": "unknown" if int(birads[i]) < 0 else str(int(birads[i]))} for i in range(n)]
    return batch


# label enums of the reference (mmgclip/prompts/enums.py:17-19,29-34) that the validation prompts are built from
BENIGN_MALIGNANT = {"benign": 0, "malignant": 1}
MASS_SHAPES = {"unknown": 0, "oval": 1, "round": 2, "irregular": 3}


def validation_prompts(metrics):
    """The prompt strings ClassifierExperiment.validate builds (ClassifierExperiment.py:147-163), keyed by metric."""
    out = {}
    if "BenignMalignantDatasetLabels" in metrics:
        out["malig"] = ["Finding suggesting malignant."]
    if "MassShapeLabels" in metrics:
        out["shapes"] = [f"Mass shape is {name}." for name in MASS_SHAPES]
    if "birads" in metrics:
        out["birads"] = ["BIRADS unknown."] + [f"BIRADS score of {i}." for i in range(0, 7)]
    return out


def synthetic_prompt_tokens(strings, S, vocab_size=28996):
    """Deterministic stand-in for the absent tokenizer vocabulary: ids hashed from the words of each prompt."""
    import zlib
    n = len(strings)
    ids = torch.zeros(n, S, dtype=torch.long)
    mask = torch.zeros(n, S, dtype=torch.long)
    for i, text in enumerate(strings):
        words = text.replace(".", " .").split()
        toks = [101] + [1000 + zlib.crc32(w.lower().encode()) % (vocab_size - 1000) for w in words][:S - 2] + [102]
        ids[i, :len(toks)] = torch.tensor(toks)
        mask[i, :len(toks)] = 1
    return TokenBatch(input_ids=ids, token_type_ids=torch.zeros(n, S, dtype=torch.long), attention_mask=mask)


class SyntheticLoader:
    """Iterable of `steps` synthetic batches (a stand-in for DataLoaders(...).get_dataloader, dataloaders.py:17-40)."""

    def __init__(self, steps, batch_size, seed=42, **kw):
        self.steps, self.batch_size, self.seed, self.kw = steps, batch_size, seed, kw

    def __len__(self):
        return self.steps

    def __iter__(self):
        for i in range(self.steps):
            yield synthetic_batch(self.batch_size, seed=self.seed + i, **self.kw)
