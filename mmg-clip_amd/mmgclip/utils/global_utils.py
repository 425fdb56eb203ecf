"""seeding / directory helpers with the semantics of mmgclip/utils/global_utils.py:7-41."""
import os
import random

import numpy as np
import torch


_SEED_EPOCH = [0, None]          # (how often seeding() ran, its last seed): private random streams re-derive themselves from it


def seed_epoch():
    """(epoch, seed) of the last `seeding()` call - (0, None) before the first.  Components that own a private random stream (the text
    tower's dropout-seed generator) restart it when the epoch moves, so `seeding(s)` reproduces a run without those components ever
    consuming torch's global CPU generator (whose draws the reference spends on the DataLoader's sampler only)."""
    return tuple(_SEED_EPOCH)


def seeding(seed):
    """random, PYTHONHASHSEED, numpy, torch (+ device generator); deterministic flags as the reference sets them."""
    _SEED_EPOCH[0] += 1
    _SEED_EPOCH[1] = int(seed)
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = True


def create_directory_if_not_exists(path):
    os.makedirs(path, exist_ok=True)
    return path
