"""seeding / directory helpers with the semantics of mmgclip/utils/global_utils.py:7-41."""
import os
import random

import numpy as np
import torch


def seeding(seed):
    """random, PYTHONHASHSEED, numpy, torch (+ device generator); deterministic flags as the reference sets them."""
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
