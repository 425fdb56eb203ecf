import math
from typing import Union

from torch.optim import Optimizer
from torch.optim.lr_scheduler import LambdaLR


class LinearWarmupCosineAnnealingLR(LambdaLR):
    """Linear warm-up from 0 over `warmup_steps` (an int, or a float fraction of `total_steps`, ceil-ed), then
    lr * cos^2((t - w) / (T - w) * pi/2).  Same constructor and multiplier as mmgclip/scheduler/warmup_cosine.py:41-61
    (so, stepped once per epoch as the reference does, epoch 1 trains at lr = 0: SURVEY.md §0)."""

    def __init__(self, optimizer: Optimizer, total_steps: int, warmup_steps: Union[int, float], last_epoch: int = -1, **kwargs):
        assert warmup_steps < total_steps, "Warmup steps should be less than total steps."
        self.tsteps = total_steps
        self.wsteps = math.ceil(total_steps * warmup_steps) if isinstance(warmup_steps, float) else warmup_steps
        super().__init__(optimizer, self._lr_multiplier, last_epoch)

    def _lr_multiplier(self, step: int) -> float:
        if step < self.wsteps:
            value = step / float(max(1, self.wsteps))
        else:
            value = math.cos((step - self.wsteps) / (self.tsteps - self.wsteps) * (math.pi / 2)) ** 2
        return max(0, value)
