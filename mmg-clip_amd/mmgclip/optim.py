"""Fused AdamW over the towers' flat parameter arenas (csrc/norm_elementwise.hip: mmg_adamw_step) with
torch.optim.AdamW semantics (mmgclip/experiments/ClassifierExperiment.py:74: lr 5e-5, weight_decay 1e-4, default betas).

Arena-backed parameters (ConvNeXt / BERT towers) are updated by ONE launch per tower; any other parameter (projection
heads, logit_scale) gets one launch per tensor.  Frozen slices of an arena are skipped by giving them zero gradient and
no weight decay: the arena kernel is only used when every parameter of the arena is trainable."""
import torch

from . import kernels as K


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, arenas=()):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.arenas = [a for a in arenas if a is not None]
        self._arena_state = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        covered = set()
        group = self.param_groups[0]
        lr, (b1, b2), eps, wd = group["lr"], group["betas"], group["eps"], group["weight_decay"]
        for arena in self.arenas:
            if not all(p.requires_grad and p.grad is not None and p.grad.data_ptr() == arena.g(n).data_ptr()
                       for n, p in zip(arena.names, arena.params)):
                continue
            st = self._arena_state.setdefault(id(arena), dict(step=0, m=torch.zeros_like(arena.data), v=torch.zeros_like(arena.data)))
            st["step"] += 1
            K.adamw_step(arena.data, arena.grad, st["m"], st["v"], None, lr, b1, b2, eps, wd, st["step"])
            arena.touch()
            covered.update(id(p) for p in arena.params)
        for group in self.param_groups:
            lr, (b1, b2), eps, wd = group["lr"], group["betas"], group["eps"], group["weight_decay"]
            for p in group["params"]:
                if p.grad is None or id(p) in covered:
                    continue
                st = self.state[p]
                if not st:
                    st["step"], st["m"], st["v"] = 0, torch.zeros_like(p.data), torch.zeros_like(p.data)
                st["step"] += 1
                g = p.grad.contiguous()
                if p.data.is_contiguous():
                    K.adamw_step(p.data, g, st["m"], st["v"], None, lr, b1, b2, eps, wd, st["step"])
                else:
                    raise RuntimeError("FusedAdamW needs contiguous parameters")
        return loss
