"""Fused AdamW over the towers' flat parameter arenas (csrc/norm_elementwise.hip: mmg_adamw_step) with
torch.optim.AdamW semantics (mmgclip/experiments/ClassifierExperiment.py:74: lr 5e-5, weight_decay 1e-4, default betas).

Arena-backed parameters (ConvNeXt / ViT / ResNet / BERT towers) are updated by ONE launch per tower; any other parameter
(projection heads, logit_scale) gets one launch per tensor.  The arenas are found from the parameters themselves
(`ParamArena` leaves a back-pointer on each parameter it binds), at every `step()`: the towers build their arenas at their
first forward, which is AFTER the reference's construction order builds the optimizer (ClassifierExperiment.py:65-74).
The one-launch path is taken when every parameter of an arena is trainable, in this optimizer, and has its gradient in the
arena's flat gradient buffer; otherwise its trainable parameters are updated one by one and the arena is told so
(`touch()`), which is what makes the towers rebuild their bf16 working copies.

State layout = torch.optim.AdamW's: `state[p] = {'step': fp32 0-d tensor, 'exp_avg', 'exp_avg_sq'}`, so
`state_dict()` / `load_state_dict()` (what EarlyStopper writes as 'optimizer_state_dict', callbacks/early_stopping.py:52-65)
round-trip and are interchangeable with the reference's torch.optim.AdamW PROVIDED both list the same parameters in the same order:
the reference passes `model.parameters()` - frozen BERT included - and so does ClassifierExperiment here (parameters without a
gradient are skipped and get no state, as in torch).  For arena parameters `exp_avg` / `exp_avg_sq` are
views into two flat buffers (rebuilt from the per-parameter tensors after a load)."""
import torch

from . import kernels as K


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, arenas=()):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        # `arenas` is accepted for backward compatibility; arenas are discovered from the parameters at step time
        self._flat = {}              # id(arena) -> dict(arena, m, v, step)

    # ---- state plumbing --------------------------------------------------------------------------------------------
    def _flat_state(self, arena):
        """Flat moment buffers of an arena, with every parameter's `exp_avg` / `exp_avg_sq` a view into them."""
        fs = self._flat.get(id(arena))
        p0 = arena.params[0]
        bound = fs is not None and fs["arena"] is arena and "exp_avg" in self.state[p0] and \
            self.state[p0]["exp_avg"].data_ptr() == fs["m"].data_ptr() + 4 * arena.offsets[arena.names[0]]
        if bound:
            return fs
        m, v = torch.zeros_like(arena.data), torch.zeros_like(arena.data)
        step = torch.zeros((), dtype=torch.float32)
        for n, p in zip(arena.names, arena.params):
            o = arena.offsets[n]
            mv, vv = m[o:o + p.numel()].view(p.shape), v[o:o + p.numel()].view(p.shape)
            st = self.state[p]
            if "exp_avg" in st:                                  # loaded from a checkpoint (or updated per tensor so far)
                mv.copy_(st["exp_avg"])
                vv.copy_(st["exp_avg_sq"])
                step = torch.maximum(step, torch.as_tensor(st["step"], dtype=torch.float32).cpu().reshape(()))
            st["exp_avg"], st["exp_avg_sq"] = mv, vv
        for p in arena.params:
            self.state[p]["step"] = step                         # one shared counter: the arena is stepped as a whole
        fs = self._flat[id(arena)] = dict(arena=arena, m=m, v=v, step=step)
        return fs

    def state_dict(self):
        """torch.optim.AdamW's layout.  The parameters of an arena share ONE live step counter; a serialised state gets a counter
        of its own per parameter, as torch writes it (torch's foreach step increments every counter it is handed: a tensor listed
        178 times would be incremented 178 times per step)."""
        sd = super().state_dict()
        sd["state"] = {k: {kk: (vv.clone() if kk == "step" and torch.is_tensor(vv) else vv) for kk, vv in v.items()}
                       for k, v in sd["state"].items()}
        return sd

    def _tensor_state(self, p):
        st = self.state[p]
        if "exp_avg" not in st:
            st["step"] = torch.zeros((), dtype=torch.float32)
            st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(p.data), torch.zeros_like(p.data)
        elif not st["exp_avg"].is_contiguous() or st["exp_avg"].device != p.device:
            st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"].to(p.device).contiguous(), st["exp_avg_sq"].to(p.device).contiguous()
        if not torch.is_tensor(st["step"]):
            st["step"] = torch.tensor(float(st["step"]), dtype=torch.float32)
        return st

    # ---- the step --------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            lr, (b1, b2), eps, wd = group["lr"], group["betas"], group["eps"], group["weight_decay"]
            in_group = {id(p) for p in group["params"]}
            arenas, done = {}, set()
            for p in group["params"]:
                a = getattr(p, "_mmg_arena", None)
                if a is not None and a.is_bound():
                    arenas[id(a)] = a
            for arena in arenas.values():
                whole = all(p.requires_grad and id(p) in in_group and p.grad is not None and
                            p.grad.data_ptr() == arena.g(n).data_ptr() for n, p in zip(arena.names, arena.params))
                if not whole:
                    continue
                fs = self._flat_state(arena)
                fs["step"] += 1
                K.adamw_step(arena.data, arena.grad, fs["m"], fs["v"], None, lr, b1, b2, eps, wd, int(fs["step"]))
                arena.touch()
                done.update(id(p) for p in arena.params)
            touched = {}
            for p in group["params"]:
                if p.grad is None or id(p) in done:
                    continue
                if not p.data.is_contiguous():
                    raise RuntimeError("FusedAdamW needs contiguous parameters")
                st = self._tensor_state(p)
                if st["step"].data_ptr() in {fs["step"].data_ptr() for fs in self._flat.values()}:
                    st["step"] = st["step"].clone()              # leaving the whole-arena path: own counter from here on
                st["step"] += 1
                K.adamw_step(p.data, p.grad.contiguous(), st["exp_avg"], st["exp_avg_sq"], None, lr, b1, b2, eps, wd, int(st["step"]))
                a = getattr(p, "_mmg_arena", None)
                if a is not None:
                    touched[id(a)] = a
            for a in touched.values():       # the update went through a raw pointer: p._version did not move, say so here
                a.touch()
        return loss
