"""Flat parameter / gradient arenas for the encoder towers.

The towers keep ordinary fp32 `nn.Parameter`s (state-dict compatible with torchvision / Hugging Face key names), but on
the GPU every parameter is a VIEW into one flat fp32 buffer and every `.grad` a view into a second one.  That gives
  * one memset for zero_grad, one fused AdamW launch (csrc/norm_elementwise.hip) for the whole tower,
  * ONE large RCCL all-reduce per tower instead of hundreds of small ones (xGMI rings are per-link bound: few, large
    collectives), and
  * bf16 working copies (cast / transposed / fused layouts the MFMA kernels read) refreshed only when a parameter changed.
"""
import torch


class ParamArena:
    def __init__(self, named_params, device):
        """named_params: ordered list of (name, nn.Parameter).  Rebinds p.data into the flat buffer (keeps identity)."""
        self.names = [n for n, _ in named_params]
        self.params = [p for _, p in named_params]
        self.offsets = {}
        off = 0
        for n, p in named_params:
            self.offsets[n] = off
            off += (p.numel() + 63) // 64 * 64          # 256-byte aligned slices
        self.size = off
        self.device = device
        self.data = torch.zeros(off, device=device, dtype=torch.float32)
        self.grad = torch.zeros(off, device=device, dtype=torch.float32)
        self._grad_views = {}
        for n, p in named_params:
            o = self.offsets[n]
            view = self.data[o:o + p.numel()].view(p.shape)
            view.copy_(p.data.to(device=device, dtype=torch.float32))
            p.data = view
            p._mmg_arena = self                 # lets an optimizer find the flat buffers from the parameter alone (optim.FusedAdamW)
            self._grad_views[n] = self.grad[o:o + p.numel()].view(p.shape)
        self._by_name = dict(named_params)
        self.manual_version = 0
        self.ready_hook = None              # ready_hook(arena, lo, hi): gradients of elements [lo, hi) are final (distributed.GradSync)
        self.open_backwards = 0             # recorded forwards of the owning tower whose backward has not arrived yet (note_forward)

    # ---- views -------------------------------------------------------------------------------------------
    def p(self, name):
        return self._by_name[name].data

    def g(self, name):
        return self._grad_views[name]

    def span(self, first, last):
        """fp32 data slice covering parameters first..last (inclusive, contiguous in arena order)."""
        a = self.offsets[first]
        b = self.offsets[last] + self._by_name[last].numel()
        return self.data[a:b]

    def gspan(self, first, last):
        a = self.offsets[first]
        b = self.offsets[last] + self._by_name[last].numel()
        return self.grad[a:b]

    def is_bound(self):
        p0 = self.params[0]
        return p0.data.data_ptr() == self.data.data_ptr() + 4 * self.offsets[self.names[0]]

    # ---- versions (to refresh bf16 working copies lazily) ----------------------------------------------------
    def version(self):
        return (self.manual_version, sum(p._version for p in self.params))

    def touch(self):
        self.manual_version += 1

    # ---- gradients -----------------------------------------------------------------------------------------
    def prepare_grads(self):
        """Called at the start of a tower backward: make every trainable p.grad the arena view.

        Kernels ACCUMULATE into the arena, so: p.grad is None (zero_grad(set_to_none=True)) -> zero the slice and
        attach; p.grad already the view -> keep (accumulate, as autograd would); foreign p.grad -> fold it in.
        """
        all_none = all(p.grad is None for p in self.params if p.requires_grad)
        if all_none:
            self.grad.zero_()
        for n, p in zip(self.names, self.params):
            if not p.requires_grad:
                continue
            v = self._grad_views[n]
            if p.grad is None:
                if not all_none:
                    v.zero_()
                p.grad = v
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
                p.grad = v

    def any_trainable(self):
        return any(p.requires_grad for p in self.params)

    # ---- "this part of the gradient is final" (bucketed all-reduce overlapped with the rest of the tower's backward) -------
    def range_of(self, prefixes):
        """[lo, hi) element range covering every parameter whose name starts with one of `prefixes` (they must be contiguous in
        arena order, which holds for whole layers / stages); None when nothing matches."""
        if isinstance(prefixes, str):
            prefixes = (prefixes,)
        idx = [i for i, n in enumerate(self.names) if n.startswith(tuple(prefixes))]
        if not idx:
            return None
        assert idx == list(range(idx[0], idx[-1] + 1)), f"parameters under {prefixes} are not contiguous in the arena"
        lo = self.offsets[self.names[idx[0]]]
        hi = self.offsets[self.names[idx[-1] + 1]] if idx[-1] + 1 < len(self.names) else self.size
        return lo, hi

    def mark_ready(self, prefixes):
        """Tower backward: every gradient under `prefixes` has received its last contribution of this step (the work that
        produced it is enqueued on the current stream)."""
        if self.ready_hook is None:
            return
        r = self.range_of(prefixes)
        if r is not None:
            self.ready_hook(self, r[0], r[1])


# ---- when is a tower's gradient complete? ------------------------------------------------------------------------------------
# A tower can run several times in one step (MMGCLIPLoss: the text tower encodes the report and the impression,
# mmgclip/networks/mmgclip_model.py:154-164) and every run has its own autograd backward accumulating into the same arena.
# `post_backward_hook(arena)` (the gradient all-reduce of distributed.GradSync) must fire after the LAST of them only.
def note_forward(tower, needs_grad):
    """Call from the tower's forward: one more backward will arrive if this forward was recorded for autograd.  The count lives
    on the tower's arena, so that whoever owns the arena's all-reduce (distributed.GradSync.finish) can reset it at the step's end."""
    if needs_grad:
        tower._arena.open_backwards += 1


def backward_finished(tower):
    """Call at the end of the tower's autograd backward; fires the hook once no recorded forward is left without its backward."""
    a = tower._arena
    a.open_backwards = max(0, a.open_backwards - 1)
    if a.open_backwards == 0 and tower.post_backward_hook is not None:
        tower.post_backward_hook(a)


def last_backward(tower):
    """True inside the LAST open backward of the tower for this step: only then are finished layers' gradients final."""
    return tower._arena.open_backwards <= 1


def begin_step(tower):
    """Forget forwards whose backward never came (e.g. an evaluation pass run with gradients enabled)."""
    if getattr(tower, "_arena", None) is not None:
        tower._arena.open_backwards = 0


def stream_anchor(tower, device):
    """The zero-sized-work leaf a tower's autograd Function takes so that autograd has a gradient to accumulate on the tower's stream.
    An AccumulateGrad node remembers the stream it was CREATED on and lives as long as any graph that reached it (a kept `loss`): a tower
    that moves to another stream (MMGCLIP's side stream switched off for bench.py's one-stream roofline steps, `MMG_TEXT_STREAM`
    flipped between calls) would hand that node a gradient from a different stream - torch warns ("AccumulateGrad node's stream does
    not match ...") and may insert a synchronisation.  So the anchor is per stream: a new leaf whenever the tower runs on a stream it
    has not been seen on (the node that tripped the warning was the text tower's anchor)."""
    key = torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else None
    anchors = tower.__dict__.setdefault("_anchors", {})
    a = anchors.get(key)
    if a is None or a.device != device:
        a = anchors[key] = torch.zeros(1, device=device, requires_grad=True)
    return a
