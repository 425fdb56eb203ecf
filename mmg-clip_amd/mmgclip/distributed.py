"""Data-parallel plumbing: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI; "gloo" in the
CPU tests).  The reference has no multi-GPU path; this is the §8(e) design:

  exchange 1  all-gather of the L2-normalised image / text embeddings ([n_local, D] each)
  exchange 2  all-gather of the two log-sum-exp vectors (so embedding gradients are complete locally; no N x D
              reduce-scatter, no autograd through a collective)
  exchange 3  SUM all-reduce of the parameter gradients on a side stream, overlapped with the backward: a tower reports
              finished layers / stages (`ParamArena.mark_ready`), contiguous ready ranges of its flat gradient buffer are
              reduced in buckets of >= MMG_BUCKET_MB (default 32 MB: xGMI rings are per-link bound - few, large collectives)
              while the earlier layers are still in backward; whatever is left goes when the tower's backward ends, plus one
              coalesced buffer for the small leftovers (heads, logit_scale).
The fused loss already carries the 1/(2N) factor of the GLOBAL mean, so gradients are summed, not averaged.
"""
import os

import torch
import torch.distributed as dist


class Comm:
    def __init__(self, group=None, always_exchange=False):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)
        # a one-rank group normally skips every exchange; `always_exchange` keeps them (how the RCCL calls, their
        # stream ordering and buffer contracts are exercised on a one-GPU box: tests/test_distributed_gpu.py)
        self.always_exchange = always_exchange

    @property
    def active(self):
        return self.world_size > 1 or self.always_exchange

    def all_gather_rows(self, t):
        """[n, ...] -> [world*n, ...] in rank order (no autograd: gradients are formed locally, see head.FusedClipLoss)."""
        t = t.contiguous()
        out = torch.empty((self.world_size * t.shape[0],) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
        if dist.get_backend(self.group) == "nccl":
            dist.all_gather_into_tensor(out, t, group=self.group)        # one RCCL all-gather, no staging copies
        else:                                                            # gloo (CPU tests / single-GPU rehearsal)
            dist.all_gather(list(out.split(t.shape[0], dim=0)), t, group=self.group)
        return out

    def all_reduce_sum(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t


def init_from_env(backend=None, single_rank=False):
    """Initialise the default process group from torchrun's environment; returns Comm or None when WORLD_SIZE <= 1
    (`single_rank=True`: build the one-rank group anyway and keep its exchanges — rehearsal of the RCCL path)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and not single_rank:
        return None
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if torch.cuda.is_available():
            idx = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
            torch.cuda.set_device(idx)
            if backend == "nccl":           # bind the communicator to this rank's GPU up front (no lazy device guess)
                kw["device_id"] = torch.device("cuda", idx)
        dist.init_process_group(backend=backend, **kw)
    return Comm(always_exchange=single_rank and world <= 1)


def _merge(ranges):
    out = []
    for lo, hi in sorted(ranges):
        if out and lo <= out[-1][1]:
            out[-1] = (out[-1][0], max(out[-1][1], hi))
        else:
            out.append((lo, hi))
    return out


def _complement(done, size):
    out, at = [], 0
    for lo, hi in _merge(done):
        if lo > at:
            out.append((at, lo))
        at = max(at, hi)
    if at < size:
        out.append((at, size))
    return out


class GradSync:
    """SUM all-reduce of parameter gradients, bucket by bucket, overlapped with the remaining backward work."""

    def __init__(self, comm, arenas=(), extra_params=(), scale=1.0, bucket_bytes=None):
        self.comm = comm
        self.scale = float(scale)           # 1 for the global-batch loss (sum); 1/world for per-rank local losses (mean)
        self.arenas = [a for a in arenas if a is not None]
        self.extra = [p for p in extra_params]
        self.side = torch.cuda.Stream() if torch.cuda.is_available() else None
        self.bucket_bytes = int(bucket_bytes if bucket_bytes is not None else float(os.environ.get("MMG_BUCKET_MB", "32")) * 2 ** 20)
        self._pending = []
        self._state = {}                    # id(arena) -> {"done": ranges already reduced this step, "ready": ranges waiting for a bucket}
        self.log = []                       # (id(arena), lo, hi) of every all-reduce of this step, in issue order (tests, diagnostics)
        for a in self.arenas:               # towers call arena.mark_ready(...) as their backward retires layers
            a.ready_hook = self.bucket_ready

    # ---- one collective --------------------------------------------------------------------------------------------
    def _reduce_async(self, arena, lo, hi):
        buf = arena.grad[lo:hi]
        self.log.append((id(arena), lo, hi))
        if self.side is None:
            self.comm.all_reduce_sum(buf)
            if self.scale != 1.0:
                buf.mul_(self.scale)
            return
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            self.comm.all_reduce_sum(buf)
            if self.scale != 1.0:
                buf.mul_(self.scale)
            done = torch.cuda.Event()
            done.record()
        self._pending.append(done)

    def _st(self, arena):
        return self._state.setdefault(id(arena), {"done": [], "ready": []})

    def _flush(self, arena, st):
        for lo, hi in _merge(st["ready"]):
            self._reduce_async(arena, lo, hi)
            st["done"].append((lo, hi))
        st["ready"] = []

    # ---- called by the towers -----------------------------------------------------------------------------------------
    def bucket_ready(self, arena, lo, hi):
        """Gradients [lo, hi) of this arena are final and their producers are enqueued on the current stream."""
        if self.comm is None or not self.comm.active:
            return
        st = self._st(arena)
        st["ready"].append((lo, hi))
        if 4 * sum(b - a for a, b in st["ready"]) >= self.bucket_bytes:
            self._flush(arena, st)

    def reduce_arena_async(self, arena):
        """Call when this arena's backward has been enqueued on the current stream: reduces everything not yet reduced."""
        if self.comm is None or not self.comm.active:
            return
        st = self._st(arena)
        st["ready"] = []
        for lo, hi in _complement(st["done"], arena.size):
            self._reduce_async(arena, lo, hi)
        st["done"] = [(0, arena.size)]

    def finish(self):
        """Reduce the leftovers (projection heads, logit_scale) as ONE coalesced buffer and join the side stream."""
        if self.comm is None or not self.comm.active:
            return
        grads = [p.grad for p in self.extra if p.grad is not None]
        if grads:
            flat = torch.cat([g.reshape(-1) for g in grads])
            self.comm.all_reduce_sum(flat)
            if self.scale != 1.0:
                flat.mul_(self.scale)
            off = 0
            for g in grads:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()
        for ev in self._pending:
            torch.cuda.current_stream().wait_event(ev)
        self._pending.clear()
        self._state.clear()
        self.last_log, self.log = self.log, []
