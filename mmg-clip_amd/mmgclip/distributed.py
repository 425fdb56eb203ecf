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

Knobs (all read when the object is built):
  MMG_GRAD_OVERLAP=0      no side stream, no buckets: ONE all-reduce per arena on the current stream after the backward
                          (the fallback for boxes where RCCL's kernels and the towers' full-grid kernels fight for CUs:
                          DESIGN.md "What round 2 measured" - two full-grid kernels do not share a CU)
  MMG_BUCKET_MB=<mb>      smallest bucket of the overlapped mode
  MMG_RCCL_MAX_CHANNELS=k caps RCCL's channels (= workgroups = CUs its kernels occupy next to the backward) by exporting
                          NCCL_MAX_NCHANNELS=k before the process group is created (init_from_env)
`GradSync.report()` / `Comm.report()` give the bytes moved and - with `timing=True` - the device time of the all-reduces and
the time the compute stream stood waiting for them at the end of the step (bench.py's `comm` block).
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
        self.counters = {"all_gather_bytes": 0, "all_gather_calls": 0, "all_reduce_bytes": 0, "all_reduce_calls": 0}

    @property
    def active(self):
        return self.world_size > 1 or self.always_exchange

    def all_gather_rows(self, t):
        """[n, ...] -> [world*n, ...] in rank order (no autograd: gradients are formed locally, see head.FusedClipLoss)."""
        t = t.contiguous()
        self.counters["all_gather_bytes"] += t.numel() * t.element_size() * self.world_size
        self.counters["all_gather_calls"] += 1
        out = torch.empty((self.world_size * t.shape[0],) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
        if dist.get_backend(self.group) == "nccl":
            dist.all_gather_into_tensor(out, t, group=self.group)        # one RCCL all-gather, no staging copies
        else:                                                            # gloo (CPU tests / single-GPU rehearsal)
            dist.all_gather(list(out.split(t.shape[0], dim=0)), t, group=self.group)
        return out

    def all_reduce_sum(self, t):
        self.counters["all_reduce_bytes"] += t.numel() * t.element_size()
        self.counters["all_reduce_calls"] += 1
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def report(self, reset=True):
        """Bytes / calls of every exchange since the last report (payload bytes as handed to the collective)."""
        out = dict(self.counters)
        if reset:
            for k in self.counters:
                self.counters[k] = 0
        return out


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
        cap = os.environ.get("MMG_RCCL_MAX_CHANNELS")
        if cap and backend == "nccl":       # one RCCL channel = one workgroup = one CU taken from the backward's kernels
            os.environ["NCCL_MAX_NCHANNELS"] = str(int(cap))
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
    """SUM all-reduce of parameter gradients: bucket by bucket on a side stream, overlapped with the remaining backward work
    (default), or one collective per arena after the backward (`overlap=False` / MMG_GRAD_OVERLAP=0).  Either way `finish()`
    guarantees that every element of every arena has been reduced exactly once when it returns."""

    def __init__(self, comm, arenas=(), extra_params=(), scale=1.0, bucket_bytes=None, overlap=None, timing=False):
        self.comm = comm
        self.scale = float(scale)           # 1 for the global-batch loss (sum); 1/world for per-rank local losses (mean)
        self.arenas = [a for a in arenas if a is not None]
        self.extra = [p for p in extra_params]
        self.overlap = (os.environ.get("MMG_GRAD_OVERLAP", "1") != "0") if overlap is None else bool(overlap)
        self.side = torch.cuda.Stream() if (torch.cuda.is_available() and self.overlap) else None
        self.bucket_bytes = int(bucket_bytes if bucket_bytes is not None else float(os.environ.get("MMG_BUCKET_MB", "32")) * 2 ** 20)
        self.timing = bool(timing) and torch.cuda.is_available()
        self._pending = []
        self._timed = []                    # (start, end) events of the collectives of the steps since the last report()
        self._waits = []                    # (before, after) events around the compute stream's join in finish()
        self._steps = 0
        self._bytes = 0
        self._calls = 0
        self._state = {}                    # id(arena) -> {"done": ranges already reduced this step, "ready": ranges waiting for a bucket}
        self.log = []                       # (id(arena), lo, hi) of every all-reduce of this step, in issue order (tests, diagnostics)
        for a in self.arenas:               # towers call arena.mark_ready(...) as their backward retires layers
            a.ready_hook = self.bucket_ready

    # ---- one collective --------------------------------------------------------------------------------------------
    def _reduce(self, buf):
        self.comm.all_reduce_sum(buf)
        if self.scale != 1.0:
            buf.mul_(self.scale)

    def _reduce_async(self, arena, lo, hi):
        buf = arena.grad[lo:hi]
        self.log.append((id(arena), lo, hi))
        self._bytes += 4 * (hi - lo)
        self._calls += 1
        if self.side is None:               # CPU (gloo) or overlap off: on the current stream, in program order
            ev = None
            if self.timing:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
            self._reduce(buf)
            if ev is not None:
                end = torch.cuda.Event(enable_timing=True)
                end.record()
                self._timed.append((ev, end))
            return
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            if self.timing:
                t0 = torch.cuda.Event(enable_timing=True)
                t0.record()
            self._reduce(buf)
            done = torch.cuda.Event(enable_timing=self.timing)
            done.record()
            if self.timing:
                self._timed.append((t0, done))
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
        if self.comm is None or not self.comm.active or not self.overlap:
            return
        st = self._st(arena)
        st["ready"].append((lo, hi))
        if 4 * sum(b - a for a, b in st["ready"]) >= self.bucket_bytes:
            self._flush(arena, st)

    def reduce_arena_async(self, arena):
        """Call when this arena's backward has been enqueued on the current stream: reduces everything not yet reduced
        (overlap off: nothing happens here, `finish()` reduces the arena in one piece)."""
        if self.comm is None or not self.comm.active or not self.overlap:
            return
        st = self._st(arena)
        st["ready"] = []
        for lo, hi in _complement(st["done"], arena.size):
            self._reduce_async(arena, lo, hi)
        st["done"] = [(0, arena.size)]

    def finish(self):
        """End of the step's backward: reduce whatever part of an arena no hook has reduced (overlap off: all of it; a tower
        whose post-backward hook never fired because one of its recorded forwards got no backward: the rest), the leftovers
        (projection heads, logit_scale) as ONE coalesced buffer, then join the side stream."""
        if self.comm is None or not self.comm.active:
            for a in self.arenas:
                a.open_backwards = 0
            return
        for a in self.arenas:
            if a.any_trainable():
                st = self._st(a)
                st["ready"] = []
                for lo, hi in _complement(st["done"], a.size):
                    self._reduce_async(a, lo, hi)
                st["done"] = [(0, a.size)]
            a.open_backwards = 0            # a forward whose backward never came must not disable the next step's hooks
        grads = [p.grad for p in self.extra if p.grad is not None]
        if grads:
            flat = torch.cat([g.reshape(-1) for g in grads])
            self._bytes += 4 * flat.numel()
            self._calls += 1
            self._reduce(flat)
            off = 0
            for g in grads:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()
        if self._pending:
            cur = torch.cuda.current_stream()
            if self.timing:
                b = torch.cuda.Event(enable_timing=True)
                b.record()
            for ev in self._pending:
                cur.wait_event(ev)
            if self.timing:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                self._waits.append((b, e))
            self._pending.clear()
        self._state.clear()
        self._steps += 1
        self.last_log, self.log = self.log, []

    def report(self, reset=True):
        """Per-step averages since the last report: gradient bytes / collectives, and with `timing` the device time of the
        all-reduces (`allreduce_busy_ms`) and how long the compute stream waited for the side stream at the end of the backward
        (`exposed_wait_ms`; in overlap-off mode the collectives run on the compute stream, so all of `allreduce_busy_ms` is exposed)."""
        n = max(self._steps, 1)
        out = {"mode": "bucketed, side stream, overlapped with backward" if self.overlap else "one all-reduce per arena after backward",
               "bucket_mb": round(self.bucket_bytes / 2 ** 20, 1) if self.overlap else None,
               "grad_allreduce_bytes_per_step": self._bytes // n, "grad_allreduce_calls_per_step": round(self._calls / n, 1)}
        if self.timing:
            torch.cuda.synchronize()
            busy = sum(a.elapsed_time(b) for a, b in self._timed)
            wait = sum(a.elapsed_time(b) for a, b in self._waits)
            out["allreduce_busy_ms_per_step"] = round(busy / n, 3)
            out["exposed_wait_ms_per_step"] = round((wait if self.overlap else busy) / n, 3)
        if reset:
            self._timed, self._waits, self._steps, self._bytes, self._calls = [], [], 0, 0, 0
        return out
