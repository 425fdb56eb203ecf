"""HIP-event timing of the hot kernels, per kernel instantiation and shape class (bench.py's roofline leg).

An event pair brackets ONE launch on the stream it is launched on.  The pair measures that kernel alone only while nothing else
runs on the device, so the profiled steps are run with every tower on ONE stream (bench.py switches the text tower's side stream
off for them, `MMGCLIP.text_stream_enabled = False`) and OUTSIDE the timed region: `value` carries no instrumentation.  What an
empty pair reads (`event_overhead_us`, calibrated at `enable()`) is subtracted from every duration.

Records are keyed by the kernel INSTANTIATION the library's dispatcher chose (`mmg_last_kernel()`, spelled as rocprofv3's kernel
trace spells it) and, below that, by a shape-class label (e.g. "M=1048576 N=1536 K=384 epi=gelu+aux"), so that a line of the bench
output can be set beside the rocprofv3 table name for name.  `flops` / `bytes` are the ALGORITHMIC figures of the launch
(DESIGN.md §4): 2*M*N*K per GEMM, operands read once + outputs written once.  Off by default: a disabled profiler costs one
attribute test per launch."""
import torch

MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense bf16, MI355X_MICROARCH.md
MFMA_FP8_PEAK_TFLOPS = 5000.0       # dense e4m3 on the K = 128 MFMA, same guide
HBM_PEAK_GBS = 8000.0


class KernelProfile:
    def __init__(self):
        self.on = False
        self.records = []            # (family, kernel, label, start, end, flops, bytes)
        self.event_overhead_us = 0.0
        self._lib = None

    def enable(self):
        from . import _hip
        self._lib = _hip.load()
        self._lib.mmg_set_kernel_notes(1)
        self.records = []
        # what an event pair with nothing between reads on an otherwise busy stream: subtracted from every launch
        pairs = []
        for _ in range(64):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            e.record()
            pairs.append((s, e))
        torch.cuda.synchronize()
        t = sorted(s.elapsed_time(e) for s, e in pairs)
        self.event_overhead_us = 1e3 * t[len(t) // 2]
        self.on = True

    def disable(self):
        self.on = False
        if self._lib is not None:
            self._lib.mmg_set_kernel_notes(0)

    def timed(self, family, flops, nbytes, fn, label=""):
        if not self.on:
            return fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self._lib.mmg_set_kernel_notes(1)            # (clears the last note: a stale name can never label this launch)
        s.record()
        r = fn()
        e.record()
        name = self._lib.mmg_last_kernel().decode()
        stem = family.replace("_fp8", "").replace("_kernels", "").replace("_kernel", "")
        if not name.startswith(stem):
            name = family                # (entry points that launch several kernels, or do not note theirs, keep the family name)
        self.records.append((family, name, label, s, e, float(flops), float(nbytes)))
        return r

    # ---- aggregation -------------------------------------------------------------------------------------------------------
    @staticmethod
    def _stats(recs, ovh_ms):
        t_ms = sum(max(r[0] - ovh_ms, 0.0) for r in recs)
        fl, by, n = sum(r[1] for r in recs), sum(r[2] for r in recs), len(recs)
        t_ms = max(t_ms, 1e-9)
        return {"launches": n, "total_ms": round(t_ms, 3), "avg_launch_us": round(t_ms * 1e3 / n, 1),
                "tflops": round(fl / (t_ms * 1e-3) / 1e12, 1), "gbytes_per_s": round(by / (t_ms * 1e-3) / 1e9, 1),
                "algorithmic_flop_per_launch": round(fl / n), "algorithmic_bytes_per_launch": round(by / n)}

    def kernels(self):
        """{kernel instantiation: stats + {"family", "shape_classes": {label: stats}}} sorted by total time, largest first."""
        torch.cuda.synchronize()
        ovh = self.event_overhead_us * 1e-3
        by_kernel = {}
        for fam, name, label, s, e, fl, by in self.records:
            k = by_kernel.setdefault(name, {"family": fam, "all": [], "classes": {}})
            rec = (s.elapsed_time(e), fl, by)
            k["all"].append(rec)
            k["classes"].setdefault(label, []).append(rec)
        out = {}
        for name, k in by_kernel.items():
            st = self._stats(k["all"], ovh)
            st["family"] = k["family"]
            cls = {lab: self._stats(r, ovh) for lab, r in k["classes"].items() if lab}
            st["shape_classes"] = dict(sorted(cls.items(), key=lambda kv: -kv[1]["total_ms"]))
            out[name] = st
        return dict(sorted(out.items(), key=lambda kv: -kv[1]["total_ms"]))

    def families(self):
        """Backward-compatible view: {family: stats} over all instantiations of a family."""
        torch.cuda.synchronize()
        ovh = self.event_overhead_us * 1e-3
        fams = {}
        for fam, name, label, s, e, fl, by in self.records:
            fams.setdefault(fam, []).append((s.elapsed_time(e), fl, by))
        out = {f: self._stats(r, ovh) for f, r in fams.items()}
        return dict(sorted(out.items(), key=lambda kv: -kv[1]["total_ms"]))

    def roofline(self, kernel, stats):
        """The bench.py `roofline` object for one kernel.  The bound is decided by the ALGORITHMIC arithmetic intensity (FLOP per
        byte over the launches) against the ridge point peak_FLOP/s / peak_B/s (312.5 FLOP/B for bf16 MFMA, 625 for e4m3): above
        the ridge the matrix cores are the roof, below it HBM is - not by whichever fraction happens to read larger.  Both
        fractions are reported either way."""
        def one(st, fp8):
            peak = MFMA_FP8_PEAK_TFLOPS if fp8 else MFMA_BF16_PEAK_TFLOPS
            f_mfma = st["tflops"] / peak
            f_hbm = st["gbytes_per_s"] / HBM_PEAK_GBS
            intensity = st["algorithmic_flop_per_launch"] / max(st["algorithmic_bytes_per_launch"], 1)
            ridge = peak * 1e12 / (HBM_PEAK_GBS * 1e9)
            if intensity >= ridge:
                r = {"bound": "mfma", "achieved": st["tflops"], "peak": peak, "unit": "TFLOP/s", "frac": round(f_mfma, 4)}
            else:
                r = {"bound": "hbm", "achieved": st["gbytes_per_s"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(f_hbm, 4)}
            r.update({"traffic": None, "launches": st["launches"], "avg_launch_us": st["avg_launch_us"],
                      "total_ms": st["total_ms"], "algorithmic_bytes_per_launch": st["algorithmic_bytes_per_launch"],
                      "algorithmic_flop_per_launch": st["algorithmic_flop_per_launch"],
                      "mfma_frac": round(f_mfma, 4), "hbm_frac": round(f_hbm, 4),
                      "arithmetic_intensity_flop_per_byte": round(intensity, 1), "ridge_flop_per_byte": round(ridge, 1)})
            return r
        fp8 = "fp8" in stats.get("family", kernel) or "tn8" in stats.get("family", kernel) or \
            (kernel.startswith("gemm_nt_kernel<") and kernel.endswith((", 1>", ", 2>")))
        r = one(stats, fp8)
        r["kernel"] = kernel
        r["family"] = stats.get("family", kernel)
        r["shape_classes"] = [dict(one(st, fp8), shape=lab) for lab, st in stats.get("shape_classes", {}).items()]
        for c in r["shape_classes"]:
            c.pop("traffic", None)
        return r


PROFILE = KernelProfile()
