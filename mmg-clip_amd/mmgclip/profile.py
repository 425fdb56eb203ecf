"""HIP-event timing of the hot kernels, per kernel family (bench.py's roofline leg).

Events are recorded on the stream the kernel is launched on (torch's current stream), so a duration is the kernel's own;
`flops` / `bytes` are the ALGORITHMIC figures of the launch: 2*M*N*K per GEMM, operands read once + outputs
written once.  Off by default: a disabled profiler costs one attribute test per launch."""
import torch

MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense bf16, MI355X_MICROARCH.md
MFMA_FP8_PEAK_TFLOPS = 5000.0       # dense e4m3 on the K = 128 MFMA, same guide
HBM_PEAK_GBS = 8000.0


class KernelProfile:
    def __init__(self):
        self.on = False
        self.records = {}            # family -> [(start, end, flops, bytes)]

    def enable(self):
        self.on, self.records = True, {}

    def disable(self):
        self.on = False

    def timed(self, family, flops, nbytes, fn):
        if not self.on:
            return fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = fn()
        e.record()
        self.records.setdefault(family, []).append((s, e, float(flops), float(nbytes)))
        return r

    def families(self):
        """{family: dict(launches, total_ms, tflops, gbytes_per_s, ...)} sorted by total time, largest first."""
        torch.cuda.synchronize()
        out = {}
        for fam, recs in self.records.items():
            t_ms = sum(s.elapsed_time(e) for s, e, _, _ in recs)
            fl, by = sum(r[2] for r in recs), sum(r[3] for r in recs)
            n = len(recs)
            out[fam] = {"launches": n, "total_ms": round(t_ms, 2), "avg_launch_us": round(t_ms * 1e3 / n, 1),
                        "tflops": round(fl / (t_ms * 1e-3) / 1e12, 1), "gbytes_per_s": round(by / (t_ms * 1e-3) / 1e9, 1),
                        "algorithmic_flop_per_launch": round(fl / n), "algorithmic_bytes_per_launch": round(by / n)}
        return dict(sorted(out.items(), key=lambda kv: -kv[1]["total_ms"]))

    def roofline(self, family, stats):
        """The bench.py `roofline` object for one family.  The bound is decided by the family's ALGORITHMIC arithmetic intensity
        (FLOP per byte over all its launches) against the ridge point peak_FLOP/s / peak_B/s (312.5 FLOP/B for bf16 MFMA, 625 for
        e4m3): above the ridge the matrix cores are the roof, below it HBM is - not by whichever fraction happens to read larger.
        Both fractions are reported either way."""
        peak = MFMA_FP8_PEAK_TFLOPS if "fp8" in family else MFMA_BF16_PEAK_TFLOPS
        f_mfma = stats["tflops"] / peak
        f_hbm = stats["gbytes_per_s"] / HBM_PEAK_GBS
        intensity = stats["algorithmic_flop_per_launch"] / max(stats["algorithmic_bytes_per_launch"], 1)
        ridge = peak * 1e12 / (HBM_PEAK_GBS * 1e9)
        if intensity >= ridge:
            r = {"bound": "mfma", "achieved": stats["tflops"], "peak": peak, "unit": "TFLOP/s", "frac": round(f_mfma, 4)}
        else:
            r = {"bound": "hbm", "achieved": stats["gbytes_per_s"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(f_hbm, 4)}
        r.update({"kernel": family, "traffic": None, "launches": stats["launches"], "avg_launch_us": stats["avg_launch_us"],
                  "total_ms": stats["total_ms"], "algorithmic_bytes_per_launch": stats["algorithmic_bytes_per_launch"],
                  "algorithmic_flop_per_launch": stats["algorithmic_flop_per_launch"],
                  "mfma_frac": round(f_mfma, 4), "hbm_frac": round(f_hbm, 4),
                  "arithmetic_intensity_flop_per_byte": round(intensity, 1), "ridge_flop_per_byte": round(ridge, 1)})
        return r


PROFILE = KernelProfile()
