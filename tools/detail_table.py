#!/usr/bin/env python3
"""Per-kernel table of a bench.py detail file (gpurun_out/bench_detail.json): ms per step, launches, average, roofline fractions."""
import json, sys
d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/bench_detail.json"))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
steps = d["roofline_method"]["profiled_steps"]
print(d["value"], d["unit"], d["ms_per_step"], "ms/step;", d["roofline_method"]["instrumented_kernel_ms_per_step"], "ms of instrumented kernels (one stream)")
for k in ([d["roofline"]] + d["roofline_other_kernels"])[:n]:
    print(f"{k['kernel'][:54]:54s} {k['ms_per_step']:8.2f} ms {k['launches'] // steps:4d}/step {k['avg_launch_us']:9.1f} us  {k['bound']:4s} {k['frac']:.3f}  mfma {k['mfma_frac']:.3f} hbm {k['hbm_frac']:.3f}")
