#!/usr/bin/env python3
"""LayerNorm forward / backward on the C2 shapes under the generic kernels and the plain 3-chunk ones (MMG_LN_PLAIN=0 / 1; the knob is
read once per process: one child per setting, twice each)."""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(ROOT, "mmg-clip_amd"))
    import torch
    from mmgclip import kernels as K
    dev = torch.device("cuda:0")
    for M, C in ((1048576, 384), (4194304, 192), (262144, 768), (16777216, 96)):
        x = torch.randn(M // 16, C, device=dev).bfloat16().repeat(16, 1)
        dy = torch.randn(M // 16, C, device=dev).bfloat16().repeat(16, 1)
        gm, bt = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        y, mean, rstd = K.layernorm_fwd(x, gm, bt, 1e-6)
        dgm, dbt = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        res = []
        for name, fn in (("fwd", lambda: K.layernorm_fwd(x, gm, bt, 1e-6)), ("bwd", lambda: K.layernorm_bwd(dy, x, mean, rstd, gm, dgm, dbt))):
            fn(); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                fn()
            e.record(); torch.cuda.synchronize()
            us = s.elapsed_time(e) * 200
            gb = (4 if name == "fwd" else 6) * M * C / 1e9
            res.append(f"{name} {us:8.1f} us {gb / us * 1e3:6.2f} TB/s")
        print(f"  M={M:9d} C={C:4d}  " + "   ".join(res), flush=True)
else:
    for knob in ("0", "1", "0", "1"):
        print("== MMG_LN_PLAIN=" + knob, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env={**os.environ, "MMG_LN_PLAIN": knob})
