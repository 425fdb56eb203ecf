"""A/B of two library builds on the fused CNBlock MLP kernels, alternating child processes (run on the GPU box)."""
import os, sys, subprocess
here = os.path.dirname(os.path.abspath(__file__))
for rnd in range(2):
    for name, lib in (("new", None), ("old", os.path.join(here, "libmmg_ab_old.so"))):
        env = dict(os.environ)
        if lib: env["MMGCLIP_HIP_LIB"] = lib
        for script in ("mlp_bench.py", "mlp_bwd_bench.py"):
            print("==", name, script, flush=True)
            r = subprocess.run([sys.executable, os.path.join(here, script)], env=env, capture_output=True, text=True)
            print("\n".join(l for l in r.stdout.splitlines() if " fused " in l), flush=True)
