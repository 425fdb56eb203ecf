"""A/B of the depthwise forward / data-gradient tile height (MMG_DWCONV_TH = 8 / 16), alternating child processes (run on the GPU box)."""
import os, sys, subprocess
here = os.path.dirname(os.path.abspath(__file__))
for rnd in range(2):
    for th in ("16", "8"):
        print("== MMG_DWCONV_TH=" + th, flush=True)
        r = subprocess.run([sys.executable, os.path.join(here, "dwconv_bench.py")], env=dict(os.environ, MMG_DWCONV_TH=th), capture_output=True, text=True)
        print("\n".join(l for l in r.stdout.splitlines() if l.startswith("DW")), flush=True)
