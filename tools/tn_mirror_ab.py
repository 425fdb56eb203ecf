"""A/B of the wide weight-gradient kernel on N1 > N2 shapes: exchanged operands + transposed flush (default) against the mirrored
384 x 192 / 384 x 96 instantiations (MMG_TN_WIDE_MIRROR=1), alternating child processes (run on the GPU box)."""
import os, sys, subprocess
here = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1:
    sys.path.insert(0, here)
    from gemm_bench import tn
    tn(16777216, 384, 96); tn(16777216, 96, 384); tn(4194304, 768, 192); tn(4194304, 192, 768); tn(1048576, 1536, 384); tn(1048576, 384, 1536)
    tn(262144, 3072, 768); tn(262144, 768, 3072)
else:
    for rnd in range(2):
        for v in ("0", "1"):
            print("== MMG_TN_WIDE_MIRROR=" + v, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, MMG_TN_WIDE_MIRROR=v))
