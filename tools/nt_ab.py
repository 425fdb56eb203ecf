"""A/B of two builds of the library on the activation-epilogue NT shapes, alternating child processes (run on the GPU box)."""
import os, sys, subprocess
here = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1:
    sys.path.insert(0, here)
    from gemm_bench import nt
    for _ in range(2):
        nt(1048576, 1536, 384, "dgelu"); nt(1048576, 1536, 384, "gelu+aux"); nt(262144, 3072, 768, "dgelu")
else:
    for rnd in range(2):
        for name, lib in (("new", None), ("old-gelu_both", os.path.join(here, "libmmg_ab_old.so"))):
            env = dict(os.environ)
            if lib: env["MMGCLIP_HIP_LIB"] = lib
            print("==", name, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env)
