"""A/B of several builds of the library on the activation-epilogue NT shapes, alternating child processes (run on the GPU box)."""
import os, sys, subprocess, glob
here = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1:
    sys.path.insert(0, here)
    from gemm_bench import nt
    nt(1048576, 1536, 384, "dgelu"); nt(262144, 3072, 768, "gelu+aux"); nt(262144, 768, 3072, "res")
else:
    libs = [("current", None)] + [(os.path.basename(p), p) for p in sorted(glob.glob(os.path.join(here, "libmmg_ab_*.so")))]
    for rnd in range(2):
        for name, lib in libs:
            env = dict(os.environ)
            if lib: env["MMGCLIP_HIP_LIB"] = lib
            print("==", name, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env)
