"""Tile-selection knobs of mmg_gemm_nt_bf16 on the shapes of the downsample layers, the stage-3 data gradient and BERT (child per knob set)."""
import os, sys, subprocess
here = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1:
    sys.path.insert(0, here)
    from gemm_bench import nt
    nt(4194304, 384, 192, "none"); nt(1048576, 384, 768, "bias"); nt(1048576, 384, 1536, "none")
    nt(10900, 2304, 768, "bias"); nt(10900, 768, 768, "res"); nt(10900, 3072, 768, "gelu+aux"); nt(10900, 768, 3072, "res"); nt(10900, 3072, 768, "dgelu")
    nt(19712, 2304, 768, "bias"); nt(19712, 768, 768, "res"); nt(19712, 3072, 768, "gelu+aux"); nt(19712, 768, 3072, "res")
else:
    for name, env in (("default", {}), ("K3=128", {"MMG_GEMM_K3": "128"}), ("K3=128 KBIG=4096", {"MMG_GEMM_K3": "128", "MMG_GEMM_KBIG": "4096"}),
                      ("default", {}), ("K3=128 KBIG=4096", {"MMG_GEMM_K3": "128", "MMG_GEMM_KBIG": "4096"})):
        print("==", name, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env))
