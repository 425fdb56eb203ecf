import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import nt
print("env", {k: v for k, v in os.environ.items() if k.startswith("MMG_")})
for mode in ("none", "gelu+aux", "dgelu"):
    nt(262144, 1536, 384, mode)
