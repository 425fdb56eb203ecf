import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import nt
print("env", {k: v for k, v in os.environ.items() if k.startswith("MMG_")})
for N, K, mode in ((768, 768, "res"), (2304, 768, "bias"), (3072, 768, "gelu+aux"), (768, 3072, "res"), (768, 2304, "none"), (3072, 768, "dgelu")):
    nt(10416, N, K, mode)
