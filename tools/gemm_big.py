import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import nt
print("env", {k: v for k, v in os.environ.items() if k.startswith("MMG_")})
nt(262144, 1536, 384, "dgelu"); nt(262144, 1536, 384, "none")
nt(65536, 3072, 768, "gelu+aux"); nt(65536, 3072, 768, "dgelu"); nt(65536, 768, 3072, "res"); nt(65536, 768, 3072, "none")
nt(19712, 2304, 768, "bias"); nt(19712, 3072, 768, "gelu+aux"); nt(19712, 768, 3072, "bias"); nt(19712, 768, 768, "bias")
