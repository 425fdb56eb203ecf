import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import tn
print("env", {k: v for k, v in os.environ.items() if k.startswith("MMG_")})
tn(4194304, 96, 384); tn(4194304, 384, 96); tn(1048576, 192, 768); tn(1048576, 768, 192); tn(262144, 384, 1536); tn(262144, 1536, 384)
tn(65536, 768, 3072); tn(65536, 3072, 768); tn(19712, 768, 3072); tn(19712, 3072, 768); tn(19712, 768, 768); tn(19712, 768, 2304)
tn(65536, 768, 3072); tn(65536, 3072, 768); tn(19712, 768, 3072); tn(19712, 3072, 768); tn(19712, 768, 768); tn(19712, 768, 2304)
