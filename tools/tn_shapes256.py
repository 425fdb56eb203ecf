"""Weight-gradient (TN) GEMM shapes of one 256-image micro-batch of the C2 step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import tn
print("env", {k: v for k, v in os.environ.items() if k.startswith("MMG_")})
for M, C in ((16777216, 96), (4194304, 192), (1048576, 384), (262144, 768)):
    tn(M, C, 4 * C)      # dW2 = dy^T g
    tn(M, 4 * C, C)      # dW1 = dh^T ln
