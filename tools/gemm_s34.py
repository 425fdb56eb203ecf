"""In-situ shapes of stages 3/4 (micro-batch 64) for NT and TN (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import nt, tn
for M, C in ((262144, 384), (65536, 768)):
    nt(M, 4 * C, C, "none")
    nt(M, 4 * C, C, "gelu+aux")
    nt(M, 4 * C, C, "dgelu")
    nt(M, C, 4 * C, "res")
    nt(M, C, 4 * C, "none")
    tn(M, C, 4 * C)
    tn(M, 4 * C, C)
nt(64 * 77, 2304, 768, "bias"); nt(64 * 77 * 4, 2304, 768, "bias"); nt(64 * 77 * 4, 3072, 768, "gelu+aux"); nt(64 * 77 * 4, 768, 3072, "bias")
tn(64 * 77 * 4, 768, 3072); tn(64 * 77 * 4, 3072, 768); tn(64 * 77 * 4, 768, 768)
