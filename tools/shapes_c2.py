"""In-situ GEMM shapes of the C2 step (256 images of 1024^2 in one micro-batch): per-shape time of the TN / NT entry points."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import nt, tn
n = 256
for si, C in enumerate((96, 192, 384, 768)):
    M = n * (256 >> si) ** 2
    print(f"--- stage {si + 1}: C={C} M={M}")
    tn(M, C, 4 * C)
    tn(M, 4 * C, C)
    if C >= 384:
        nt(M, 4 * C, C, "gelu+aux")
        nt(M, C, 4 * C, "res")
        nt(M, 4 * C, C, "dgelu")
        nt(M, C, 4 * C, "none")
for si, C in enumerate((96, 192, 384)):
    M = n * (128 >> si) ** 2
    print(f"--- downsample {si}: {4 * C} -> {2 * C}, M={M}")
    nt(M, 2 * C, 4 * C, "bias")
    nt(M, 4 * C, 2 * C, "none")
    tn(M, 2 * C, 4 * C)
T = 10900
print("--- BERT packed tokens ~", T)
nt(T, 2304, 768, "bias"); nt(T, 768, 768, "res"); nt(T, 3072, 768, "gelu+aux"); nt(T, 768, 3072, "res"); nt(T, 3072, 768, "dgelu")
tn(T, 768, 3072); tn(T, 3072, 768); tn(T, 768, 768); tn(T, 2304, 768)
