"""A/B of the two-workgroup-per-CU 256x128 NT tile (MMG_GEMM_2WG=1) on the epilogue-heavy shapes (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import nt
for env in ("0", "1"):
    os.environ["MMG_GEMM_2WG"] = env
    print("MMG_GEMM_2WG =", env, flush=True)
    for M, N, K, mode in ((1048576, 1536, 384, "dgelu"), (1048576, 1536, 384, "gelu+aux"), (1048576, 384, 1536, "none"), (1048576, 384, 1536, "res"),
                          (262144, 3072, 768, "dgelu"), (262144, 3072, 768, "gelu+aux"), (262144, 768, 3072, "none"), (1048576, 384, 768, "bias"),
                          (4194304, 384, 192, "none"), (4194304, 192, 384, "bias"), (10900, 3072, 768, "gelu+aux"), (10900, 768, 3072, "res")):
        nt(M, N, K, mode)
