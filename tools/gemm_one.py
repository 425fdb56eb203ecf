#!/usr/bin/env python3
"""Launch one GEMM configuration a few times (for rocprofv3 --pmc runs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mmg-clip_amd"))
import torch
from mmgclip import linalg as L
M, N, K, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
dev = torch.device("cuda:0")
a = torch.randn(M, K, device=dev).bfloat16(); b = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16); bias = torch.randn(N, device=dev)
kw = {}
if mode == "gelu+aux": kw = dict(bias=bias, epi=L.EPI_GELU, aux_out=torch.empty(M, N, device=dev, dtype=torch.bfloat16))
for _ in range(5):
    L._gemm_nt_raw(a, b, out=out, **kw)
torch.cuda.synchronize()
