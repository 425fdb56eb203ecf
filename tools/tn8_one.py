#!/usr/bin/env python3
"""One 8-bit weight-gradient GEMM shape, a few launches (driver for tools/collect_gemm_pmc.sh: PMC_PROG=tools/tn8_one.py ... M N1 N2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mmg-clip_amd")]
import torch
from mmgclip import linalg as L
dev = torch.device("cuda:0")
M, N1, N2 = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (262144, 512, 2048)))
g = torch.Generator().manual_seed(0)
a8 = (torch.randn(M // 64, N1, generator=g) * 2).to(torch.float8_e5m2).view(torch.uint8).to(dev).repeat(64, 1).contiguous()
b8 = torch.randn(M // 64, N2, generator=g).to(torch.float8_e4m3fn).view(torch.uint8).to(dev).repeat(64, 1).contiguous()
out, cs = torch.zeros(N1, N2, device=dev), torch.zeros(N1, device=dev)
for _ in range(4):
    L.gemm_tn_fp8_acc(a8, b8, out, colsum=cs)
torch.cuda.synchronize()
