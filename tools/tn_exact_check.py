#!/usr/bin/env python3
"""Do the weight-gradient GEMMs read what is in LDS?  Small-integer operands make every product and every fp32 partial sum exact whatever the order
of the atomics, so ONE wrong operand byte anywhere in a long reduction under full load shows up as an inequality.  Written after the tiled attention
kernels turned out to read transposed fragments unreliably while an LDS-DMA was in flight (round 4): gemm_tn_wide (bf16) and gemm_tn8_wide do
exactly that by design (inline-assembly ds_read_tr under a running DMA ring)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mmg-clip_amd")]
import torch                                   # noqa: E402
from mmgclip import linalg as L                # noqa: E402
dev = torch.device("cuda:0")


def ints(shape, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi + 1, shape, generator=g).float().to(dev)


bad = 0
for (M, N1, N2) in ((262144, 512, 2048), (262144, 2048, 512), (65536, 1024, 4096), (1048576, 256, 1024)):
    a, b = ints((M, N1), -2, 2, 1), ints((M, N2), -1, 1, 2)
    ref = a.t() @ b                                        # exact: |sums| <= 2 M < 2^24
    a8 = a.to(torch.float8_e5m2).view(torch.uint8)
    b8 = b.to(torch.float8_e4m3fn).view(torch.uint8)
    for rep in range(3):
        out = torch.zeros(N1, N2, device=dev)
        cs = torch.zeros(N1, device=dev)
        L.gemm_tn_fp8_acc(a8, b8, out, colsum=cs)
        nbad = int((out != ref).sum()) + int((cs != a.sum(0)).sum())
        bad += nbad
        print(f"8-bit  M={M} {N1}x{N2} rep {rep}: {nbad} wrong elements", flush=True)
    del a8, b8
    a16, b16 = a.bfloat16(), b.bfloat16()
    for rep in range(3):
        out = torch.zeros(N1, N2, device=dev)
        cs = torch.zeros(N1, device=dev)
        L.gemm_tn_acc(a16, b16, out, colsum=cs)
        nbad = int((out != ref).sum()) + int((cs != a.sum(0)).sum())
        bad += nbad
        print(f"bf16   M={M} {N1}x{N2} rep {rep}: {nbad} wrong elements", flush=True)
    del a, b, a16, b16, ref
for (M, N1, N2) in ((4194304, 96, 384), (4194304, 192, 768), (1048576, 384, 1536), (1048576, 1536, 384)):      # ConvNeXt-T stage shapes (gemm_tn_wide)
    a, b = ints((M, N1), -1, 1, 3).bfloat16(), ints((M, N2), -1, 1, 4).bfloat16()
    ref = a.float().t() @ b.float()
    for rep in range(3):
        out = torch.zeros(N1, N2, device=dev)
        cs = torch.zeros(N1, device=dev)
        L.gemm_tn_acc(a, b, out, colsum=cs)
        nbad = int((out != ref).sum()) + int((cs != a.float().sum(0)).sum())
        bad += nbad
        print(f"bf16   M={M} {N1}x{N2} rep {rep}: {nbad} wrong elements", flush=True)
    del a, b, ref
print("TOTAL wrong elements:", bad)
sys.exit(1 if bad else 0)
