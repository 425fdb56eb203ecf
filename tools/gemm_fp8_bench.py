"""bf16 vs e4m3 NT GEMM on the ConvNeXt-B stage-3 / stage-4 block shapes (one 64-image micro-batch at 1024^2)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mmg-clip_amd")]
from mmgclip import linalg as L  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for M, C in ((64 * 64 * 64, 512), (64 * 32 * 32, 1024)):
    x = torch.randn(M, C, device=dev).bfloat16()
    w1 = (torch.randn(4 * C, C, device=dev) * 0.02).bfloat16()
    w2 = (torch.randn(C, 4 * C, device=dev) * 0.02).bfloat16()
    b1, b2, cs = torch.zeros(4 * C, device=dev), torch.zeros(C, device=dev), torch.ones(C, device=dev)
    res = torch.randn(M, C, device=dev).bfloat16()
    hpre = torch.empty(M, 4 * C, device=dev, dtype=torch.bfloat16)
    h = torch.empty(M, 4 * C, device=dev, dtype=torch.bfloat16)
    y = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
    x8 = x.float().clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    w18 = (w1.float() * 64).to(torch.float8_e4m3fn).view(torch.uint8)
    w28 = (w2.float() * 64).to(torch.float8_e4m3fn).view(torch.uint8)
    h8 = torch.empty(M, 4 * C, device=dev, dtype=torch.uint8)
    fl = 2.0 * M * C * 4 * C
    rows = [
        ("bf16 fc1 gelu +hpre", lambda: L.gemm_nt(x, w1, out=h, bias=b1, epi=L.EPI_GELU, aux_out=hpre)),
        ("fp8  fc1 gelu +hpre", lambda: L.gemm_nt_fp8(x8, w18, out=h8, bias=b1, epi=L.EPI_GELU, aux_out=hpre, out_kind=L.OUT_E4M3, alpha=1 / 64)),
        ("bf16 fc1 gelu      ", lambda: L.gemm_nt(x, w1, out=h, bias=b1, epi=L.EPI_GELU)),
        ("fp8  fc1 gelu      ", lambda: L.gemm_nt_fp8(x8, w18, out=h8, bias=b1, epi=L.EPI_GELU, out_kind=L.OUT_E4M3, alpha=1 / 64)),
        ("bf16 fc2 ls + res  ", lambda: L.gemm_nt(h, w2, out=y, bias=b2, colscale=cs, residual=res)),
        ("fp8  fc2 ls + res  ", lambda: L.gemm_nt_fp8(h8, w28, out=y, bias=b2, colscale=cs, residual=res, alpha=1 / 64)),
    ]
    for name, fn in rows:
        ms = timeit(fn)
        print(f"M={M} C={C} {name}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)
