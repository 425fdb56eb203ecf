#!/usr/bin/env python3
"""Micro-benchmark of the GEMM entry points on the shapes of the C2 step (run on the GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mmg-clip_amd"))
import torch  # noqa: E402
from mmgclip import linalg as L  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3   # us


def nt(M, N, K, mode):
    a = torch.randn(M, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bias = torch.randn(N, device=dev)
    kw = {}
    nbytes = 2 * (M * K + N * K + M * N)
    if mode == "gelu+aux":
        kw = dict(bias=bias, epi=L.EPI_GELU, aux_out=torch.empty(M, N, device=dev, dtype=torch.bfloat16))
        nbytes += 2 * M * N
    elif mode == "gelu":
        kw = dict(bias=bias, epi=L.EPI_GELU)
    elif mode == "dgelu":
        kw = dict(epi=L.EPI_DGELU, aux_in=torch.randn(M, N, device=dev).bfloat16())
        nbytes += 2 * M * N
    elif mode == "res":
        kw = dict(bias=bias, colscale=bias, residual=torch.randn(M, N, device=dev).bfloat16())
        nbytes += 2 * M * N
    elif mode == "bias":
        kw = dict(bias=bias)
    us = timeit(lambda: L._gemm_nt_raw(a, b, out=out, **kw))
    print(f"NT M={M:8d} N={N:5d} K={K:5d} {mode:9s} {us:9.1f} us  {2.0 * M * N * K / us / 1e6:8.1f} TFLOP/s  {nbytes / us / 1e3:8.1f} GB/s", flush=True)


def tn(M, N1, N2):
    a = torch.randn(M, N1, device=dev).bfloat16()
    b = torch.randn(M, N2, device=dev).bfloat16()
    out = torch.zeros(N1, N2, device=dev)
    cs = torch.zeros(N1, device=dev)
    us = timeit(lambda: L.gemm_tn_acc(a, b, out, colsum=cs))
    print(f"TN M={M:8d} N1={N1:5d} N2={N2:5d} {us:9.1f} us  {2.0 * M * N1 * N2 / us / 1e6:8.1f} TFLOP/s  {2 * M * (N1 + N2) / us / 1e3:8.1f} GB/s", flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "nt"):
        for mode in ("none", "bias", "gelu", "gelu+aux", "dgelu", "res"):
            nt(65536, 1536, 384, mode)
        nt(65536, 384, 1536, "res")
        nt(65536, 384, 1536, "none")
        for mode in ("none", "gelu+aux"):
            nt(1048576, 384, 96, mode)
            nt(262144, 768, 192, mode)
            nt(16384, 3072, 768, mode)
        nt(1048576, 96, 384, "res")
        nt(262144, 192, 768, "res")
        nt(16384, 768, 3072, "res")
        nt(19712, 2304, 768, "bias")
        nt(19712, 3072, 768, "gelu+aux")
        nt(19712, 768, 3072, "res")
        nt(8192, 8192, 8192, "none")
    if which in ("all", "tn"):
        tn(1048576, 96, 384)
        tn(1048576, 384, 96)
        tn(262144, 192, 768)
        tn(65536, 384, 1536)
        tn(65536, 1536, 384)
        tn(16384, 768, 3072)
        tn(19712, 768, 3072)
        tn(19712, 2304, 768)
