#!/usr/bin/env python3
"""NT GEMM shapes with N = 192 / 384: the 256 x 192 tile of round 4 (MMG_GEMM_192, default from K = 384) against the 128-wide tiles, interleaved
rounds in one process; results checked against each other and against torch on a row sample."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mmg-clip_amd")):
    sys.path.insert(0, p)
import torch                                 # noqa: E402
from mmgclip import linalg as L              # noqa: E402
dev = torch.device("cuda:0")


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        out = fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3, out


g = torch.Generator().manual_seed(0)
for M, N, K, bias, res in ((1048576, 384, 1536, False, False), (1048576, 384, 768, True, False), (4194304, 192, 384, True, False),
                           (262144, 384, 1536, False, True), (1048576, 576, 384, False, False), (37000, 384, 1536, False, False)):
    a = (torch.randn(M // 64, K, generator=g)).to(torch.bfloat16).to(dev).repeat(64, 1)[:M] if M % 64 == 0 else torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    b = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
    bi = torch.randn(N, device=dev) if bias else None
    rs = torch.randn(M, N, device=dev).to(torch.bfloat16) if res else None
    outs = {}
    line = f"M={M} N={N} K={K}{' +bias' if bias else ''}{' +res' if res else ''}:"
    for rnd in range(2):
        for mode in ("0", "384"):
            os.environ["MMG_GEMM_192"] = mode
            t, o = timed(lambda: L._gemm_nt_raw(a, b, bias=bi, residual=rs))
            outs[mode] = o
            line += f"  {'128-wide' if mode == '0' else '256x192'} {t:8.1f} us ({2.0 * M * N * K / t / 1e6:6.1f} TF/s)"
    ref = a[:512].float() @ b.float().t() + (bi if bias else 0) + (rs[:512].float() if res else 0)
    e0 = float((outs["0"][:512].float() - ref).abs().max() / ref.abs().max())
    e1 = float((outs["384"][:512].float() - ref).abs().max() / ref.abs().max())
    same = float((outs["0"].float() - outs["384"].float()).abs().max())
    print(line + f"   | rel err vs torch {e0:.2e} / {e1:.2e}, max |128 - 192| {same:.2e}", flush=True)
    del a, b, outs, o
