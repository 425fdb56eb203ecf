#!/usr/bin/env python3
"""Matrix-core vs VALU depthwise forward / data gradient on the ConvNeXt-B stage shapes at 1024^2 input, 64-image micro-batch (config C5)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mmg-clip_amd"))
import torch
from mmgclip import kernels as K
dev = torch.device("cuda:0")


def timeit(fn, iters=4):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


n = 64
for H, C in ((256, 128), (128, 256), (64, 512), (32, 1024)):
    x = torch.randn(n * H * H, C, device=dev).bfloat16()
    w = torch.randn(49, C, device=dev) * 0.1
    b = torch.randn(C, device=dev)
    out = torch.empty_like(x)
    line = f"H={H} C={C} n={n}:"
    for mode in ("0", "1"):
        os.environ["MMG_DWCONV_MFMA"] = mode
        t1 = timeit(lambda: K.dwconv7(x, w, b, n, H, H, C, out=out))
        t2 = timeit(lambda: K.dwconv7(x, w, None, n, H, H, C, add=x, flip=True, out=out))
        line += f"  {'mfma' if mode == '1' else 'valu'} fwd {t1:8.1f} us  dgrad {t2:8.1f} us |"
    print(line, flush=True)
    del x, out
