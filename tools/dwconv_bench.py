#!/usr/bin/env python3
"""Micro-benchmark of the depthwise 7x7 kernels on the ConvNeXt-T stage shapes at 1024^2 input, 16 images."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mmg-clip_amd"))
import torch
from mmgclip import kernels as K
dev = torch.device("cuda:0")

def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

print("env", {k: v for k, v in os.environ.items() if k.startswith("MMG_")})
for n, H, C in [(16, 256, 96), (16, 128, 192), (16, 64, 384), (16, 32, 768)]:
    x = torch.randn(n * H * H, C, device=dev).bfloat16()
    dy = torch.randn(n * H * H, C, device=dev).bfloat16()
    w = torch.randn(49, C, device=dev) * 0.1
    b = torch.randn(C, device=dev)
    out = torch.empty_like(x)
    dw, db = torch.zeros(49, C, device=dev), torch.zeros(C, device=dev)
    fl = 2.0 * 49 * n * H * H * C
    for rnd in range(2):          # MFMA (csrc/dwconv7_mfma.hip) against VALU (csrc/dwconv7.hip), interleaved in one process
        line = f"DW n={n} H={H} C={C} round {rnd}:"
        for mode in ("0", "1"):
            os.environ["MMG_DWCONV_MFMA"] = mode
            t1 = timeit(lambda: K.dwconv7(x, w, b, n, H, H, C, out=out))
            t2 = timeit(lambda: K.dwconv7(dy, w, None, n, H, H, C, add=x, flip=True, out=out))
            line += f"  {'mfma' if mode == '1' else 'valu'}: fwd {t1:7.1f} us {fl/t1/1e6:6.1f} TF/s, bwd-data {t2:7.1f} us {fl/t2/1e6:6.1f} TF/s |"
        print(line, flush=True)
    t3 = timeit(lambda: K.dwconv7_wgrad(x, dy, dw, db, n, H, H, C))
    print(f"DW n={n} H={H} C={C}: wgrad {t3:7.1f} us {fl/t3/1e6:6.1f} TF/s", flush=True)
