#!/usr/bin/env python3
"""Timing ablations of the matrix-core depthwise kernel (csrc/dwconv7_mfma.hip, MMG_DWM_DBG): which phase of an item costs what?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mmg-clip_amd"))
import torch
from mmgclip import kernels as K
dev = torch.device("cuda:0")
n, H, C = 16, 256, 96
x = torch.randn(n * H * H, C, device=dev).bfloat16()
w = torch.randn(49, C, device=dev) * 0.1
b = torch.randn(C, device=dev)
out = torch.empty_like(x)
os.environ["MMG_DWCONV_MFMA"] = "1"


def timeit(iters=10):
    K.dwconv7(x, w, b, n, H, H, C, out=out); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        K.dwconv7(x, w, b, n, H, H, C, out=out)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


names = {0: "everything", 1: "no B (transpose in)", 2: "no D (MFMA)", 4: "no E (transpose out)", 8: "no F (stores)", 16: "no global loads",
         3: "no B, D", 7: "no B, D, E", 15: "no B, D, E, F", 31: "nothing but deposit + barriers", 24: "no loads, no stores"}
for rnd in range(2):
    for m, nm in names.items():
        os.environ["MMG_DWM_DBG"] = str(m)
        print(f"round {rnd}  dbg {m:2d}  {nm:32s} {timeit():8.1f} us", flush=True)
