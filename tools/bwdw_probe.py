#!/usr/bin/env python3
"""Phase probe of the on-chip weight-gradient backward (csrc/cnblock_bwdw.hip built with -DBW_PROBE into tools/libmmg_ab_bwdw_probe.so):
shader-clock share of every phase, summed over waves.   MMGCLIP_HIP_LIB=tools/libmmg_ab_bwdw_probe.so python tools/bwdw_probe.py"""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MMGCLIP_HIP_LIB", os.path.join(ROOT, "tools", "libmmg_ab_bwdw_probe.so"))
for p in (ROOT, os.path.join(ROOT, "mmg-clip_amd")):
    sys.path.insert(0, p)
import torch                                 # noqa: E402
from mmgclip import _hip, kernels as K       # noqa: E402

dev = torch.device("cuda:0")
C, M = 96, int(os.environ.get("M", 16 * 256 * 256))
g = torch.Generator().manual_seed(0)
xd = torch.randn(M // 16, C, generator=g).to(torch.bfloat16).repeat(16, 1).to(dev)
dy = (0.5 * torch.randn(M // 16, C, generator=g)).to(torch.bfloat16).repeat(16, 1).to(dev)
lnw, lnb = (1 + 0.2 * torch.randn(C, generator=g)).to(dev), (0.1 * torch.randn(C, generator=g)).to(dev)
w1, b1 = (torch.randn(4 * C, C, generator=g) / C ** 0.5).to(dev), (0.1 * torch.randn(4 * C, generator=g)).to(dev)
w2 = (torch.randn(C, 4 * C, generator=g) / (4 * C) ** 0.5).to(dev)
ls = (0.3 + 0.7 * torch.rand(C, generator=g)).to(dev)
z = lambda *s: torch.zeros(*s, device=dev)   # noqa: E731
packed, b1f = K.cnblock_bwdw_pack(w1, w2, lnw, lnb, ls, b1)
lib = _hip.load()
fn = lib.mmg_debug_bwdw_probe
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
fn.restype = ctypes.c_int
run = lambda: K.cnblock_bwdw(dy, xd, lnw, lnb, 1e-6, packed, b1f, z(4 * C, C), z(4 * C), z(C, 4 * C), z(C), z(C), z(C))   # noqa: E731
run(); torch.cuda.synchronize()
fn(None, 1)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); run(); e.record(); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 32)()
fn(buf, 0)
names = ["tile loop", "P0 stage rows", "wait A", "P1 products", "P1 load issue", "P1 GELU", "P1 weight grad", "wait B", "P2 dLN", "wait C1+C2", "P3 LN bwd", "waves"]
print(f"M={M}: both launches {s.elapsed_time(e) * 1e3:.1f} us (probe build)")
tiles = M // 64
for mode in range(2):
    v = [buf[mode * 16 + i] for i in range(12)]
    waves = max(v[11], 1)
    per_tile = lambda x: x / waves / (tiles / (waves / 8))    # noqa: E731  (cycles per wave per tile)
    print(f"launch {mode + 1}: {waves} waves, {v[0] / waves:.0f} cycles per wave = {per_tile(v[0]):.0f} per tile")
    for i in range(1, 11):
        if v[i]:
            print(f"   {names[i]:16s} {per_tile(v[i]):9.0f} cycles/tile  {100.0 * v[i] / v[0]:5.1f} %")
