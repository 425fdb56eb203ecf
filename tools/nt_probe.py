#!/usr/bin/env python3
"""Phase probe of gemm_nt_kernel (csrc/gemm_bf16.hip built with -DNT_PROBE into tools/libmmg_ab_ntprobe.so): shader-clock share of the
set-up, the main loop and the parts of the epilogue, of wave 0 of every workgroup, on the fat-epilogue shapes of the stage-3 / stage-4 backward."""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MMGCLIP_HIP_LIB", os.path.join(ROOT, "tools", "libmmg_ab_ntprobe.so"))
for p in (ROOT, os.path.join(ROOT, "mmg-clip_amd")):
    sys.path.insert(0, p)
import torch                                 # noqa: E402
from mmgclip import _hip, linalg as L        # noqa: E402

dev = torch.device("cuda:0")
lib = _hip.load()
fn = lib.mmg_debug_nt_probe
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
fn.restype = ctypes.c_int
names = ["whole kernel", "set-up", "main loop", "epi: barriers + acc -> LDS", "-", "epi: row math + stores issued (incl. waits for its reads)", "waves", "-"]
for M, N, K in ((1048576, 1536, 384), (262144, 3072, 768), (262144, 768, 3072)):
    a = torch.randn(M, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    hpre = torch.randn(M, N, device=dev).bfloat16()
    out, g = torch.empty(M, N, device=dev, dtype=torch.bfloat16), torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for mode, kw in (("dgelu", dict(epi=L.EPI_DGELU, aux_in=hpre, aux_out=g)), ("gelu+aux", dict(bias=torch.randn(N, device=dev), epi=L.EPI_GELU, aux_out=g)),
                     ("none", dict())):
        L._gemm_nt_raw(a, b, out=out, **kw)
        torch.cuda.synchronize()
        fn(None, 1)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        L._gemm_nt_raw(a, b, out=out, **kw)
        e.record()
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 8)()
        fn(buf, 0)
        v = list(buf)
        waves = max(v[6], 1)
        print(f"NT M={M} N={N} K={K} {mode}: {s.elapsed_time(e) * 1e3:.1f} us (probe build), {waves} waves, {v[0] / waves:.0f} cycles per wave", flush=True)
        for i in (1, 2, 3, 5):
            print(f"   {names[i]:38s} {v[i] / waves:9.0f} cycles  {100.0 * v[i] / max(v[0], 1):5.1f} %")
