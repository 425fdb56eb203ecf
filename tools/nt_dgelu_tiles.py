#!/usr/bin/env python3
"""The fat-epilogue NT shapes of the stage-3 / stage-4 backward on the tiles the dispatcher can choose (one child process per knob set:
the knobs are read once per process)."""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(ROOT, "mmg-clip_amd"))
    import torch
    from mmgclip import linalg as L
    dev = torch.device("cuda:0")
    for M, N, K in ((1048576, 1536, 384), (262144, 3072, 768)):
        a = torch.randn(M, K, device=dev).bfloat16()
        b = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        hpre = torch.randn(M, N, device=dev).bfloat16()
        out, g = torch.empty(M, N, device=dev, dtype=torch.bfloat16), torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        for mode, kw in (("dgelu", dict(epi=L.EPI_DGELU, aux_in=hpre, aux_out=g)), ("gelu+aux", dict(bias=torch.randn(N, device=dev), epi=L.EPI_GELU, aux_out=g))):
            for _ in range(2):
                L._gemm_nt_raw(a, b, out=out, **kw)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                L._gemm_nt_raw(a, b, out=out, **kw)
            e.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(e) * 200
            print(f"  NT M={M} N={N} K={K} {mode:9s} {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
else:
    sets = [{}, {"MMG_GEMM_256": "0"}, {"MMG_GEMM_256": "0", "MMG_GEMM_KBIG": "128"}, {}]
    # A/B builds of gemm_bf16.hip (linked into tools/libmmg_ab_<name>.so): --lib ntexact = -DNT_GELU_EXACT (rcp / exp GELU forms instead of the
    # polynomials), --lib ntlate = -DNT_EPI_READS_LATE (a slab's epilogue reads at the top of the slab), --lib ntsync = -DNT_EPI_SYNCTHREADS
    # (slab barriers as __syncthreads(), i.e. with vmcnt(0): round 2), --lib ntold = the file as of the previous commit (tools/nt_epi_ab.sh)
    if "--lib" in sys.argv:
        alt = os.path.join(ROOT, "tools", "libmmg_ab_%s.so" % sys.argv[sys.argv.index("--lib") + 1])
        sets = [{}, {"MMGCLIP_HIP_LIB": alt}, {}, {"MMGCLIP_HIP_LIB": alt}]
    for knobs in sets:
        print("==", knobs or "default", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env={**os.environ, **knobs})
