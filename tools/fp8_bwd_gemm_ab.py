#!/usr/bin/env python3
"""The four backward GEMMs of a ConvNeXt-B stage-3 / stage-4 block (config C5 shapes, 64-image micro-batch) in bf16 and on 8-bit operands, one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mmg-clip_amd")):
    sys.path.insert(0, p)
import torch                                 # noqa: E402
from mmgclip import linalg as L, kernels as K              # noqa: E402
dev = torch.device("cuda:0")


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


g = torch.Generator().manual_seed(0)
for M, C in ((262144, 512), (65536, 1024)):
    H4 = 4 * C
    dx = (torch.randn(M // 64, C, generator=g) * 1e-3).to(torch.bfloat16).to(dev).repeat(64, 1).contiguous()
    aux = torch.rand(M // 64, H4, generator=g).to(torch.bfloat16).to(dev).repeat(64, 1).contiguous()
    gact = torch.randn(M // 64, H4, generator=g).to(torch.bfloat16).to(dev).repeat(64, 1).contiguous()
    ln = torch.randn(M // 64, C, generator=g).to(torch.bfloat16).to(dev).repeat(64, 1).contiguous()
    w2gt = (torch.randn(H4, C, generator=g) / C ** 0.5).to(dev)
    w1t = (torch.randn(C, H4, generator=g) / H4 ** 0.5).to(dev)
    w2gt16, w1t16 = w2gt.to(torch.bfloat16), w1t.to(torch.bfloat16)
    w2gt8, s2 = K.quantize_e4m3(w2gt); w1t8, s1 = K.quantize_e4m3(w1t)
    gact8 = gact.float().clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    ln8 = ln.float().to(torch.float8_e4m3fn).view(torch.uint8)
    dW2, dW1 = torch.zeros(C, H4, device=dev), torch.zeros(H4, C, device=dev)
    db = torch.zeros(H4, device=dev)
    # bf16
    t_q = 0.0
    dh = L.gemm_nt(dx, w2gt16, epi=L.EPI_MUL_AUX, aux_in=aux)
    t1 = timed(lambda: L.gemm_nt(dx, w2gt16, epi=L.EPI_MUL_AUX, aux_in=aux))
    t2 = timed(lambda: L.gemm_nt(dh, w1t16))
    t3 = timed(lambda: L.gemm_tn_acc(dx, gact, dW2))
    t4 = timed(lambda: L.gemm_tn_acc(dh, ln, dW1, colsum=db))
    print(f"M={M} C={C} bf16 : dh {t1:7.1f}  dln {t2:7.1f}  dW2 {t3:7.1f}  dW1 {t4:7.1f}  sum {t1 + t2 + t3 + t4:8.1f} us", flush=True)
    dy8, sdy = K.quantize_e5m2(dx)
    t_q = timed(lambda: K.quantize_e5m2(dx))
    dh8 = L.gemm_nt_fp8_bwd(dy8, w2gt8, aux_in=aux, epi=L.EPI_MUL_AUX, out_kind=L.OUT_E5M2, alpha_dev=s2[1:])
    u1 = timed(lambda: L.gemm_nt_fp8_bwd(dy8, w2gt8, aux_in=aux, epi=L.EPI_MUL_AUX, out_kind=L.OUT_E5M2, alpha_dev=s2[1:]))
    u2 = timed(lambda: L.gemm_nt_fp8_bwd(dh8, w1t8, alpha_dev=s1[1:], alpha_dev2=sdy[1:]))
    u3 = timed(lambda: L.gemm_tn_fp8_acc(dy8, gact8, dW2, alpha_dev=sdy[1:]))
    u4 = timed(lambda: L.gemm_tn_fp8_acc(dh8, ln8, dW1, alpha_dev=sdy[1:], colsum=db))
    print(f"M={M} C={C} 8-bit: dh {u1:7.1f}  dln {u2:7.1f}  dW2 {u3:7.1f}  dW1 {u4:7.1f}  quantize dy {t_q:6.1f}  sum {u1 + u2 + u3 + u4 + t_q:8.1f} us", flush=True)
    # the weight-gradient kernel's forms (csrc/gemm_tn_fp8.hip): 128 x 128 tiles on a (tile, chunk) grid, the same with a row chunk's tiles on one XCD, 256 x 256 tiles
    for tag, env in (("128 grid", {"MMG_TN8_WIDE": "0", "MMG_TN8_XCD": "0"}), ("128 xcd ", {"MMG_TN8_WIDE": "0", "MMG_TN8_XCD": "1"}), ("256 xcd ", {"MMG_TN8_WIDE": "1"})):
        for k in ("MMG_TN8_WIDE", "MMG_TN8_XCD"):
            os.environ.pop(k, None)
        os.environ.update(env)
        v3 = timed(lambda: L.gemm_tn_fp8_acc(dy8, gact8, dW2, alpha_dev=sdy[1:]))
        v4 = timed(lambda: L.gemm_tn_fp8_acc(dh8, ln8, dW1, alpha_dev=sdy[1:], colsum=db))
        print(f"    weight-gradient kernel {tag}: dW2 {v3:7.1f}  dW1 {v4:7.1f} us", flush=True)
    for k in ("MMG_TN8_WIDE", "MMG_TN8_XCD"):
        os.environ.pop(k, None)
