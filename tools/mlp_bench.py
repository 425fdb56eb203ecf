"""Fused CNBlock MLP forward vs the unfused LN + GEMM + GEMM path (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mmg-clip_amd"))
import torch
from mmgclip import kernels as K, linalg as L

dev = torch.device("cuda")
for C, px, n in ((96, 256 * 256, 64), (192, 128 * 128, 64), (384, 64 * 64, 64), (512, 64 * 64, 32)):
    M = px * n
    g = torch.Generator().manual_seed(0)
    xd = torch.randn(M // 64, C, generator=g).to(torch.bfloat16).repeat(64, 1).to(dev)
    res = torch.randn_like(xd)
    lnw, lnb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    w1, b1 = (torch.randn(4 * C, C, generator=g) / C ** 0.5).to(dev), torch.zeros(4 * C, device=dev)
    w2, b2 = (torch.randn(C, 4 * C, generator=g) / (4 * C) ** 0.5).to(dev), torch.zeros(C, device=dev)
    gamma = torch.ones(C, device=dev)
    packed = K.cnblock_pack(w1, w2)
    w1b, w2b = K.cast_bf16(w1), K.cast_bf16(w2)

    def fused(hp):
        return K.cnblock_mlp_fwd(xd, lnw, lnb, 1e-6, packed, b1, b2, gamma, res, want_hpre=hp, want_stats=hp)[0]

    def unfused():
        ln, _, _ = K.layernorm_fwd(xd, lnw, lnb, 1e-6, want_stats=True)
        hpre = torch.empty(M, 4 * C, device=dev, dtype=torch.bfloat16)
        gg = L.gemm_nt(ln, w1b, bias=b1, epi=L.EPI_GELU, aux_out=hpre)
        return L.gemm_nt(gg, w2b, bias=b2, colscale=gamma, residual=res)

    for name, fn in (("fused", lambda: fused(False)), ("fused+hpre", lambda: fused(True)), ("unfused", unfused), ("fused", lambda: fused(False))):
        fn(); fn(); fn(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 20
        fl = 2 * 2 * M * C * 4 * C
        print(f"C={C} M={M} {name:11s} {dt*1e3:8.3f} ms  {fl/dt/1e12:7.1f} TFLOP/s  {dt/n*1e6:7.1f} us/image  (3C traffic {3*M*C*2/dt/1e9:6.0f} GB/s)", flush=True)
    a, b = fused(False), unfused()
    print("   max |fused - unfused| =", float((a.float() - b.float()).abs().max()))
