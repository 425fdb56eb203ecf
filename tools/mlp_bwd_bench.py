"""Fused CNBlock MLP backward data path vs the unfused DGELU GEMM + LN recompute + dX GEMM (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mmg-clip_amd"))
import torch
from mmgclip import kernels as K, linalg as L

dev = torch.device("cuda")
for C, px, n in ((96, 256 * 256, 64), (192, 128 * 128, 64), (384, 64 * 64, 64)):
    mode = K.cnblock_bwd_mode(C)
    if not mode:
        continue
    M = px * n
    g = torch.Generator().manual_seed(0)
    xd = torch.randn(M // 64, C, generator=g).to(torch.bfloat16).repeat(64, 1).to(dev)
    dy = torch.randn_like(xd)
    lnw, lnb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    w1, b1 = (torch.randn(4 * C, C, generator=g) / C ** 0.5).to(dev), torch.zeros(4 * C, device=dev)
    w2 = (torch.randn(C, 4 * C, generator=g) / (4 * C) ** 0.5).to(dev)
    gamma = torch.ones(C, device=dev)
    packed = K.cnblock_pack(w1, w2, gamma, backward=mode)
    w1t, w2gt = K.transpose_cast_bf16(w1), K.transpose_cast_bf16(w2, gamma)
    hpre = torch.randn(M, 4 * C, device=dev, dtype=torch.bfloat16)

    def fused():
        return K.cnblock_mlp_bwd(dy, xd, lnw, lnb, 1e-6, packed, b1, hpre if mode == 2 else None)

    def unfused():
        gg = torch.empty_like(hpre)
        dh = L.gemm_nt(dy, w2gt, epi=L.EPI_DGELU, aux_in=hpre, aux_out=gg)
        ln, _, _ = K.layernorm_fwd(xd, lnw, lnb, 1e-6, want_stats=False)
        return L.gemm_nt(dh, w1t)

    for name, fn in (("fused", fused), ("unfused", unfused)):
        fn(); fn(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 10
        print(f"C={C} M={M} {name:8s} {dt*1e3:8.3f} ms  {dt/n*1e6:7.1f} us/image  (12C traffic {12*M*C*2/dt/1e9:6.0f} GB/s)", flush=True)
