"""Run each fused CNBlock MLP kernel a few times (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mmg-clip_amd"))
import torch
from mmgclip import kernels as K
dev = torch.device("cuda")
for C, px, n in ((96, 256 * 256, 16), (192, 128 * 128, 16), (384, 64 * 64, 16)):
    M = px * n
    g = torch.Generator().manual_seed(0)
    xd = torch.randn(M // 16, C, generator=g).to(torch.bfloat16).repeat(16, 1).to(dev)
    res = torch.randn_like(xd)
    lnw, lnb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    w1, b1 = (torch.randn(4 * C, C, generator=g) / C ** 0.5).to(dev), torch.zeros(4 * C, device=dev)
    w2, b2 = (torch.randn(C, 4 * C, generator=g) / (4 * C) ** 0.5).to(dev), torch.zeros(C, device=dev)
    gamma = torch.ones(C, device=dev)
    packed = K.cnblock_pack(w1, w2)
    for _ in range(3):
        K.cnblock_mlp_fwd(xd, lnw, lnb, 1e-6, packed, b1, b2, gamma, res)
    mode = K.cnblock_bwd_mode(C)
    if mode:
        pb = K.cnblock_pack(w1, w2, gamma, backward=mode)
        hp = torch.randn(M, 4 * C, device=dev, dtype=torch.bfloat16) if mode == 2 else None
        for _ in range(3):
            K.cnblock_mlp_bwd(res, xd, lnw, lnb, 1e-6, pb, b1, hp)
torch.cuda.synchronize()
