"""Run the on-chip weight-gradient backward (and the path it replaces) a few times, for rocprofv3 passes."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mmg-clip_amd"))
import torch                                   # noqa: E402
from mmgclip import kernels as K, linalg as L  # noqa: E402
dev = torch.device("cuda")
C, M = 96, int(os.environ.get("M", 16 * 256 * 256))
g = torch.Generator().manual_seed(0)
xd = torch.randn(M // 16, C, generator=g).to(torch.bfloat16).repeat(16, 1).to(dev)
dy = (0.5 * torch.randn(M // 16, C, generator=g)).to(torch.bfloat16).repeat(16, 1).to(dev)
lnw, lnb = (1 + 0.2 * torch.randn(C, generator=g)).to(dev), (0.1 * torch.randn(C, generator=g)).to(dev)
w1, b1 = (torch.randn(4 * C, C, generator=g) / C ** 0.5).to(dev), (0.1 * torch.randn(4 * C, generator=g)).to(dev)
w2 = (torch.randn(C, 4 * C, generator=g) / (4 * C) ** 0.5).to(dev)
ls = (0.3 + 0.7 * torch.rand(C, generator=g)).to(dev)
z = lambda *s: torch.zeros(*s, device=dev)   # noqa: E731
packed_new, b1f = K.cnblock_bwdw_pack(w1, w2, lnw, lnb, ls, b1)
for _ in range(3):
    K.cnblock_bwdw(dy, xd, lnw, lnb, 1e-6, packed_new, b1f, z(4 * C, C), z(4 * C), z(C, 4 * C), z(C), z(C), z(C))
if os.environ.get("BWDW_OLD", "1") == "1":
    packed_old = K.cnblock_pack(w1, w2, ls, backward=1)
    for _ in range(3):
        ldw, ldb = z(C), z(C)
        dh, gg, xln, dd, mean, rstd = K.cnblock_mlp_bwd(dy, xd, lnw, lnb, 1e-6, packed_old, b1, None, ln_grads=(ldw, ldb))
        L.gemm_tn_acc(dy, gg, z(C, 4 * C), colsum=z(C))
        L.gemm_tn_acc(dh, xln, z(4 * C, C), colsum=z(4 * C))
torch.cuda.synchronize()
