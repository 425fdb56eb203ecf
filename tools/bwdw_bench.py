#!/usr/bin/env python3
"""Stage-1 CNBlock backward at C2 size (M = 256 x 256 x 256 rows, C = 96): the round-2 path (fused data-path kernel + two wide
weight-gradient GEMMs) against the on-chip weight-gradient kernels of round 3 (mmg_cnblock_bwdw), interleaved rounds in one process."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mmg-clip_amd")):
    sys.path.insert(0, p)
import torch                                 # noqa: E402
from mmgclip import kernels as K, linalg as L    # noqa: E402

dev = torch.device("cuda:0")
C = 96
M = int(os.environ.get("M", 256 * 256 * 256))
g = torch.Generator().manual_seed(0)
xd = (torch.randn(M // 64, C, generator=g)).to(torch.bfloat16).to(dev).repeat(64, 1)
dy = (0.5 * torch.randn(M // 64, C, generator=g)).to(torch.bfloat16).to(dev).repeat(64, 1)
lnw, lnb = (1 + 0.2 * torch.randn(C, generator=g)).to(dev), (0.1 * torch.randn(C, generator=g)).to(dev)
w1, b1 = (torch.randn(4 * C, C, generator=g) / C ** 0.5).to(dev), (0.1 * torch.randn(4 * C, generator=g)).to(dev)
w2 = (torch.randn(C, 4 * C, generator=g) / (4 * C) ** 0.5).to(dev)
ls = (0.3 + 0.7 * torch.rand(C, generator=g)).to(dev)
z = lambda *s: torch.zeros(*s, device=dev)   # noqa: E731

packed_old = K.cnblock_pack(w1, w2, ls, backward=1)
packed_new, b1f = K.cnblock_bwdw_pack(w1, w2, lnw, lnb, ls, b1)


def old():
    dW1, db1, dW2, db2, ldw, ldb = z(4 * C, C), z(4 * C), z(C, 4 * C), z(C), z(C), z(C)
    dh, gg, xln, dd, mean, rstd = K.cnblock_mlp_bwd(dy, xd, lnw, lnb, 1e-6, packed_old, b1, None, ln_grads=(ldw, ldb))
    L.gemm_tn_acc(dy, gg, dW2, colsum=db2)
    L.gemm_tn_acc(dh, xln, dW1, colsum=db1)
    return dd, dW1, db1, dW2, db2, ldw, ldb


def new():
    dW1, db1, dW2, db2, ldw, ldb = z(4 * C, C), z(4 * C), z(C, 4 * C), z(C), z(C), z(C)
    dd = K.cnblock_bwdw(dy, xd, lnw, lnb, 1e-6, packed_new, b1f, dW1, db1, dW2, db2, ldw, ldb)
    return dd, dW1, db1, dW2, db2, ldw, ldb


def timed(fn, n=3):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        out = fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n, out


VARS = [v for v in os.environ.get("BWDW_VARS", "0,2,2h").split(",") if v]      # launch-1 schedules (csrc/cnblock_bwdw.hip: VAR), same process


def new_var(v):        # "2h" = launch-1 schedule 2 + launch 2 in ht-outer step order (MMG_BWDW_HTO)
    os.environ["MMG_BWDW_VAR"] = v.rstrip("h")
    os.environ["MMG_BWDW_HTO"] = "1" if v.endswith("h") else "0"
    return new()


_, a = timed(old, 1)
names = ("dd", "dW1", "db1", "dW2raw", "db2raw", "ln_dw", "ln_db")
for v in VARS:
    _, b = timed(lambda: new_var(v), 1)
    for n_, x, y in zip(names, a, b):
        x, y = x.float(), y.float()
        print(f"  VAR {v}: {n_:7s} rel diff new vs old {float((x - y).norm() / x.norm()):.3e}   (|old| {float(x.norm()):.3e})", flush=True)
    del b
del a
for r in range(3):
    t_old, _ = timed(old)
    line = f"round {r}: old (mlp_bwd + 2 TN) {t_old:.3f} ms"
    for v in VARS:
        t_new, _ = timed(lambda: new_var(v))
        line += f" | VAR {v}: {t_new:.3f} ms ({t_new / t_old:.3f})"
    print(line, flush=True)
