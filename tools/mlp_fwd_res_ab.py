#!/usr/bin/env python3
"""Stage-1 fused CNBlock forward (C = 96) at C2 size: the streaming form (weights re-streamed into LDS for every 128 rows, one
barrier per chunk) against the resident form (round 4: both matrices live in LDS for the whole launch, waves free-running) with 8
and 12 waves per workgroup; interleaved rounds in ONE process, outputs compared bit for bit (same arithmetic per row)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mmg-clip_amd")):
    sys.path.insert(0, p)
import torch                                 # noqa: E402
from mmgclip import kernels as K             # noqa: E402

dev = torch.device("cuda:0")
C = 96
M = int(os.environ.get("M", 256 * 256 * 256))
g = torch.Generator().manual_seed(0)
xd = torch.randn(M // 64, C, generator=g).to(torch.bfloat16).to(dev).repeat(64, 1)
res = torch.randn(M // 64, C, generator=g).to(torch.bfloat16).to(dev).repeat(64, 1)
lnw, lnb = (1 + 0.2 * torch.randn(C, generator=g)).to(dev), (0.1 * torch.randn(C, generator=g)).to(dev)
w1, b1 = (torch.randn(4 * C, C, generator=g) / C ** 0.5).to(dev), (0.1 * torch.randn(4 * C, generator=g)).to(dev)
w2, b2 = (torch.randn(C, 4 * C, generator=g) / (4 * C) ** 0.5).to(dev), (0.1 * torch.randn(C, generator=g)).to(dev)
gamma = (0.3 + 0.7 * torch.rand(C, generator=g)).to(dev)
packed = K.cnblock_pack(w1, w2)
MODES = [m for m in os.environ.get("MODES", "0,8,12").split(",") if m]


def run(mode):
    os.environ["MMG_MLP_FWD_RES"] = mode
    return K.cnblock_mlp_fwd(xd, lnw, lnb, 1e-6, packed, b1, b2, gamma, res, want_hpre=False, want_stats=False)[0]


def timed(fn, n=5):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        out = fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n, out


ref = run("0")
for m in MODES[1:]:
    y = run(m)
    print(f"  resident {m} waves: bit-identical to the streaming form: {bool(torch.equal(y, ref))}   max |diff| {float((y.float() - ref.float()).abs().max()):.3e}", flush=True)
# odd row counts: tails of 1 ... 31 rows in the last wave tile
for tail in (1, 17, 31, 33):
    Mt = 4096 + tail
    a = K.cnblock_mlp_fwd(xd[:Mt], lnw, lnb, 1e-6, packed, b1, b2, gamma, res[:Mt], want_hpre=False, want_stats=False)[0] if not os.environ.update(MMG_MLP_FWD_RES="0") else None
    b = K.cnblock_mlp_fwd(xd[:Mt], lnw, lnb, 1e-6, packed, b1, b2, gamma, res[:Mt], want_hpre=False, want_stats=False)[0] if not os.environ.update(MMG_MLP_FWD_RES="12") else None
    print(f"  M = {Mt}: resident == streaming: {bool(torch.equal(a, b))}", flush=True)
for r in range(3):
    line = f"round {r}:"
    for m in MODES:
        t, _ = timed(lambda: run(m))
        line += f"  {'streaming' if m == '0' else 'resident ' + m + 'w'} {t:.3f} ms"
    print(line, flush=True)
