"""Flash attention kernels on the ViT-B/16 1024^2 shape (64 images x 12 heads, S = 4097): time per launch and TFLOP/s
(forward 4 S^2 64 FLOP per head, backward 2.5x), for MMG_ATT_RB = rows-per-wave / 16."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mmg-clip_amd")]
from mmgclip import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
B, S, heads = int(os.environ.get("B", 64)), int(os.environ.get("S", 4097)), 12
Hd = heads * 64
qkv = (torch.randn(B * S, 3 * Hd, device=dev) * 0.5).bfloat16()
dctx = torch.randn(B * S, Hd, device=dev).bfloat16()
flop_f = 4.0 * S * S * 64 * heads * B


def timeit(fn, n=3):
    fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for spec in sys.argv[1:] or ["0"]:
    rb, _, rbk = spec.partition(":")            # "4:3" = 4 row blocks per wave in fwd / dQ, 3 in dK/dV
    os.environ["MMG_ATT_RB"] = rb
    os.environ.pop("MMG_ATT_RB_DKV", None)
    if rbk:
        os.environ["MMG_ATT_RB_DKV"] = rbk
    ctx, lse = K.attention_fwd(qkv, None, B, S, heads)
    tf = timeit(lambda: K.attention_fwd(qkv, None, B, S, heads))
    tb = timeit(lambda: K.attention_bwd(qkv, None, ctx, lse, dctx, B, S, heads))
    print(f"RB={spec}: fwd {tf:7.2f} ms {flop_f / tf / 1e9:7.1f} TFLOP/s   bwd {tb:7.2f} ms {2.5 * flop_f / tb / 1e9:7.1f} TFLOP/s", flush=True)
