#!/usr/bin/env python3
"""Run-to-run and grouping reproducibility of the tiled attention forward at ViT-B/16 1024^2 size (S = 4097, 12 heads) with the chip from lightly to
fully loaded: a query row's arithmetic does not depend on rows-per-wave (MMG_ATT_RB), so every launch must give the SAME BITS.  Round 4: this is the
tool that showed workgroups reading K / V tiles whose LDS-DMA had not landed (DESIGN.md "A wait the tiled attention kernels never had")."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mmg-clip_amd")]
from mmgclip import kernels as K
dev = torch.device("cuda:0")
heads, S = 12, 4097
for B in (2, 16, 64):
    g = torch.Generator().manual_seed(B)
    qkv = (torch.randn(B * S, 3 * heads * 64, generator=g) * 0.5).to(dev).bfloat16()
    for rb in ("1", "2", "4"):
        os.environ["MMG_ATT_RB"] = rb
        for rep in range(3):
            ctx, lse = K.attention_fwd(qkv, None, B, S, heads)
            torch.cuda.synchronize()
            bad = (~torch.isfinite(ctx.float())).sum().item()
            badl = (~torch.isfinite(lse.float())).sum().item()
            if rep == 0 and rb == "1":
                ref = ctx.clone()
            d = float((ctx.float() - ref.float()).abs().max())
            print(f"B={B} RB={rb} rep={rep}: non-finite ctx {bad} lse {badl}  max|ctx - RB1| {d:.3e}", flush=True)
