"""Functional host wrappers (no autograd) over the HBM-bound / conv / attention entry points of the HIP library.

Shapes follow the device layout: activations are bf16 row-major [rows, channels] (NHWC flattened for images).
"""
import os

import torch

from . import _hip
from ._hip import call, ptr, stream
from .profile import PROFILE

BF16 = torch.bfloat16


def layernorm_fwd(x, gamma, beta, eps, patch_hw=None, want_stats=True):
    """x bf16 [M,C] -> y (bf16 [M,C] or the 2x2-patchified [M/4,4C] when patch_hw=(H,W)), mean, rstd."""
    M, C = x.shape
    if patch_hw is None:
        y = torch.empty(M, C, device=x.device, dtype=BF16)
        patch, H, W = 0, 0, 0
    else:
        H, W = patch_hw
        y = torch.empty((M // (H * W)) * (H // 2) * (W // 2), 4 * C, device=x.device, dtype=BF16)
        patch = 1
    mean = torch.empty(M, device=x.device, dtype=torch.float32) if want_stats else None
    rstd = torch.empty(M, device=x.device, dtype=torch.float32) if want_stats else None
    PROFILE.timed("layernorm_fwd_kernel", 8.0 * M * C, 4 * M * C,
                  lambda: call("mmg_layernorm_fwd", ptr(x), x.stride(0), ptr(gamma), ptr(beta), float(eps), ptr(y), y.stride(0), ptr(mean),
                               ptr(rstd), M, C, patch, H, W, stream()), f"M={M} C={C}" + (" patchified" if patch else ""))
    return y, mean, rstd


def layernorm_fwd_f32(x, gamma, beta, eps, want_stats=True, want_f32=False, res=None):
    """x fp32 [M,C] (+ res fp32, added IN PLACE into x first) -> y bf16 [M,C], yf (fp32 copy of y or None), mean, rstd
    (text tower: residual stream and pre-LayerNorm sums stay fp32)."""
    M, C = x.shape
    assert x.dtype == torch.float32 and (res is None or res.dtype == torch.float32)
    y = torch.empty(M, C, device=x.device, dtype=BF16)
    yf = torch.empty(M, C, device=x.device, dtype=torch.float32) if want_f32 else None
    mean = torch.empty(M, device=x.device, dtype=torch.float32) if want_stats else None
    rstd = torch.empty(M, device=x.device, dtype=torch.float32) if want_stats else None
    PROFILE.timed("layernorm_fwd_kernel", 8.0 * M * C, (6 + (4 if want_f32 else 0) + (8 if res is not None else 0)) * M * C,
                  lambda: call("mmg_layernorm_fwd_f32", ptr(x), x.stride(0), ptr(res), res.stride(0) if res is not None else 0, ptr(gamma),
                               ptr(beta), float(eps), ptr(y), y.stride(0), ptr(yf), C if want_f32 else 0, ptr(mean), ptr(rstd), M, C, stream()),
                  f"fp32 rows M={M} C={C}" + (" +res" if res is not None else ""))
    return y, yf, mean, rstd


def layernorm_bwd_f32(dy, x, mean, rstd, gamma, dgamma=None, dbeta=None, add=None):
    M, C = x.shape
    assert x.dtype == torch.float32
    dx = torch.empty(M, C, device=x.device, dtype=BF16)
    PROFILE.timed("layernorm_bwd_kernel", 16.0 * M * C, (8 + (2 if add is not None else 0)) * M * C,
                  lambda: call("mmg_layernorm_bwd_f32", ptr(dy), dy.stride(0), ptr(x), x.stride(0), ptr(mean), ptr(rstd), ptr(gamma), ptr(dx),
                               dx.stride(0), ptr(dgamma), ptr(dbeta), M, C, ptr(add), add.stride(0) if add is not None else 0, stream()),
                  f"fp32 rows M={M} C={C}" + (" +add" if add is not None else ""))
    return dx


def layernorm_fwd_fp8(x, gamma, beta, eps, want_stats=True):
    """x bf16 [M,C] -> y e4m3 bytes (uint8 [M,C], unscaled, saturating), mean, rstd: operand of linalg.gemm_nt_fp8."""
    M, C = x.shape
    y = torch.empty(M, C, device=x.device, dtype=torch.uint8)
    mean = torch.empty(M, device=x.device, dtype=torch.float32) if want_stats else None
    rstd = torch.empty(M, device=x.device, dtype=torch.float32) if want_stats else None
    call("mmg_layernorm_fwd_fp8", ptr(x), x.stride(0), ptr(gamma), ptr(beta), float(eps), ptr(y), y.stride(0), ptr(mean),
         ptr(rstd), M, C, stream())
    return y, mean, rstd


def quantize_e5m2(x, state=None, colsum=None):
    """bf16 tensor (numel % 8 == 0) -> (e5m2 bytes uint8 of x.shape, scales fp32 [2] = (scale, 1/scale)), per-tensor power-of-two scale from
    the tensor's own absmax, computed and consumed on the device (the gradient operand of the fp8 backward, csrc/fp8_ops.hip).
    `state` (a dict the caller keeps per tensor role): delayed scaling - the first call is the two-pass form above and records the absmax; later
    calls take the scale from the previous call's absmax in one pass and record their own.
    `colsum` (fp32 [C], x 2-D): += x.sum(0) as well - in the same pass over x where the delayed form allows it (MMG_FP8_FUSED_CAST=0: never)."""
    assert x.dtype == BF16 and x.is_contiguous()
    q = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
    scales = torch.empty(2, device=x.device, dtype=torch.float32)
    if state is not None and "amax" in state:
        nxt = state["spare"]
        C = x.shape[-1]
        if colsum is not None and x.dim() == 2 and C % 16 == 0 and C <= 4096 and 256 % (C // 16) == 0 \
                and os.environ.get("MMG_FP8_FUSED_CAST", "1") != "0":
            call("mmg_quantize_e5m2_colsum_bf16", ptr(x), x.shape[0], C, ptr(state["amax"]), ptr(nxt), ptr(q), ptr(scales), ptr(colsum), stream())
            colsum = None
        else:
            call("mmg_quantize_e5m2_bf16_delayed", ptr(x), x.numel(), ptr(state["amax"]), ptr(nxt), ptr(q), ptr(scales), stream())
        state["amax"], state["spare"] = nxt, state["amax"]
        if colsum is not None:
            call("mmg_colsum_bf16", ptr(x), x.stride(0), x.shape[0], C, ptr(colsum), stream())
        return q, scales
    if colsum is not None:
        call("mmg_colsum_bf16", ptr(x), x.stride(0), x.shape[0], x.shape[1], ptr(colsum), stream())
    amax = torch.empty(1, device=x.device, dtype=torch.float32)
    call("mmg_quantize_e5m2_bf16", ptr(x), x.numel(), ptr(amax), ptr(q), ptr(scales), stream())
    if state is not None:
        state["amax"], state["spare"] = amax, torch.zeros(1, device=x.device, dtype=torch.float32)
    return q, scales


def quantize_e4m3(w):
    """fp32 tensor -> (e4m3 bytes uint8 of w.shape, scales fp32 [2] = (scale, 1/scale)) with the per-tensor power-of-two
    scale 2^floor(log2(448 / max|w|)); amax and scale stay on the device."""
    w = w.contiguous()
    assert w.dtype == torch.float32 and w.numel() % 4 == 0
    amax = torch.zeros(1, device=w.device, dtype=torch.float32)
    q = torch.empty(w.shape, device=w.device, dtype=torch.uint8)
    scales = torch.empty(2, device=w.device, dtype=torch.float32)
    call("mmg_absmax_f32", ptr(w), w.numel(), ptr(amax), stream())
    call("mmg_quantize_e4m3_f32", ptr(w), w.numel(), ptr(amax), ptr(q), ptr(scales), stream())
    return q, scales


def layernorm_bwd(dy, x, mean, rstd, gamma, dgamma=None, dbeta=None, patch_hw=None, out=None, add=None):
    M, C = x.shape
    dx = out if out is not None else torch.empty(M, C, device=x.device, dtype=BF16)
    patch, H, W = (0, 0, 0) if patch_hw is None else (1, patch_hw[0], patch_hw[1])
    PROFILE.timed("layernorm_bwd_kernel", 16.0 * M * C, (8 if add is not None else 6) * M * C,
                  lambda: call("mmg_layernorm_bwd", ptr(dy), dy.stride(0), ptr(x), x.stride(0), ptr(mean), ptr(rstd), ptr(gamma), ptr(dx),
                               dx.stride(0), ptr(dgamma), ptr(dbeta), M, C, patch, H, W, ptr(add), add.stride(0) if add is not None else 0,
                               stream()), f"M={M} C={C}" + (" patchified" if patch else "") + (" +add" if add is not None else ""))
    return dx


def gelu(x, out=None):
    y = out if out is not None else torch.empty_like(x)
    call("mmg_gelu_fwd_bf16", ptr(x), ptr(y), x.numel(), stream())
    return y


def cast_bf16(x, out=None):
    y = out if out is not None else torch.empty(x.shape, device=x.device, dtype=BF16)
    call("mmg_cast_f32_bf16", ptr(x), ptr(y), x.numel(), stream())
    return y


def cast_f32(x, out=None):
    y = out if out is not None else torch.empty(x.shape, device=x.device, dtype=torch.float32)
    call("mmg_cast_bf16_f32", ptr(x), ptr(y), x.numel(), stream())
    return y


def transpose_cast_bf16(w, rowscale=None, out=None):
    """fp32 [R,C] -> bf16 [C,R] (optionally rows scaled first)."""
    R, C = w.shape
    y = out if out is not None else torch.empty(C, R, device=w.device, dtype=BF16)
    call("mmg_transpose_cast_bf16", ptr(w), R, C, ptr(rowscale), ptr(y), y.stride(0), stream())
    return y


def avgpool_fwd(x, n, HW, C):
    y = torch.empty(n, C, device=x.device, dtype=torch.float32)
    call("mmg_avgpool_fwd", ptr(x), ptr(y), n, HW, C, stream())
    return y


def avgpool_bwd(dy, n, HW, C, out=None):
    dx = out if out is not None else torch.empty(n * HW, C, device=dy.device, dtype=BF16)
    call("mmg_avgpool_bwd", ptr(dy), ptr(dx), n, HW, C, stream())
    return dx


def patchify(img, P, Kp, scale16):
    n, Cin, H, W = img.shape
    out = torch.empty(n * (H // P) * (W // P), Kp, device=img.device, dtype=BF16)
    call("mmg_patchify", ptr(img), ptr(out), n, Cin, H, W, P, Kp, 1 if scale16 else 0, stream())
    return out


def adamw_step(p, g, m, v, p16, lr, beta1, beta2, eps, wd, step, grad_scale=1.0):
    call("mmg_adamw_step", ptr(p), ptr(g), ptr(m), ptr(v), ptr(p16), p.numel(), float(lr), float(beta1), float(beta2),
         float(eps), float(wd), int(step), float(grad_scale), stream())


def dwconv_mfma_pays(n, H, W, C, flip):
    """Where the matrix-core kernel beats the VALU one in the same-process A/B (tools/dwm_scale.py at n = 256, tools/dwm_scale_b.py at n = 64;
    profiles/r04_dwconv_mfma_scale2.txt, _scale_b2.txt - after the kernel's loads were made unconditional and its spills removed: -25 % / -23 % at
    96 channels of 256 x 256, -15 % / -19 % at 192 of 128 x 128, -36 % / -25 % at ConvNeXt-B's 128, -5 % / -14 % at its 256): large maps with few
    channel slabs, both directions.  On the 64 x 64 / 32 x 32 maps of stages 3 - 4 (tiles of 16 x 16: the halo is 1.9 x the tile, few items per CU) the
    two are at parity or the VALU kernels win.  The rule reads the MAP size and the width only, never the image count: a tower must take the same
    kernel whatever micro-batch it is run in (taps are rounded to bf16 here, kept fp32 there)."""
    return H * W >= 96 * 96 and C <= 256


def dwconv7(x, w49, bias, n, H, W, C, add=None, flip=False, out=None):
    y = out if out is not None else torch.empty(n * H * W, C, device=x.device, dtype=BF16)
    px = n * H * W * C
    # MMG_DWCONV_MFMA (read per call: A/B runs): 1 = the Toeplitz-operand matrix-core kernel (csrc/dwconv7_mfma.hip, taps rounded to bf16),
    # 0 = the fp32 VALU kernels (csrc/dwconv7.hip), unset / auto = by shape (dwconv_mfma_pays)
    mode = os.environ.get("MMG_DWCONV_MFMA", "auto")
    if mode == "1" or (mode == "auto" and dwconv_mfma_pays(n, H, W, C, flip)):
        PROFILE.timed("dwconv7_mfma_kernel", 98.0 * px, (6 if add is not None else 4) * px,
                      lambda: call("mmg_dwconv7_nhwc_mfma", ptr(x), ptr(w49), ptr(bias), ptr(add), ptr(y), n, H, W, C, 1 if flip else 0, stream()),
                      f"n={n} {H}x{W} C={C}" + (" +add" if add is not None else ""))
        return y
    fam = "dwconv7_rows2_kernel" if os.environ.get("MMG_DWCONV_ROWS2", "1") != "0" else "dwconv7_kernel"
    PROFILE.timed(fam, 98.0 * px, (6 if add is not None else 4) * px,
                  lambda: call("mmg_dwconv7_nhwc", ptr(x), ptr(w49), ptr(bias), ptr(add), ptr(y), n, H, W, C, 1 if flip else 0, stream()),
                  f"n={n} {H}x{W} C={C}" + (" +add" if add is not None else ""))
    return y


def dwconv7_wgrad(x, dy, dw49, dbias, n, H, W, C):
    px = n * H * W * C
    fam = "dwconv7_wgrad_rows2_kernel" if os.environ.get("MMG_DWCONV_ROWS2", "1") != "0" else "dwconv7_wgrad_kernel"
    PROFILE.timed(fam, 98.0 * px, 4 * px,
                  lambda: call("mmg_dwconv7_wgrad", ptr(x), ptr(dy), ptr(dw49), ptr(dbias), n, H, W, C, stream()), f"n={n} {H}x{W} C={C}")


def cnblock_supported(C):
    return C in (96, 128, 192, 256, 384, 512)


def cnblock_pack(w1, w2, gamma=None, backward=False):
    """Packed bf16 weight image of the fused CNBlock MLP kernels (W1 fp32 [4C,C], W2 fp32 [C,4C])."""
    C = w1.shape[1]
    backward = int(backward)
    n = _hip.load().mmg_cnblock_packed_elems(C, backward)
    if n <= 0:
        raise ValueError(f"fused CNBlock MLP: C={C} is not supported")
    out = torch.empty(n, device=w1.device, dtype=BF16)
    call("mmg_cnblock_pack_weights", ptr(w1), ptr(w2), ptr(gamma), ptr(out), C, backward, stream())
    return out


def cnblock_mlp_fwd(xd, ln_w, ln_b, eps, packed, b1, b2, gamma, residual, want_hpre=False, want_stats=False, want_xln=False, want_gact=False,
                    hpre_kind=0):
    """-> y, hpre, mean, rstd (and, with want_xln, the LayerNorm output as a fifth value; with want_gact, GELU(hidden) - the activation as
    the second GEMM consumed it, bf16 [M,4C] - as the last value).  hpre_kind = 1 (with want_gact): `hpre` holds GELU'(hidden)."""
    M, C = xd.shape
    y = torch.empty_like(xd)
    hpre = torch.empty(M, 4 * C, device=xd.device, dtype=BF16) if want_hpre else None
    xln = torch.empty(M, C, device=xd.device, dtype=BF16) if (want_xln and want_hpre) else None
    gact = torch.empty(M, 4 * C, device=xd.device, dtype=BF16) if (want_gact and want_hpre) else None
    assert want_hpre == want_stats, "hpre and the LN statistics are saved together"
    mean = torch.empty(M, device=xd.device, dtype=torch.float32) if want_stats else None
    rstd = torch.empty(M, device=xd.device, dtype=torch.float32) if want_stats else None
    PROFILE.timed("cnblock_mlp_fwd_kernel", 16.0 * M * C * C,
                  (6 + (8 if want_hpre else 0) + (2 if xln is not None else 0) + (8 if gact is not None else 0)) * M * C + 16 * C * C,
                  lambda: call("mmg_cnblock_mlp_fwd", ptr(xd), ptr(ln_w), ptr(ln_b), float(eps), ptr(packed), ptr(b1), ptr(b2),
                               ptr(gamma), ptr(residual), ptr(y), ptr(hpre), ptr(xln), ptr(gact), ptr(mean), ptr(rstd),
                               int(hpre_kind) if gact is not None else 0, M, C, stream()),
                  f"M={M} C={C}" + (" +hpre" if want_hpre else "") + (" +xln" if xln is not None else "") + (" +gact" if gact is not None else ""))
    out = (y, hpre, mean, rstd) + ((xln,) if want_xln else ())
    return out + ((gact,) if want_gact else ())


def cnblock_bwd_mode(C):
    """0 = no fused backward, 1 = hidden row recomputed, 2 = needs the forward's saved pre-activation."""
    return int(_hip.load().mmg_cnblock_mlp_bwd_supported(C))


def cnblock_mlp_bwd(dy, xd, ln_w, ln_b, eps, packed_bwd, b1, hpre=None, ln_grads=None):
    """-> dh, g [M,4C]; xln, dxln [M,C]; mean, rstd [M]  (see include/mmgclip_hip.h).  With ln_grads = (dgamma, dbeta) the
    LayerNorm backward is fused: dxln is then the gradient w.r.t. xd and the two fp32 [C] buffers are accumulated."""
    M, C = xd.shape
    dev = xd.device
    dh = torch.empty(M, 4 * C, device=dev, dtype=BF16)
    g = torch.empty(M, 4 * C, device=dev, dtype=BF16)
    xln, dxln = torch.empty_like(xd), torch.empty_like(xd)
    mean = torch.empty(M, device=dev, dtype=torch.float32)
    rstd = torch.empty(M, device=dev, dtype=torch.float32)
    # algorithmic work: dG and dX GEMMs (the recomputed hidden GEMM is overhead, not counted); reads dy, xd (+h), writes dh, g, xln, dxln
    PROFILE.timed("cnblock_mlp_bwd_kernel", 16.0 * M * C * C, (24 + (8 if hpre is not None else 0)) * M * C + 24 * C * C,
                  lambda: call("mmg_cnblock_mlp_bwd", ptr(dy), ptr(xd), ptr(ln_w), ptr(ln_b), float(eps), ptr(packed_bwd), ptr(b1),
                               ptr(hpre), ptr(dh), ptr(g), ptr(xln), ptr(dxln), ptr(mean), ptr(rstd),
                               ptr(ln_grads[0]) if ln_grads else None, ptr(ln_grads[1]) if ln_grads else None, M, C, stream()),
                  f"M={M} C={C}")
    return dh, g, xln, dxln, mean, rstd


def cnblock_bwdw_supported(C, M):
    """On-chip weight-gradient backward of the CNBlock MLP (csrc/cnblock_bwdw.hip): C = 96, M a multiple of 64."""
    return bool(_hip.load().mmg_cnblock_bwdw_supported(C)) and M % 64 == 0 and M * C * 2 < 2 ** 32


def cnblock_bwdw_pack(w1, w2, ln_w, ln_b, layer_scale, b1):
    """-> (packed bf16 buffer, b1' fp32 [4C]) for cnblock_bwdw."""
    C = w1.shape[1]
    n = _hip.load().mmg_cnblock_bwdw_packed_elems(C)
    if n <= 0:
        raise ValueError(f"on-chip weight-gradient CNBlock backward: C={C} is not supported")
    packed = torch.empty(n, device=w1.device, dtype=BF16)
    b1f = torch.empty(4 * C, device=w1.device, dtype=torch.float32)
    call("mmg_cnblock_bwdw_pack", ptr(w1), ptr(w2), ptr(ln_w), ptr(ln_b), ptr(layer_scale), ptr(b1), ptr(packed), ptr(b1f), C, stream())
    return packed, b1f


def cnblock_bwdw(dy, xd, ln_w, ln_b, eps, packed, b1f, dW1, db1, dW2raw, db2raw, ln_dw, ln_db):
    """-> dd [M,C] bf16; accumulates the six fp32 gradient buffers (see include/mmgclip_hip.h)."""
    M, C = xd.shape
    dd = torch.empty_like(xd)
    # algorithmic work: dG, dW2, dW1, d LN-out products (the two recomputed h products are overhead); reads dy, xd twice, writes dd
    PROFILE.timed("cnblock_bwdw_kernel", 32.0 * M * C * C, 10 * M * C + 24 * C * C,
                  lambda: call("mmg_cnblock_bwdw", ptr(dy), ptr(xd), ptr(ln_w), ptr(ln_b), float(eps), ptr(packed), ptr(b1f), ptr(dd),
                               ptr(dW1), ptr(db1), ptr(dW2raw), ptr(db2raw), ptr(ln_dw), ptr(ln_db), M, C, stream()), f"M={M} C={C}")
    return dd


def attention_fwd(qkv, mask, B, S, heads, want_lse=True, force_long=False, cu=None):
    """Whole-sequence-in-LDS kernel up to S = 512, flash-style tiled kernel beyond (or when force_long).
    cu (int32 [B+1]): packed layout, rows cu[b]..cu[b+1] are sequence b (S = longest sequence, mask unused)."""
    Hd = heads * 64
    ctx = torch.empty(qkv.shape[0], Hd, device=qkv.device, dtype=BF16)
    lse = torch.empty(B, heads, S, device=qkv.device, dtype=torch.float32) if want_lse else None
    if cu is not None:
        call("mmg_attention_varlen_fwd", ptr(qkv), qkv.stride(0), ptr(cu), ptr(ctx), Hd, ptr(lse), B, S, heads, Hd, 0.125, stream())
        return ctx, lse
    long_path = force_long or S > 512
    name = "mmg_attention_long_fwd" if long_path else "mmg_attention_fwd"
    PROFILE.timed("attn_flash_fwd_kernel" if long_path else "attn_fwd_kernel", 4.0 * B * heads * S * S * 64, 8 * B * S * Hd,
                  lambda: call(name, ptr(qkv), qkv.stride(0), ptr(mask), ptr(ctx), Hd, ptr(lse), B, S, heads, Hd, 0.125, stream()))
    return ctx, lse


def attention_dropout_fwd(qkv, mask, B, S, heads, p, seed, site, want_lse=True, cu=None, first_sequence=0):
    """attention_fwd with HF's attention_probs_dropout (csrc/dropout.h): S <= 512; cu selects the packed layout."""
    Hd = heads * 64
    ctx = torch.empty(qkv.shape[0], Hd, device=qkv.device, dtype=BF16)
    lse = torch.empty(B, heads, S, device=qkv.device, dtype=torch.float32) if want_lse else None
    call("mmg_attention_dropout_fwd", ptr(qkv), qkv.stride(0), ptr(mask) if cu is None else None, ptr(cu), ptr(ctx), Hd, ptr(lse),
         B, S, heads, Hd, 0.125, float(p), int(seed), int(site), int(first_sequence), stream())
    return ctx, lse


def attention_dropout_bwd(qkv, mask, ctx, lse, dctx, B, S, heads, p, seed, site, cu=None, first_sequence=0):
    """Backward of attention_dropout_fwd (same p, seed, site): whole-sequence kernel up to S = 256, tiled kernels up to 512 (padded)."""
    Hd = heads * 64
    dqkv = torch.empty(qkv.shape[0], 3 * Hd, device=qkv.device, dtype=BF16)
    if S > 256:
        assert cu is None, "the packed layout is used up to S = 256 only"
        delta = torch.empty(B * heads * S, device=qkv.device, dtype=torch.float32)
        call("mmg_attention_dropout_long_bwd", ptr(qkv), qkv.stride(0), ptr(mask), ptr(ctx), ctx.stride(0), ptr(lse), ptr(dctx),
             dctx.stride(0), ptr(dqkv), dqkv.stride(0), ptr(delta), B, S, heads, Hd, 0.125, float(p), int(seed), int(site),
             int(first_sequence), stream())
        return dqkv
    call("mmg_attention_dropout_bwd", ptr(qkv), qkv.stride(0), ptr(mask) if cu is None else None, ptr(cu), ptr(ctx), ctx.stride(0),
         ptr(lse), ptr(dctx), dctx.stride(0), ptr(dqkv), dqkv.stride(0), B, S, heads, Hd, 0.125, float(p), int(seed), int(site),
         int(first_sequence), stream())
    return dqkv


def dropout_f32_(x, p, seed, site, rows=None, want_bf16=False):
    """x fp32 [M,C] <- dropout(x) in place (mask index token * C + column, token = rows[m] or m); -> bf16 copy or None."""
    M, C = x.shape
    assert x.dtype == torch.float32 and x.stride(1) == 1
    xb = torch.empty(M, C, device=x.device, dtype=BF16) if want_bf16 else None
    call("mmg_dropout_f32", ptr(x), x.stride(0), ptr(xb), C if want_bf16 else 0, ptr(rows), M, C, float(p), int(seed), int(site),
         stream())
    return xb


def dropout_bf16(x, p, seed, site, rows=None):
    """dropout(x) for a bf16 [M,C] tensor (the forward's mask applied to a gradient)."""
    M, C = x.shape
    assert x.dtype == BF16 and x.stride(1) == 1
    out = torch.empty(M, C, device=x.device, dtype=BF16)
    call("mmg_dropout_bf16", ptr(x), x.stride(0), ptr(out), C, ptr(rows), M, C, float(p), int(seed), int(site), stream())
    return out


def attention_bwd(qkv, mask, ctx, lse, dctx, B, S, heads, out=None, force_long=False, cu=None):
    Hd = heads * 64
    dqkv = out if out is not None else torch.empty(qkv.shape[0], 3 * Hd, device=qkv.device, dtype=BF16)
    if cu is not None:
        call("mmg_attention_varlen_bwd", ptr(qkv), qkv.stride(0), ptr(cu), ptr(ctx), ctx.stride(0), ptr(lse), ptr(dctx),
             dctx.stride(0), ptr(dqkv), dqkv.stride(0), B, S, heads, Hd, 0.125, stream())
        return dqkv
    if force_long or S > 256:
        delta = torch.empty(B * heads * S, device=qkv.device, dtype=torch.float32)
        # (one family for the three launches of the tiled backward: delta, dQ, dK/dV)
        PROFILE.timed("attn_flash_bwd_kernels", 10.0 * B * heads * S * S * 64, 22 * B * S * Hd,
                      lambda: call("mmg_attention_long_bwd", ptr(qkv), qkv.stride(0), ptr(mask), ptr(ctx), ctx.stride(0), ptr(lse),
                                   ptr(dctx), dctx.stride(0), ptr(dqkv), dqkv.stride(0), ptr(delta), B, S, heads, Hd, 0.125, stream()))
    else:
        call("mmg_attention_bwd", ptr(qkv), qkv.stride(0), ptr(mask), ptr(ctx), ctx.stride(0), ptr(lse), ptr(dctx),
             dctx.stride(0), ptr(dqkv), dqkv.stride(0), B, S, heads, Hd, 0.125, stream())
    return dqkv


def bert_embed_fwd(ids, type_ids, word, pos, typ, S):
    M = ids.numel()
    H = word.shape[1]
    out = torch.empty(M, H, device=word.device, dtype=BF16)
    call("mmg_bert_embed_fwd", ptr(ids), ptr(type_ids), ptr(word), ptr(pos), ptr(typ), ptr(out), M, S, H, word.shape[0],
         typ.shape[0], stream())
    return out


def bert_embed_bwd(g, ids, type_ids, dword, dpos, dtype, B, S):
    H = g.shape[1]
    call("mmg_bert_embed_bwd", ptr(g), ptr(ids), ptr(type_ids), ptr(dword), ptr(dpos), ptr(dtype), B, S, H,
         dword.shape[0], dtype.shape[0], stream())


def eos_pool_fwd(hidden, mask, B, S):
    H = hidden.shape[1]
    out = torch.empty(B, H, device=hidden.device, dtype=torch.float32)
    idx = torch.empty(B, device=hidden.device, dtype=torch.int32)
    call("mmg_eos_pool_fwd_f32" if hidden.dtype == torch.float32 else "mmg_eos_pool_fwd", ptr(hidden), ptr(mask), ptr(out), ptr(idx),
         B, S, H, stream())
    return out, idx


def eos_pool_bwd(dout, idx, B, S):
    H = dout.shape[1]
    dh = torch.empty(B * S, H, device=dout.device, dtype=BF16)
    call("mmg_eos_pool_bwd", ptr(dout), ptr(idx), ptr(dh), B, S, H, stream())
    return dh


# ---- ResNet-50 tower pieces (csrc/resnet_ops.hip) -----------------------------------------------------------------------
def conv_out_hw(H, W, k, stride, pad):
    return (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1


def im2col(x, n, H, W, C, k, stride, pad, Kp):
    Ho, Wo = conv_out_hw(H, W, k, stride, pad)
    col = torch.empty(n * Ho * Wo, Kp, device=x.device, dtype=BF16)
    call("mmg_im2col_nhwc", ptr(x), ptr(col), n, H, W, C, k, k, stride, pad, Kp, stream())
    return col


def col2im(dcol, n, H, W, C, k, stride, pad):
    dx = torch.empty(n * H * W, C, device=dcol.device, dtype=BF16)
    call("mmg_col2im_nhwc", ptr(dcol), ptr(dx), n, H, W, C, k, k, stride, pad, dcol.shape[1], stream())
    return dx


def maxpool3x3s2(x, n, H, W, C):
    Ho, Wo = conv_out_hw(H, W, 3, 2, 1)
    y = torch.empty(n * Ho * Wo, C, device=x.device, dtype=BF16)
    call("mmg_maxpool3x3s2_nhwc", ptr(x), ptr(y), n, H, W, C, stream())
    return y


def batchnorm_fwd(x, gamma, beta, running_mean, running_var, train, eps=1e-5, momentum=0.1, residual=None, relu=False):
    """nn.BatchNorm2d on rows [M, C] (+ shortcut add + ReLU in the same pass) -> y, mean, rstd."""
    M, C = x.shape
    dev = x.device
    st = torch.zeros(2, C, device=dev, dtype=torch.float32) if train else None
    if train:
        call("mmg_bn_stats", ptr(x), M, C, ptr(st[0]), ptr(st[1]), stream())
    out4 = torch.empty(4, C, device=dev, dtype=torch.float32)          # mean, rstd, scale, shift
    call("mmg_bn_finalize", ptr(st[0]) if train else None, ptr(st[1]) if train else None, M, C, ptr(gamma), ptr(beta), float(eps),
         float(momentum), ptr(running_mean), ptr(running_var), 1 if train else 0, ptr(out4[0]), ptr(out4[1]), ptr(out4[2]),
         ptr(out4[3]), stream())
    y = torch.empty_like(x)
    call("mmg_bn_apply", ptr(x), ptr(out4[2]), ptr(out4[3]), ptr(residual), ptr(y), M, C, 1 if relu else 0, stream())
    return y, out4[0], out4[1]


def batchnorm_bwd(dy, x, out, mean, rstd, gamma, dgamma, dbeta, want_dres=False):
    """Training-mode backward of y = relu?(bn(x) (+ res)); `out` = the layer's own output (ReLU mask) or None.
    Accumulates dgamma / dbeta; returns dx (and the masked gradient of the residual branch when want_dres)."""
    M, C = x.shape
    sums = torch.zeros(2, C, device=x.device, dtype=torch.float32)
    call("mmg_bn_bwd_reduce", ptr(dy), ptr(x), ptr(out), ptr(mean), ptr(rstd), M, C, ptr(sums[0]), ptr(sums[1]), stream())
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    call("mmg_bn_bwd_apply", ptr(dy), ptr(x), ptr(out), ptr(mean), ptr(rstd), ptr(gamma), ptr(sums[0]), ptr(sums[1]), M, C,
         ptr(dx), ptr(dres), ptr(dgamma), ptr(dbeta), stream())
    return dx, dres
