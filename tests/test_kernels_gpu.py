"""HBM-bound / conv / attention kernels vs plain PyTorch fp32 references of the same ops (bf16-rounded inputs)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _r(shape, dev, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev)


def _close(a, b, rtol, atol):
    np.testing.assert_allclose(a.detach().float().cpu().numpy(), b.detach().float().cpu().numpy(), rtol=rtol, atol=atol)


@pytest.mark.parametrize("M,C", [(1024, 96), (777, 192), (512, 384), (300, 768), (64, 3072), (33, 64)])
def test_layernorm_fwd_bwd(dev, M, C):
    from mmgclip import kernels as K
    x = _r((M, C), dev, 1, 2.0).to(BF)
    gamma, beta = _r((C,), dev, 2).abs() + 0.5, _r((C,), dev, 3)
    y, mean, rstd = K.layernorm_fwd(x, gamma, beta, 1e-6)
    xr = x.float().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), gr, br, 1e-6)
    _close(y, ref, 1e-2, 2e-2)
    dy = _r((M, C), dev, 4).to(BF)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dx = K.layernorm_bwd(dy, x, mean, rstd, gamma, dg, db)
    ref.backward(dy.float())
    _close(dx, xr.grad, 2e-2, 2e-2)
    _close(dg, gr.grad, 1e-3, 1e-2 * M ** 0.5 * 0.05 + 1e-3)
    _close(db, br.grad, 1e-3, 1e-3 * M ** 0.5)


@pytest.mark.parametrize("n,H,W,C", [(2, 8, 12, 96), (3, 6, 10, 192), (1, 4, 6, 384), (2, 4, 4, 64)])      # 64: the generic kernels
def test_layernorm_patchified(dev, n, H, W, C):
    """LN writing the 2x2-patchified layout == LN then unfold(2,2) in (kh,kw,c) order."""
    from mmgclip import kernels as K
    x = _r((n * H * W, C), dev, 5).to(BF)
    gamma, beta = _r((C,), dev, 6).abs() + 0.5, _r((C,), dev, 7)
    y, mean, rstd = K.layernorm_fwd(x, gamma, beta, 1e-6, patch_hw=(H, W))
    ref = F.layer_norm(x.float(), (C,), gamma, beta, 1e-6).reshape(n, H // 2, 2, W // 2, 2, C)
    ref = ref.permute(0, 1, 3, 2, 4, 5).reshape(n * (H // 2) * (W // 2), 4 * C)
    _close(y, ref, 1e-2, 2e-2)
    dyp = _r((n * (H // 2) * (W // 2), 4 * C), dev, 8).to(BF)
    dx = K.layernorm_bwd(dyp, x, mean, rstd, gamma, None, None, patch_hw=(H, W))
    xr = x.float().requires_grad_(True)
    r2 = F.layer_norm(xr, (C,), gamma, beta, 1e-6).reshape(n, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 2, 4, 5)
    r2.reshape(n * (H // 2) * (W // 2), 4 * C).backward(dyp.float())
    _close(dx, xr.grad, 2e-2, 2e-2)


def test_elementwise_and_casts(dev):
    from mmgclip import kernels as K
    x = _r((1000, 64), dev, 9, 2.0)
    xb = K.cast_bf16(x)
    assert torch.equal(xb, x.to(BF))
    assert torch.equal(K.cast_f32(xb), xb.float())
    _close(K.gelu(xb), F.gelu(xb.float()), 1e-2, 1e-2)
    w = _r((96, 384), dev, 10)
    rs = _r((96,), dev, 11)
    assert torch.equal(K.transpose_cast_bf16(w), w.t().contiguous().to(BF))
    assert torch.equal(K.transpose_cast_bf16(w, rs), (w * rs[:, None]).t().contiguous().to(BF))


def test_avgpool(dev):
    from mmgclip import kernels as K
    n, HW, C = 3, 1024, 768
    x = _r((n * HW, C), dev, 12).to(BF)
    y = K.avgpool_fwd(x, n, HW, C)
    _close(y, x.float().reshape(n, HW, C).mean(1), 1e-4, 1e-4)
    dy = _r((n, C), dev, 13)
    dx = K.avgpool_bwd(dy, n, HW, C)
    _close(dx, (dy / HW)[:, None, :].expand(n, HW, C).reshape(n * HW, C), 1e-2, 1e-6)


@pytest.mark.parametrize("cin,kp", [(1, 32), (3, 64)])       # 1 channel: the 16-byte fast path; 3 channels: the generic kernel
def test_patchify_matches_conv_unfold(dev, cin, kp):
    from mmgclip import kernels as K
    img = torch.rand(2, cin, 32, 48, generator=torch.Generator().manual_seed(14)).to(dev)
    p = K.patchify(img, 4, kp, True)
    scaled = (img * 65535.0 - 32767.5) / 32767.5
    ref = scaled.reshape(2, cin, 8, 4, 12, 4).permute(0, 2, 4, 3, 5, 1).reshape(2 * 8 * 12, 16 * cin)
    _close(p[:, :16 * cin], ref, 1e-2, 1e-2)
    assert (p[:, 16 * cin:] == 0).all()


def test_adamw_matches_torch(dev):
    from mmgclip import kernels as K
    p0 = _r((5000,), dev, 15)
    p = p0.clone()
    tp = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([tp], lr=5e-5, weight_decay=1e-4)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    p16 = torch.empty(5000, device=dev, dtype=BF)
    for step in range(1, 4):
        g = _r((5000,), dev, 100 + step)
        tp.grad = g.clone()
        opt.step()
        K.adamw_step(p, g, m, v, p16, 5e-5, 0.9, 0.999, 1e-8, 1e-4, step)
    _close(p, tp.data, 1e-6, 1e-7)
    assert torch.equal(p16, p.to(BF))


@pytest.mark.parametrize("n,H,W,C", [(2, 32, 32, 96), (1, 56, 56, 96), (2, 14, 14, 384), (3, 7, 7, 768), (1, 64, 40, 192), (2, 37, 50, 64)])
def test_dwconv7(dev, n, H, W, C, monkeypatch):
    from mmgclip import kernels as K
    monkeypatch.setenv("MMG_DWCONV_MFMA", "0")      # the fp32-tap VALU kernels (the matrix-core kernel has its own test below)
    x = _r((n, H, W, C), dev, 16).to(BF)
    w = _r((C, 1, 7, 7), dev, 17, 0.1)
    b = _r((C,), dev, 18)
    w49 = w.reshape(C, 49).t().contiguous()
    y = K.dwconv7(x.reshape(-1, C), w49, b, n, H, W, C)
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, br, padding=3, groups=C)
    _close(y.reshape(n, H, W, C), ref.permute(0, 2, 3, 1), 1e-2, 2e-2)
    dy = _r((n, H, W, C), dev, 19).to(BF)
    ref.backward(dy.float().permute(0, 3, 1, 2))
    res = _r((n * H * W, C), dev, 20).to(BF)
    dx = K.dwconv7(dy.reshape(-1, C), w49, None, n, H, W, C, add=res, flip=True)
    _close(dx.reshape(n, H, W, C), xr.grad.permute(0, 2, 3, 1) + res.float().reshape(n, H, W, C), 1e-2, 3e-2)
    # the other row blocking (one / two output rows per lane) accumulates every output in the same order: bit-identical
    monkeypatch.setenv("MMG_DWCONV_ROWS2", "0" if os.environ.get("MMG_DWCONV_ROWS2", "1") != "0" else "1")
    assert torch.equal(K.dwconv7(x.reshape(-1, C), w49, b, n, H, W, C), y)
    assert torch.equal(K.dwconv7(dy.reshape(-1, C), w49, None, n, H, W, C, add=res, flip=True), dx)
    monkeypatch.undo()
    monkeypatch.setenv("MMG_DWCONV_MFMA", "0")
    monkeypatch.setenv("MMG_DWCONV_TH", "16")     # 16-row tiles (two passes over one staged tile): the same sums in the same order
    assert torch.equal(K.dwconv7(x.reshape(-1, C), w49, b, n, H, W, C), y)
    assert torch.equal(K.dwconv7(dy.reshape(-1, C), w49, None, n, H, W, C, add=res, flip=True), dx)
    monkeypatch.undo()
    monkeypatch.setenv("MMG_DWCONV_MFMA", "0")
    for rows2, th in (("1", "8"), ("1", "16"), ("0", "8")):   # two dy rows per lane on 8- and 16-row tiles (defaults by size) / one
        monkeypatch.setenv("MMG_DWCONV_ROWS2", rows2)
        monkeypatch.setenv("MMG_DWG_TH", th)
        dw, db = torch.zeros(49, C, device=dev), torch.zeros(C, device=dev)
        K.dwconv7_wgrad(x.reshape(-1, C), dy.reshape(-1, C), dw, db, n, H, W, C)
        _close(dw, wr.grad.reshape(C, 49).t(), 2e-3, 2e-3 * (n * H * W) ** 0.5)
        _close(db, br.grad, 2e-3, 2e-3 * (n * H * W) ** 0.5)


@pytest.mark.parametrize("n,H,W,C", [(2, 32, 32, 96), (1, 56, 56, 96), (2, 14, 14, 384), (3, 7, 7, 768), (1, 64, 40, 192), (2, 37, 50, 64), (5, 48, 80, 128)])
def test_dwconv7_matrix_core_kernel(dev, n, H, W, C, monkeypatch):
    """csrc/dwconv7_mfma.hip (round 4): forward and data gradient of the depthwise 7x7 convolution as Toeplitz-operand MFMAs, against fp32
    torch AND against the VALU kernels.  The taps are rounded to bf16 here (fp32 there): against torch with bf16-rounded weights the only
    errors left are the fp32 accumulation order and the output rounding; maps that are not multiples of the 16 x 16 tile, several items
    per workgroup (persistent loop, double-buffered halo tiles) and the residual-gradient operand are covered."""
    from mmgclip import kernels as K
    from tests.conftest import measured
    x = _r((n, H, W, C), dev, 16).to(BF)
    w = _r((C, 1, 7, 7), dev, 17, 0.1)
    b = _r((C,), dev, 18)
    w49 = w.reshape(C, 49).t().contiguous()
    monkeypatch.setenv("MMG_DWCONV_MFMA", "1")
    y = K.dwconv7(x.reshape(-1, C), w49, b, n, H, W, C)
    wq = w.to(BF).float()                                       # what the matrix cores multiply by
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), wq, b, padding=3, groups=C).permute(0, 2, 3, 1)
    _close(y.reshape(n, H, W, C), ref, 5e-3, 1e-2)             # bf16 output rounding only (2^-9 relative)
    ref32 = F.conv2d(x.float().permute(0, 3, 1, 2), w, b, padding=3, groups=C).permute(0, 2, 3, 1)
    _close(y.reshape(n, H, W, C), ref32, 1e-2, 2e-2)            # the bar of the VALU kernel's test
    dy = _r((n, H, W, C), dev, 19).to(BF)
    res = _r((n * H * W, C), dev, 20).to(BF)
    dx = K.dwconv7(dy.reshape(-1, C), w49, None, n, H, W, C, add=res, flip=True)
    dref = F.conv_transpose2d(dy.float().permute(0, 3, 1, 2), wq, None, padding=3, groups=C).permute(0, 2, 3, 1)
    _close(dx.reshape(n, H, W, C), dref + res.float().reshape(n, H, W, C), 1e-2, 3e-2)
    monkeypatch.setenv("MMG_DWCONV_MFMA", "0")
    yv = K.dwconv7(x.reshape(-1, C), w49, b, n, H, W, C)
    dxv = K.dwconv7(dy.reshape(-1, C), w49, None, n, H, W, C, add=res, flip=True)
    _close(y, yv, 1e-2, 2e-2)
    _close(dx, dxv, 1e-2, 3e-2)
    measured("dwconv7_mfma_vs_valu", n=n, H=H, W=W, C=C, fwd_max_abs=float((y.float() - yv.float()).abs().max()),
             fwd_vs_bf16_tap_torch=float((y.reshape(n, H, W, C).float() - ref).abs().max()))


def _attn_ref(qkv, mask, B, S, heads):
    Hd = heads * 64
    q, k, v = qkv.float().reshape(B, S, 3, heads, 64).permute(2, 0, 3, 1, 4)
    s = q @ k.transpose(-1, -2) * 0.125
    if mask is not None:
        s = s + (1.0 - mask.float())[:, None, None, :] * torch.finfo(torch.float32).min
    p = s.softmax(-1)
    return (p @ v).permute(0, 2, 1, 3).reshape(B * S, Hd)


@pytest.mark.parametrize("B,S,heads", [(3, 77, 12), (2, 32, 2), (2, 128, 4), (1, 256, 2), (5, 19, 1)])
def test_attention_fwd_bwd(dev, B, S, heads):
    from mmgclip import kernels as K
    Hd = heads * 64
    qkv = _r((B * S, 3 * Hd), dev, 21).to(BF)
    g = torch.Generator().manual_seed(22)
    lens = torch.randint(max(1, S // 3), S + 1, (B,), generator=g)
    lens[0] = S
    mask = (torch.arange(S)[None, :] < lens[:, None]).long().to(dev)
    ctx, lse = K.attention_fwd(qkv, mask, B, S, heads)
    qr = qkv.float().requires_grad_(True)
    ref = _attn_ref(qr, mask, B, S, heads)
    _close(ctx, ref, 2e-2, 2e-2)
    dctx = _r((B * S, Hd), dev, 23).to(BF)
    dqkv = K.attention_bwd(qkv, mask, ctx, lse, dctx, B, S, heads)
    ref.backward(dctx.float())
    _close(dqkv, qr.grad, 5e-2, 5e-2)


def test_attention_fwd_long(dev):
    from mmgclip import kernels as K
    B, S, heads = 1, 512, 2
    qkv = _r((B * S, 3 * heads * 64), dev, 24).to(BF)
    ctx, _ = K.attention_fwd(qkv, None, B, S, heads, want_lse=False)
    _close(ctx, _attn_ref(qkv, None, B, S, heads), 2e-2, 2e-2)


def test_bert_embeddings_and_eos_pool(dev):
    from mmgclip import kernels as K
    B, S, H, V = 4, 77, 768, 1000
    g = torch.Generator().manual_seed(25)
    ids = torch.randint(0, V, (B, S), generator=g).to(dev)
    tt = torch.randint(0, 2, (B, S), generator=g).to(dev)
    word, pos, typ = _r((V, H), dev, 26).to(BF), _r((512, H), dev, 27).to(BF), _r((2, H), dev, 28).to(BF)
    out = K.bert_embed_fwd(ids, tt, word, pos, typ, S)
    ref = word.float()[ids] + pos.float()[:S][None] + typ.float()[tt]
    _close(out, ref.reshape(B * S, H), 1e-2, 2e-2)
    gr = _r((B * S, H), dev, 29).to(BF)
    dword, dpos, dtyp = torch.zeros(V, H, device=dev), torch.zeros(512, H, device=dev), torch.zeros(2, H, device=dev)
    K.bert_embed_bwd(gr, ids, tt, dword, dpos, dtyp, B, S)
    rw = torch.zeros(V, H, device=dev).index_add_(0, ids.reshape(-1), gr.float())
    rp = gr.float().reshape(B, S, H).sum(0)
    rt = torch.zeros(2, H, device=dev).index_add_(0, tt.reshape(-1), gr.float())
    _close(dword, rw, 1e-4, 1e-4)
    _close(dpos[:S], rp, 1e-4, 1e-4)
    assert (dpos[S:] == 0).all()
    _close(dtyp, rt, 1e-4, 1e-3)
    # EOS pooling
    lens = torch.tensor([77, 8, 30, 1])
    mask = (torch.arange(S)[None, :] < lens[:, None]).long().to(dev)
    hid = _r((B * S, H), dev, 30).to(BF)
    pooled, idx = K.eos_pool_fwd(hid, mask, B, S)
    assert idx.cpu().tolist() == (lens - 1).tolist()
    _close(pooled, hid.float().reshape(B, S, H)[torch.arange(B), lens - 1], 0, 0)
    dp = _r((B, H), dev, 31)
    dh = K.eos_pool_bwd(dp, idx, B, S).float().reshape(B, S, H)
    ref = torch.zeros(B, S, H, device=dev)
    ref[torch.arange(B), lens - 1] = dp.to(BF).float()
    assert torch.equal(dh, ref)


@pytest.mark.parametrize("rb", [1, 2, 4, 43])        # 43: 4 row blocks per wave in forward / dQ, 3 in dK/dV
@pytest.mark.parametrize("B,S,heads,masked", [(2, 77, 3, True), (1, 300, 2, True), (1, 1025, 2, False), (2, 130, 1, False),
                                              (4, 700, 2, False), (2, 520, 4, True)])
def test_attention_long_fwd_bwd(dev, B, S, heads, masked, rb, monkeypatch):
    """Flash-style tiled kernels (any S) vs the fp32 reference, incl. a ragged last tile, a key-padding mask, every rows-per-wave
    variant (16 * rb query rows / keys per wave) and both workgroup orders (B * heads a multiple of 8: XCD-grouped)."""
    from mmgclip import kernels as K
    monkeypatch.setenv("MMG_ATT_RB", str(rb // 10 if rb > 9 else rb))
    if rb > 9:
        monkeypatch.setenv("MMG_ATT_RB_DKV", str(rb % 10))
    Hd = heads * 64
    qkv = _r((B * S, 3 * Hd), dev, 41).to(BF)
    mask = None
    if masked:
        lens = torch.tensor([S] + [max(1, S // 2)] * (B - 1))
        mask = (torch.arange(S)[None, :] < lens[:, None]).long().to(dev)
    ctx, lse = K.attention_fwd(qkv, mask, B, S, heads, force_long=True)
    qr = qkv.float().requires_grad_(True)
    ref = _attn_ref(qr, mask, B, S, heads)
    _close(ctx, ref, 2e-2, 2e-2)
    dctx = _r((B * S, Hd), dev, 42).to(BF)
    dqkv = K.attention_bwd(qkv, mask, ctx, lse, dctx, B, S, heads, force_long=True)
    ref.backward(dctx.float())
    _close(dqkv, qr.grad, 5e-2, 5e-2)
    if S <= 256:      # the two implementations agree with each other much more tightly than with fp32
        ctx2, lse2 = K.attention_fwd(qkv, mask, B, S, heads)
        _close(ctx, ctx2, 1e-2, 1e-2)
        _close(lse, lse2, 1e-4, 1e-4)


@pytest.mark.parametrize("C,M", [(96, 128 * 5 + 37), (192, 300), (128, 129), (256, 200), (384, 128 * 2 + 19), (512, 150)])
def test_fused_cnblock_mlp_forward_matches_unfused_reference(dev, C, M):
    """mmg_cnblock_mlp_fwd = LN -> Linear(C,4C) -> GELU -> Linear(4C,C) -> layer scale -> + residual in one launch
    (torchvision CNBlock.block[2..5]); fp32 torch of the same op on the same bf16 inputs / bf16-rounded weights."""
    from mmgclip import kernels as K
    g = torch.Generator().manual_seed(C + M)
    xd = torch.randn(M, C, generator=g).to(torch.bfloat16)
    res = torch.randn(M, C, generator=g).to(torch.bfloat16)
    lnw, lnb = 1 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    w1, b1 = torch.randn(4 * C, C, generator=g) / C ** 0.5, 0.1 * torch.randn(4 * C, generator=g)
    w2, b2 = torch.randn(C, 4 * C, generator=g) / (4 * C) ** 0.5, 0.1 * torch.randn(C, generator=g)
    gamma = 0.3 + 0.7 * torch.rand(C, generator=g)
    ln = torch.nn.functional.layer_norm(xd.float(), (C,), lnw, lnb, 1e-6).to(torch.bfloat16).float()
    h = ln @ w1.to(torch.bfloat16).float().t() + b1
    gl = torch.nn.functional.gelu(h).to(torch.bfloat16).float()
    want = res.float() + gamma * (gl @ w2.to(torch.bfloat16).float().t() + b2)
    d = lambda t: t.to(dev)   # noqa: E731
    packed = K.cnblock_pack(d(w1), d(w2))
    y, hpre, mean, rstd = K.cnblock_mlp_fwd(d(xd), d(lnw), d(lnb), 1e-6, packed, d(b1), d(b2), d(gamma), d(res),
                                            want_hpre=True, want_stats=True)
    torch.cuda.synchronize()
    assert torch.allclose(y.float().cpu(), want, atol=3e-2, rtol=2e-2), float((y.float().cpu() - want).abs().max())
    assert torch.allclose(hpre.float().cpu(), h, atol=3e-2, rtol=2e-2)
    assert torch.allclose(mean.cpu(), xd.float().mean(1), atol=1e-5)
    assert torch.allclose(rstd.cpu(), (xd.float().var(1, unbiased=False) + 1e-6).rsqrt(), rtol=1e-4)
    y2, _, _, _ = K.cnblock_mlp_fwd(d(xd), d(lnw), d(lnb), 1e-6, packed, d(b1), d(b2), d(gamma), d(res))
    assert torch.equal(y2, y)
    # optional LayerNorm output next to hpre (what a GEMM-pair backward reads instead of recomputing it): the unfused kernel's bits
    y3, hpre3, _, _, xln = K.cnblock_mlp_fwd(d(xd), d(lnw), d(lnb), 1e-6, packed, d(b1), d(b2), d(gamma), d(res),
                                             want_hpre=True, want_stats=True, want_xln=True)
    ln_dev, _, _ = K.layernorm_fwd(d(xd), d(lnw), d(lnb), 1e-6, want_stats=False)
    assert torch.equal(y3, y) and torch.equal(hpre3, hpre)
    assert torch.allclose(xln.float(), ln_dev.float(), atol=2e-2, rtol=1e-2) and (xln != ln_dev).float().mean() < 0.02
    # optional GELU(hidden) next to hpre, as the second GEMM consumed it (operand of that backward's dW2 GEMM): GELU of the fp32 hidden row,
    # i.e. within one bf16 rounding of GELU(hpre)
    outs = K.cnblock_mlp_fwd(d(xd), d(lnw), d(lnb), 1e-6, packed, d(b1), d(b2), d(gamma), d(res), want_hpre=True, want_stats=True, want_gact=True)
    assert len(outs) == 5 and torch.equal(outs[0], y) and torch.equal(outs[1], hpre)
    assert torch.allclose(outs[4].float().cpu(), F.gelu(h), atol=3e-2, rtol=2e-2)
    assert torch.allclose(outs[4].float(), F.gelu(hpre.float()), atol=2e-2, rtol=1.2e-2)


@pytest.mark.parametrize("C,M", [(96, 128 * 3 + 50), (128, 200), (192, 333), (384, 128 + 77)])
def test_fused_cnblock_mlp_backward_data_path(dev, C, M):
    """mmg_cnblock_mlp_bwd vs fp32 torch autograd of the same MLP: g, dh (operands of the weight-gradient GEMMs), LN output,
    gradient w.r.t. the LN output and the LN statistics."""
    from mmgclip import kernels as K
    g_ = torch.Generator().manual_seed(7 * C + M)
    xd = torch.randn(M, C, generator=g_).to(torch.bfloat16)
    dy = (0.5 * torch.randn(M, C, generator=g_)).to(torch.bfloat16)
    lnw, lnb = 1 + 0.2 * torch.randn(C, generator=g_), 0.1 * torch.randn(C, generator=g_)
    w1, b1 = torch.randn(4 * C, C, generator=g_) / C ** 0.5, 0.1 * torch.randn(4 * C, generator=g_)
    w2 = torch.randn(C, 4 * C, generator=g_) / (4 * C) ** 0.5
    gamma = 0.3 + 0.7 * torch.rand(C, generator=g_)
    w1b, w2g = w1.to(torch.bfloat16).float(), (w2 * gamma[:, None]).to(torch.bfloat16).float()
    ln = F.layer_norm(xd.float(), (C,), lnw, lnb, 1e-6).to(torch.bfloat16).float().requires_grad_(True)
    h = ln @ w1b.t() + b1
    gl = F.gelu(h)
    dG = dy.float() @ w2g                                     # gradient w.r.t. GELU output (gamma folded like the kernel)
    (dh_want,) = torch.autograd.grad(gl, h, dG, retain_graph=True)
    dln_want = dh_want.to(torch.bfloat16).float() @ w1b
    d = lambda t: t.to(dev)   # noqa: E731
    mode = K.cnblock_bwd_mode(C)
    packed = K.cnblock_pack(d(w1), d(w2), d(gamma), backward=mode)
    hsaved = d(h.detach().to(torch.bfloat16)) if mode == 2 else None       # C = 384 reads the forward's pre-activation
    dh, gg, xln, dxln, mean, rstd = K.cnblock_mlp_bwd(d(dy), d(xd), d(lnw), d(lnb), 1e-6, packed, d(b1), hsaved)
    torch.cuda.synchronize()
    _close(xln, ln.detach(), 1e-2, 1e-2)
    _close(gg, gl.detach(), 2e-2, 2e-2)
    _close(dh, dh_want, 2e-2, 2e-2)
    _close(dxln, dln_want, 2e-2, 3e-2)
    _close(mean, xd.float().mean(1), 1e-5, 1e-5)
    _close(rstd, (xd.float().var(1, unbiased=False) + 1e-6).rsqrt(), 1e-4, 1e-5)


@pytest.mark.parametrize("C,M", [(96, 128 * 2 + 61), (128, 170)])
def test_fused_cnblock_mlp_backward_with_layernorm_backward(dev, C, M):
    """ln_dw / ln_db given: the epilogue also applies the LayerNorm backward (d loss / d xd, dgamma, dbeta) — compared with
    fp32 torch autograd of LN -> Linear -> GELU from the same d(GELU output)."""
    from mmgclip import kernels as K
    g_ = torch.Generator().manual_seed(11 * C + M)
    xd = torch.randn(M, C, generator=g_).to(torch.bfloat16)
    dy = (0.5 * torch.randn(M, C, generator=g_)).to(torch.bfloat16)
    lnw, lnb = 1 + 0.2 * torch.randn(C, generator=g_), 0.1 * torch.randn(C, generator=g_)
    w1, b1 = torch.randn(4 * C, C, generator=g_) / C ** 0.5, 0.1 * torch.randn(4 * C, generator=g_)
    w2 = torch.randn(C, 4 * C, generator=g_) / (4 * C) ** 0.5
    gamma = 0.3 + 0.7 * torch.rand(C, generator=g_)
    w1b, w2g = w1.to(torch.bfloat16).float(), (w2 * gamma[:, None]).to(torch.bfloat16).float()
    x = xd.float().requires_grad_(True)
    pw, pb = lnw.clone().requires_grad_(True), lnb.clone().requires_grad_(True)
    gl = F.gelu(F.layer_norm(x, (C,), pw, pb, 1e-6) @ w1b.t() + b1)
    dx_want, dw_want, db_want = torch.autograd.grad(gl, (x, pw, pb), dy.float() @ w2g)
    d = lambda t: t.to(dev)   # noqa: E731
    packed = K.cnblock_pack(d(w1), d(w2), d(gamma), backward=1)
    dw, db = torch.full((C,), 0.5, device=dev), torch.full((C,), -0.25, device=dev)        # accumulated into
    _, _, _, dd, _, _ = K.cnblock_mlp_bwd(d(dy), d(xd), d(lnw), d(lnb), 1e-6, packed, d(b1), None, ln_grads=(dw, db))
    torch.cuda.synchronize()
    _close(dd, dx_want, 3e-2, 3e-2)
    rel = lambda a, b: float((a.cpu() - b).norm() / b.norm())   # noqa: E731
    assert rel(dw - 0.5, dw_want) < 2e-2 and rel(db + 0.25, db_want) < 2e-2, (rel(dw - 0.5, dw_want), rel(db + 0.25, db_want))


@pytest.mark.parametrize("M", [64, 64 * 5, 64 * 300])
def test_cnblock_backward_with_on_chip_weight_gradients(dev, M):
    """mmg_cnblock_bwdw (round 3: the stage-1 block backward whose g / dh never reach HBM) vs fp32 torch autograd of
    LN -> Linear -> GELU -> Linear -> layer scale: d loss / d xd, both weight gradients in the contract of the GEMM path
    (dW2raw = dy^T g un-scaled, dW1, db1, db2raw = colsum dy) and the LayerNorm weight / bias gradients; every fp32 output is
    ACCUMULATED into what the buffer held.  M = 64 * 300 runs more tiles than the persistent grid has workgroups."""
    from mmgclip import kernels as K
    C = 96
    g_ = torch.Generator().manual_seed(13 * C + M)
    xd = (torch.randn(M, C, generator=g_) * 1.5 + 0.3).to(torch.bfloat16)
    dy = (0.5 * torch.randn(M, C, generator=g_)).to(torch.bfloat16)
    lnw, lnb = 1 + 0.2 * torch.randn(C, generator=g_), 0.1 * torch.randn(C, generator=g_)
    w1, b1 = torch.randn(4 * C, C, generator=g_) / C ** 0.5, 0.1 * torch.randn(4 * C, generator=g_)
    w2 = torch.randn(C, 4 * C, generator=g_) / (4 * C) ** 0.5
    ls = 0.3 + 0.7 * torch.rand(C, generator=g_)
    assert K.cnblock_bwdw_supported(C, M) and not K.cnblock_bwdw_supported(C, M + 1) and not K.cnblock_bwdw_supported(192, M)
    # fp64 reference of the same maths (no bf16 roundings inside: the kernel's are what the tolerances below allow for)
    x = xd.double().requires_grad_(True)
    pw, pb = lnw.double().requires_grad_(True), lnb.double().requires_grad_(True)
    W1, B1 = w1.double().requires_grad_(True), b1.double().requires_grad_(True)
    h = F.layer_norm(x, (C,), pw, pb, 1e-6) @ W1.t() + B1
    gl = F.gelu(h)
    dG = dy.double() @ (w2.double() * ls.double()[:, None])           # gradient w.r.t. the GELU output (layer scale folded)
    dx_want, dlw_want, dlb_want, dW1_want, db1_want = torch.autograd.grad(gl, (x, pw, pb, W1, B1), dG)
    dW2raw_want = dy.double().t() @ gl.detach()
    db2raw_want = dy.double().sum(0)
    d = lambda t: t.to(dev)   # noqa: E731
    packed, b1f = K.cnblock_bwdw_pack(d(w1), d(w2), d(lnw), d(lnb), d(ls), d(b1))
    assert float((b1f.cpu().double() - (b1.double() + w1.double() @ lnb.double())).abs().max()) < 1e-5
    bufs = {k: torch.full(shape, v, device=dev) for k, shape, v in (("dW1", (4 * C, C), 0.25), ("db1", (4 * C,), -1.0), ("dW2raw", (C, 4 * C), 0.5),
                                                                    ("db2raw", (C,), 2.0), ("ln_dw", (C,), 0.5), ("ln_db", (C,), -0.25))}
    dd = K.cnblock_bwdw(d(dy), d(xd), d(lnw), d(lnb), 1e-6, packed, b1f, bufs["dW1"], bufs["db1"], bufs["dW2raw"], bufs["db2raw"],
                        bufs["ln_dw"], bufs["ln_db"])
    torch.cuda.synchronize()
    rel = lambda a, b: float((a.detach().cpu().double() - b).norm() / b.norm())   # noqa: E731
    errs = {"dd": rel(dd.float(), dx_want), "dW1": rel(bufs["dW1"] - 0.25, dW1_want), "db1": rel(bufs["db1"] + 1.0, db1_want),
            "dW2raw": rel(bufs["dW2raw"] - 0.5, dW2raw_want), "db2raw": rel(bufs["db2raw"] - 2.0, db2raw_want),
            "ln_dw": rel(bufs["ln_dw"] - 0.5, dlw_want), "ln_db": rel(bufs["ln_db"] + 0.25, dlb_want)}
    from tests.conftest import measured
    measured("cnblock_bwdw", M=M, **errs)
    # bf16 operands (8 significant bits) through three products: the same bars as the GEMM-path tests above (2e-2 ... 3e-2)
    assert errs["dd"] < 2e-2 and errs["dW1"] < 2e-2 and errs["dW2raw"] < 2e-2, errs
    assert errs["db1"] < 2e-2 and errs["db2raw"] < 1e-3 and errs["ln_dw"] < 2e-2 and errs["ln_db"] < 2e-2, errs
    _close(dd, dx_want.float(), 3e-2, 3e-2)
    with pytest.raises(RuntimeError, match="multiple of 64"):
        K.cnblock_bwdw(d(dy)[:63], d(xd)[:63], d(lnw), d(lnb), 1e-6, packed, b1f, bufs["dW1"], bufs["db1"], bufs["dW2raw"], bufs["db2raw"],
                       bufs["ln_dw"], bufs["ln_db"])


def test_cnblock_bwdw_rows_beyond_2_31_bytes(dev):
    """the kernel addresses row tiles by a scalar byte offset (tile * 12 288) that passes 2^31 at 11.2 M rows - the C2 step runs it at
    16.8 M.  12 M rows made of 64 copies of one block of rows: every row's `dd` depends on that row alone, so the copies beyond 2^31 bytes must equal
    the first block BIT FOR BIT (a wrapped offset reads or writes somewhere else), and the weight gradients must be 64 x those of one block."""
    from mmgclip import kernels as K
    C, P, REP = 96, 64 * 2930, 64                          # 187 520 rows per block, 12 001 280 rows = 2.30 GB per tensor
    g_ = torch.Generator().manual_seed(5)
    xd1 = (torch.randn(P, C, generator=g_) * 1.5 + 0.3).to(torch.bfloat16).to(dev)
    dy1 = (0.5 * torch.randn(P, C, generator=g_)).to(torch.bfloat16).to(dev)
    lnw, lnb = (1 + 0.2 * torch.randn(C, generator=g_)).to(dev), (0.1 * torch.randn(C, generator=g_)).to(dev)
    w1, b1 = (torch.randn(4 * C, C, generator=g_) / C ** 0.5).to(dev), (0.1 * torch.randn(4 * C, generator=g_)).to(dev)
    w2 = (torch.randn(C, 4 * C, generator=g_) / (4 * C) ** 0.5).to(dev)
    ls = (0.3 + 0.7 * torch.rand(C, generator=g_)).to(dev)
    packed, b1f = K.cnblock_bwdw_pack(w1, w2, lnw, lnb, ls, b1)
    z = lambda *s: torch.zeros(*s, device=dev)   # noqa: E731

    def run(xd, dy):
        bufs = [z(4 * C, C), z(4 * C), z(C, 4 * C), z(C), z(C), z(C)]
        dd = K.cnblock_bwdw(dy, xd, lnw, lnb, 1e-6, packed, b1f, *bufs)
        torch.cuda.synchronize()
        return dd, bufs
    dd1, g1 = run(xd1, dy1)
    xd, dy = xd1.repeat(REP, 1).contiguous(), dy1.repeat(REP, 1).contiguous()
    assert xd.numel() * 2 > 2 ** 31 and K.cnblock_bwdw_supported(C, xd.shape[0])
    dd, gN = run(xd, dy)
    assert torch.equal(dd.view(REP, P, C), dd1.unsqueeze(0).expand(REP, P, C))
    for a, b in zip(gN, g1):
        assert float((a - REP * b).norm() / (REP * b).norm()) < 1e-4        # fp32 sums of 64 x as many terms, in another order


# ---- fp8 (e4m3) operand producers --------------------------------------------------------------------------------------------
def test_quantize_e4m3_matches_the_oracle_bytes(dev):
    """scale is a power of two, so src * scale is exact: bytes and scales must equal the oracle's bit for bit."""
    from mmgclip import kernels as K
    from oracle.encoders_oracle import q_e4m3_weight
    for seed, std, shape in ((0, 0.02, (512, 128)), (1, 3.0, (96, 384)), (2, 1e-4, (64, 64))):
        w = torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * std
        q, scales = K.quantize_e4m3(w.to(dev))
        s, inv = scales.cpu().tolist()
        assert s * inv == 1.0 and math.log2(s) == round(math.log2(s))
        amax = float(w.abs().max())
        assert 224.0 < amax * s <= 448.0
        deq = q.cpu().view(torch.float8_e4m3fn).float() * inv
        assert torch.equal(deq, q_e4m3_weight(w))
    z, scales = K.quantize_e4m3(torch.zeros(8, 8, device=dev))
    assert scales.cpu().tolist() == [1.0, 1.0] and int(z.max()) == 0


def test_layernorm_fwd_fp8_bytes(dev):
    from mmgclip import kernels as K
    from oracle.encoders_oracle import q_e4m3
    for M, C in ((1000, 128), (520, 512), (77, 1024)):
        g = torch.Generator().manual_seed(C)
        x = (torch.randn(M, C, generator=g) * 3 + 1).bfloat16()
        w, b = 1 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
        y8, mean, rstd = K.layernorm_fwd_fp8(x.to(dev), w.to(dev), b.to(dev), 1e-6)
        ref32 = torch.nn.functional.layer_norm(x.float(), (C,), w, b, 1e-6)
        ref = q_e4m3(ref32)
        y = y8.cpu().view(torch.float8_e4m3fn).float()
        mism = y != ref
        assert float(mism.float().mean()) < 5e-3                      # boundary cases only (fp32 LN evaluated in another order)
        assert bool(((y - ref).abs() <= torch.maximum(0.126 * torch.maximum(y.abs(), ref.abs()), torch.tensor(2.0 ** -9))).all())
        assert float((mean.cpu() - x.float().mean(1)).abs().max()) < 1e-4


@pytest.mark.parametrize("M,C", [(600, 512), (4200, 256), (300, 1024)])
def test_cnblock_fp8_forward_one_block(dev, M, C):
    """LayerNorm -> e4m3, Linear(e4m3 x e4m3) + GELU -> e4m3, Linear + layer scale + residual on identical bf16 inputs: with the
    same values at every rounding point the device must reproduce the oracle's e4m3 arithmetic to bf16 storage accuracy, far
    below the distance between the e4m3 and the fp32 block."""
    from mmgclip import kernels as K, linalg as L
    from oracle.encoders_oracle import q_e4m3, q_e4m3_weight
    g = torch.Generator().manual_seed(C + M)
    d = (torch.randn(M, C, generator=g) * 2 + 0.5).bfloat16()
    x = torch.randn(M, C, generator=g).bfloat16()
    lw, lb = 1 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    w1, b1 = torch.randn(4 * C, C, generator=g) * 0.05, 0.1 * torch.randn(4 * C, generator=g)
    w2, b2 = torch.randn(C, 4 * C, generator=g) * 0.05, 0.1 * torch.randn(C, generator=g)
    gamma = 0.3 + 0.7 * torch.rand(C, generator=g)
    # oracle
    ln = F.layer_norm(d.float(), (C,), lw, lb, 1e-6)
    hpre_o = F.linear(q_e4m3(ln), q_e4m3_weight(w1), b1)
    y_o = F.linear(q_e4m3(F.gelu(hpre_o)), q_e4m3_weight(w2), b2) * gamma + x.float()
    y_32 = F.linear(F.gelu(F.linear(ln, w1, b1)), w2, b2) * gamma + x.float()
    # device
    ln8, _, _ = K.layernorm_fwd_fp8(d.to(dev), lw.to(dev), lb.to(dev), 1e-6, want_stats=False)
    w18, s1 = K.quantize_e4m3(w1.to(dev))
    w28, s2 = K.quantize_e4m3(w2.to(dev))
    hpre = torch.empty(M, 4 * C, device=dev, dtype=torch.bfloat16)
    h8 = L.gemm_nt_fp8(ln8, w18, bias=b1.to(dev), epi=L.EPI_GELU, aux_out=hpre, out_kind=L.OUT_E4M3, alpha_dev=s1[1:])
    y = L.gemm_nt_fp8(h8, w28, bias=b2.to(dev), colscale=gamma.to(dev), residual=x.to(dev), alpha_dev=s2[1:])
    rel = lambda a, b: float((a.float().cpu() - b).norm() / b.norm())       # noqa: E731
    assert rel(hpre, hpre_o) < 4e-3                         # bf16 storage of the side output
    assert rel(y, y_o) < 6e-3, rel(y, y_o)                  # bf16 storage + the rare rounding-boundary element
    assert rel(y, y_32) > 4 * rel(y, y_o)                   # ... while e4m3 itself moves the block by several times that


def test_polynomial_gelu_over_every_bf16_input(dev):
    """csrc/common.h gelu_bf16 / gelu_bf16_grad (the forms used wherever the result is rounded to bf16): EVERY finite bf16 input,
    against erf-GELU in fp64.  Stated bars (round 4: the degree-6 polynomial): relative 5.8e-4 for x > 0, absolute 2.4e-4 for x < 0 (GELU), absolute 5.4e-4 (GELU');
    on top of each, the rounding of the bf16 result itself (relative 2^-8)."""
    import math
    from mmgclip import kernels as K
    bits = torch.arange(0, 65536, dtype=torch.int32)
    x = bits.to(torch.int16).view(torch.bfloat16)
    x = x[torch.isfinite(x.float())]
    pad = (-x.numel()) % 8
    x = torch.cat([x, torch.zeros(pad, dtype=torch.bfloat16)]).to(dev)
    y = K.gelu(x).float().cpu().double()
    xd = x.float().cpu().double()
    ref = 0.5 * xd * (1 + torch.erf(xd / math.sqrt(2)))
    err = (y - ref).abs()
    tiny = 1e-37                                            # (subnormal results flush)
    assert (err[xd > 0] <= (2 ** -8 + 5.8e-4) * ref[xd > 0].abs() + tiny).all()
    assert (err[xd < 0] <= 2 ** -8 * ref[xd < 0].abs() + 2.4e-4).all()
    big = xd.abs() > 8                                      # beyond the clamp: identity / (to 1e-6 |x|) zero
    assert (y[big & (xd > 0)] == xd[big & (xd > 0)]).all() and (y[big & (xd < 0)].abs() <= 1e-6 * xd[big & (xd < 0)].abs()).all()
    # derivative: mmg_act_grad_bf16 applied to a gradient of ones
    from mmgclip._hip import call, ptr, stream
    g = torch.empty_like(x)
    call("mmg_act_grad_bf16", ptr(torch.ones_like(x)), ptr(x), ptr(g), x.numel(), 0, stream())
    g = g.float().cpu().double()
    dref = 0.5 * (1 + torch.erf(xd / math.sqrt(2))) + xd * torch.exp(-0.5 * xd * xd) / math.sqrt(2 * math.pi)
    assert ((g - dref).abs() <= 2 ** -8 * dref.abs() + 5.5e-4).all()
    # the exp-free derivative polynomial of the on-chip weight-gradient backward (kind 2): the same bar; exactly 0 / 1 beyond the clamp
    call("mmg_act_grad_bf16", ptr(torch.ones_like(x)), ptr(x), ptr(g := torch.empty_like(x)), x.numel(), 2, stream())
    g = g.float().cpu().double()
    assert ((g - dref).abs() <= 2 ** -8 * dref.abs() + 5.5e-4).all()
    assert (g[xd >= 4] == 1).all() and (g[xd <= -4] == 0).all()
