"""bf16 MFMA GEMM kernels vs a plain PyTorch fp32 reference of the same op (inputs rounded to bf16 first)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(shape, dev, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev).to(torch.bfloat16)


def test_nt_integer_exact_asymmetric(dev):
    """Small-integer operands are exact in bf16/fp32: catches any fragment/layout transposition bit-exactly."""
    from mmgclip import linalg
    M, N, K = 256, 128, 64
    a = (torch.arange(M * K).reshape(M, K) % 7 - 3).float()
    b = ((torch.arange(N * K).reshape(N, K) * 3) % 5 - 2).float()
    c = linalg.gemm_nt(a.to(dev).bfloat16(), b.to(dev).bfloat16(), out_dtype=torch.float32)
    ref = a @ b.t()
    assert torch.equal(c.cpu(), ref)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 96, 96), (384, 384, 96), (512, 768, 3072), (300, 200, 160),
                                   (64, 512, 768), (1024, 2304, 768), (37, 512, 768), (4096, 192, 768), (4100, 768, 384), (4353, 512, 1536)])
def test_nt_plain(dev, M, N, K):
    from mmgclip import linalg
    a, b = _rand((M, K), dev, 1.0, 1), _rand((N, K), dev, 0.05, 2)
    ref = a.float() @ b.float().t()
    c32 = linalg.gemm_nt(a, b, out_dtype=torch.float32)
    np.testing.assert_allclose(c32.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-4 * K ** 0.5)
    c16 = linalg.gemm_nt(a, b)
    np.testing.assert_allclose(c16.float().cpu().numpy(), ref.cpu().numpy(), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("M,N,K", [(512, 384, 96), (4200, 512, 384)])      # second shape: the 256x256 tile path
def test_nt_epilogues(dev, M, N, K):
    from mmgclip import linalg
    a, b = _rand((M, K), dev, 1.0, 3), _rand((N, K), dev, 0.1, 4)
    bias = torch.randn(N, device=dev)
    cs = torch.rand(N, device=dev) + 0.5
    res = _rand((M, N), dev, 1.0, 5)
    pre = a.float() @ b.float().t() + bias
    # GELU forward with saved pre-activation
    h = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    y = linalg.gemm_nt(a, b, bias=bias, epi=linalg.EPI_GELU, aux_out=h)
    np.testing.assert_allclose(h.float().cpu().numpy(), pre.cpu().numpy(), rtol=1e-2, atol=1e-2)
    np.testing.assert_allclose(y.float().cpu().numpy(), torch.nn.functional.gelu(pre).cpu().numpy(), rtol=1e-2, atol=1e-2)
    # bias + layer scale + residual (ConvNeXt block tail)
    y = linalg.gemm_nt(a, b, bias=bias, colscale=cs, residual=res, out_dtype=torch.float32)
    np.testing.assert_allclose(y.cpu().numpy(), (pre * cs + res.float()).cpu().numpy(), rtol=1e-4, atol=1e-3)
    # GELU backward: (a b^T) * gelu'(h)
    hh = _rand((M, N), dev, 1.0, 6)
    act = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    y = linalg.gemm_nt(a, b, epi=linalg.EPI_DGELU, aux_in=hh, aux_out=act, out_dtype=torch.float32)
    np.testing.assert_allclose(act.float().cpu().numpy(), torch.nn.functional.gelu(hh.float()).cpu().numpy(), rtol=1e-2, atol=1e-2)
    x = hh.float().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    np.testing.assert_allclose(y.cpu().numpy(), ((a.float() @ b.float().t()) * x.grad).cpu().numpy(), rtol=1e-3, atol=1e-3)
    # the same with a bf16 data gradient, as the towers call it: GELU / GELU' by the exp-free polynomials (2^-11; the output rounds at 2^-9)
    act2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    y2 = linalg.gemm_nt(a, b, epi=linalg.EPI_DGELU, aux_in=hh, aux_out=act2)
    assert y2.dtype == torch.bfloat16
    np.testing.assert_allclose(y2.float().cpu().numpy(), ((a.float() @ b.float().t()) * x.grad).cpu().numpy(), rtol=1e-2, atol=1e-2)
    np.testing.assert_allclose(act2.float().cpu().numpy(), torch.nn.functional.gelu(hh.float()).cpu().numpy(), rtol=1e-2, atol=1e-2)
    # epilogue 5: GELU' alone (callers that kept GELU(h) from the forward) - bit-identical to epilogue 2's data gradient, fp32 and bf16
    y5 = linalg.gemm_nt(a, b, epi=linalg.EPI_DGELU_ONLY, aux_in=hh, out_dtype=torch.float32)
    assert torch.equal(y5, y)
    assert torch.equal(linalg.gemm_nt(a, b, epi=linalg.EPI_DGELU_ONLY, aux_in=hh), y2)
    # epilogue 6 (round 4): GELU forward whose side output is GELU'(pre-activation); epilogue 7: multiply by that saved tensor - together the
    # data gradient of epilogue 5 up to the bf16 rounding of the saved derivative (computed from the fp32 pre-activation here, from bf16(h) there)
    dsave = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    y6 = linalg.gemm_nt(a, b, bias=bias, epi=linalg.EPI_GELU_DAUX, aux_out=dsave)
    assert torch.equal(y6, linalg.gemm_nt(a, b, bias=bias, epi=linalg.EPI_GELU))
    xp = pre.clone().requires_grad_(True)
    torch.nn.functional.gelu(xp).sum().backward()
    np.testing.assert_allclose(dsave.float().cpu().numpy(), xp.grad.cpu().numpy(), rtol=2 ** -8, atol=6e-4)
    y7 = linalg.gemm_nt(a, b, epi=linalg.EPI_MUL_AUX, aux_in=dsave, out_dtype=torch.float32)
    np.testing.assert_allclose(y7.cpu().numpy(), ((a.float() @ b.float().t()) * dsave.float()).cpu().numpy(), rtol=1e-4, atol=1e-3)
    y7b = linalg.gemm_nt(a, b, epi=linalg.EPI_MUL_AUX, aux_in=dsave)
    np.testing.assert_allclose(y7b.float().cpu().numpy(), ((a.float() @ b.float().t()) * xp.grad).cpu().numpy(), rtol=1e-2, atol=1e-2)
    # ReLU pair
    y = linalg.gemm_nt(a, b, bias=bias, epi=linalg.EPI_RELU, out_dtype=torch.float32)
    np.testing.assert_allclose(y.cpu().numpy(), torch.relu(pre).cpu().numpy(), rtol=1e-4, atol=1e-3)
    y = linalg.gemm_nt(a, b, epi=linalg.EPI_DRELU, aux_in=hh, out_dtype=torch.float32)
    np.testing.assert_allclose(y.cpu().numpy(), ((a.float() @ b.float().t()) * (hh.float() > 0)).cpu().numpy(), rtol=1e-4, atol=1e-3)


def test_nt_gelu_epilogues_on_a_bert_ffn_shape_against_erf_gelu(dev):
    """every NT GEMM with a bf16 output (BERT FFN and ViT MLP included) evaluates GELU / GELU' by the clamped polynomials of
    csrc/common.h.  BERT-FFN shape (tokens x 3072 x 768) with pre-activations that SPAN [-8, 8] (one weight row per target value),
    against torch's erf GELU in fp64: absolute error of a bf16 result = polynomial (<= 5.8e-4 relative for x > 0, <= 2.4e-4 absolute
    for x < 0; exactly x / 0 beyond +-4) + bf16 rounding (2^-9 relative).  Non-finite pre-activations, documented behaviour: a NaN
    pre-activation gives a NaN activation (x * t); GELU' clamps it away (fmed3 returns a finite operand), so a NaN in the SAVED h does
    not propagate into the data gradient - the forward's own NaN output is what shows it; +-inf gives +inf / NaN (-inf * 0)."""
    from mmgclip import linalg
    from tests.conftest import measured
    M, N, K = 1155, 3072, 768                      # 15 sequences of 77 tokens
    a = torch.zeros(M, K, device=dev, dtype=torch.bfloat16)
    a[:, 0] = 1.0                                   # pre[m, n] = b[n, 0] + bias[n]: exact control of the pre-activation
    a[:, 1:] = _rand((M, K - 1), dev, 1.0, 11)
    b = torch.zeros(N, K, device=dev, dtype=torch.bfloat16)
    target = torch.linspace(-8.0, 8.0, N, device=dev)
    b[:, 0] = target.to(torch.bfloat16)
    b[:, 1:] = _rand((N, K - 1), dev, 0.02, 12)
    bias = 0.01 * torch.randn(N, device=dev)
    pre = a.double() @ b.double().t() + bias.double()
    assert float(pre.min()) < -7.5 and float(pre.max()) > 7.5
    h = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    y = linalg.gemm_nt(a, b, bias=bias, epi=linalg.EPI_GELU, aux_out=h)
    ref = torch.nn.functional.gelu(pre)
    err = (y.double() - ref).abs()
    bar = 2.0 ** -8 * ref.abs() + 6e-4               # bf16 rounding of the result + the polynomial's bound
    assert bool((err <= bar).all()), float((err - bar).max())
    # data gradient through the saved bf16 pre-activation, bf16 output (the polynomial GELU') and fp32 output (the erf form)
    dg = _rand((M, K), dev, 1.0, 13)
    w = _rand((N, K), dev, 0.05, 14)
    lin = dg.double() @ w.double().t()
    x = h.double().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    want = lin * x.grad
    y16 = linalg.gemm_nt(dg, w, epi=linalg.EPI_DGELU_ONLY, aux_in=h)
    y32 = linalg.gemm_nt(dg, w, epi=linalg.EPI_DGELU_ONLY, aux_in=h, out_dtype=torch.float32)
    e16 = float(((y16.double() - want).abs() / (2.0 ** -8 * want.abs() + 6e-4 * lin.abs() + 1e-3)).max())
    e32 = float(((y32.double() - want).abs() / (1e-5 * want.abs() + 2e-5 * lin.abs() + 1e-4)).max())
    measured("nt_gelu_bert_ffn_shape", gelu_max_abs_err=float(err.max()), gelu_max_err_over_bar=float((err / bar).max()),
             dgelu_bf16_err_over_bar=e16, dgelu_fp32_err_over_bar=e32)
    assert e16 <= 1.0 and e32 <= 1.0, (e16, e32)
    # non-finite pre-activations (documented above)
    b2 = b.clone()
    b2[5, 0] = float("nan")
    y_nan = linalg.gemm_nt(a[:256], b2, bias=bias, epi=linalg.EPI_GELU)
    assert bool(torch.isnan(y_nan[:, 5]).all()) and bool(torch.isfinite(y_nan[:, :5]).all())
    h_nan = h[:256].clone()
    h_nan[:, 7] = float("nan")
    g_nan = linalg.gemm_nt(dg[:256], w, epi=linalg.EPI_DGELU_ONLY, aux_in=h_nan)
    assert bool(torch.isfinite(g_nan).all())          # GELU' clamps the NaN away: documented, not propagated


def test_tn_integer_exact_asymmetric(dev):
    from mmgclip import linalg
    M, N1, N2 = 256, 128, 256
    a = (torch.arange(M * N1).reshape(M, N1) % 5 - 2).float()
    b = ((torch.arange(M * N2).reshape(M, N2) * 7) % 3 - 1).float()
    out = torch.zeros(N1, N2, device=dev)
    linalg.gemm_tn_acc(a.to(dev).bfloat16(), b.to(dev).bfloat16(), out)
    assert torch.equal(out.cpu(), a.t() @ b)


@pytest.mark.parametrize("M,N1,N2", [(256, 128, 128), (4096, 384, 96), (8192, 96, 384), (1000, 200, 72), (37, 512, 768), (33000, 192, 768), (32800, 512, 128),
                                     (19712, 768, 3072), (65536, 192, 768)])
def test_tn(dev, M, N1, N2):
    from mmgclip import linalg
    a, b = _rand((M, N1), dev, 1.0, 7), _rand((M, N2), dev, 1.0, 8)
    out = torch.ones(N1, N2, device=dev)           # accumulate semantics
    cs = torch.full((N1,), 2.0, device=dev)        # fused bias gradient (column sums of a), also accumulated
    linalg.gemm_tn_acc(a, b, out, colsum=cs)
    ref = 1.0 + (a.double().t() @ b.double()).float()
    np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), rtol=2e-4, atol=2e-4 * M ** 0.5)
    np.testing.assert_allclose(cs.cpu().numpy(), 2.0 + a.double().sum(0).float().cpu().numpy(), rtol=2e-4, atol=2e-4 * M ** 0.5)


def test_colsum(dev):
    from mmgclip import linalg
    for M, N in [(1000, 96), (4096, 3072), (37, 512)]:
        a = _rand((M, N), dev, 1.0, 9)
        out = torch.zeros(N, device=dev)
        linalg.colsum_acc(a, out)
        np.testing.assert_allclose(out.cpu().numpy(), a.float().sum(0).cpu().numpy(), rtol=1e-4, atol=1e-3)


# ---- fp8 (OCP e4m3) NT GEMM ------------------------------------------------------------------------------------------------
def _e4m3_bytes(shape, seed, scale=1.0):
    x = torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale
    q = x.clamp(-448, 448).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), q.to(torch.float64)


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 136, 128), (4096, 256, 512), (4200, 512, 2048), (4100, 384, 256)])
def test_nt_fp8_plain(dev, M, N, K):
    """all three tile configurations (128x128, 256x128, 256x256), ragged M / N, fp32 output."""
    from mmgclip import linalg as L
    a8, a = _e4m3_bytes((M, K), 1)
    b8, b = _e4m3_bytes((N, K), 2)
    ref = a @ b.T
    out = L.gemm_nt_fp8(a8.to(dev), b8.to(dev), out_kind=L.OUT_F32).cpu().double()
    err = float((out - ref).abs().max() / ref.abs().max())
    # the K = 128 MFMA aligns the 128 products of a step to their largest exponent before adding (measured 2e-5 of the output
    # range on random data; exact on the integer test below), so this is not an fp32 dot product to the last bit
    assert err < 1e-4, err


def test_nt_fp8_integer_exact_asymmetric(dev):
    """small integers are exact in e4m3 and in the fp32 accumulator: a swapped / permuted operand map cannot hide."""
    from mmgclip import linalg as L
    g = torch.Generator().manual_seed(3)
    a = torch.randint(-4, 5, (320, 256), generator=g).float()
    b = torch.randint(-4, 5, (264, 256), generator=g).float()
    b[:, :128] *= 2                                          # the two 64-byte halves of a k-step weigh differently
    out = L.gemm_nt_fp8(a.to(torch.float8_e4m3fn).view(torch.uint8).to(dev), b.to(torch.float8_e4m3fn).view(torch.uint8).to(dev),
                        out_kind=L.OUT_F32).cpu()
    assert torch.equal(out, a @ b.T)


@pytest.mark.parametrize("M,N,K", [(512, 384, 128), (4200, 512, 384)])
def test_nt_fp8_epilogues(dev, M, N, K):
    """bias + GELU with the bf16 pre-activation side output and an e4m3 result; then scale (device alpha) + layer scale + residual."""
    from mmgclip import linalg as L
    from oracle.encoders_oracle import q_e4m3
    a8, a = _e4m3_bytes((M, K), 4)
    b8, b = _e4m3_bytes((N, K), 5)
    g = torch.Generator().manual_seed(6)
    bias = torch.randn(N, generator=g)
    cs = torch.rand(N, generator=g) + 0.5
    res = torch.randn(M, N, generator=g).bfloat16()
    alpha_dev = torch.tensor([0.03125], device=dev)
    pre_ref = (a @ b.T) * (0.25 * 0.03125) + bias.double()
    aux = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    h8 = L.gemm_nt_fp8(a8.to(dev), b8.to(dev), bias=bias.to(dev), epi=L.EPI_GELU, aux_out=aux, out_kind=L.OUT_E4M3, alpha=0.25,
                       alpha_dev=alpha_dev)
    assert h8.dtype == torch.uint8
    assert float((aux.cpu().double() - pre_ref).abs().max()) < 1e-2 * float(pre_ref.abs().max())       # bf16 storage
    h_ref = q_e4m3(torch.nn.functional.gelu(pre_ref).float())
    h = h8.cpu().view(torch.float8_e4m3fn).float()
    # an e4m3 step is 6-12 % of the value: a result on a rounding boundary may land on either neighbour, never further.  The
    # polynomial GELU (relative error <= 5.8e-4, common.h; degree 6 since round 4) moves results within 2 * 5.8e-4 / 0.06 ~ 1.9 % of a boundary across it.
    mism = (h != h_ref)
    assert float(mism.float().mean()) < 2.5e-2, float(mism.float().mean())
    assert bool(((h - h_ref).abs() <= torch.maximum(0.126 * torch.maximum(h.abs(), h_ref.abs()), torch.tensor(2.0 ** -9))).all())
    out = L.gemm_nt_fp8(a8.to(dev), b8.to(dev), bias=bias.to(dev), colscale=cs.to(dev), residual=res.to(dev), alpha=0.25,
                        alpha_dev=alpha_dev)
    ref = pre_ref * cs.double() + res.double()
    assert out.dtype == torch.bfloat16
    assert float((out.cpu().double() - ref).abs().max()) < 1e-2 * float(ref.abs().max())


def test_nt_fp8_rejects_bad_k(dev):
    from mmgclip import linalg as L
    a = torch.zeros(128, 96, device=dev, dtype=torch.uint8)
    with pytest.raises(RuntimeError, match="multiple of 128"):
        L.gemm_nt_fp8(a, a)


# ---- wide-tile weight-gradient kernel (gemm_tn_wide.hip: M >= 65536, widths that fit 96/192 x 384) ---------------------------
@pytest.mark.parametrize("mirror", ["0", "1"])
def test_tn_wide_integer_exact_asymmetric(dev, mirror, monkeypatch):
    """Every configuration of the 8-wave kernel on exact small integers (fp32 sums stay exact): 192x384, 96x384 tiles, N1 > N2 shapes
    both as the exchanged-operand / transposed-flush form (default) and on the mirrored 384x192 / 384x96 instantiations
    (MMG_TN_WIDE_MIRROR=1), several tiles per side, asymmetric operands (a transposed or mis-swizzled fragment cannot pass)."""
    from mmgclip import linalg
    monkeypatch.setenv("MMG_TN_WIDE_MIRROR", mirror)
    for M, N1, N2 in ((65536, 192, 384), (65536 + 32, 384, 192), (70000, 96, 384), (65568, 384, 96), (66000, 384, 768), (65536, 768, 384),
                      (65536, 128, 512), (65600, 512, 128), (66000, 256, 1024), (65536 + 64, 1024, 256), (65536, 512, 2048)):   # ConvNeXt-B widths
        g = torch.Generator().manual_seed(M + N1)
        a = torch.randint(-2, 3, (M, N1), generator=g).float()
        b = torch.randint(-1, 2, (M, N2), generator=g).float()
        a[:, 0] = 1.0                                         # column sums that do not cancel
        out = torch.zeros(N1, N2, device=dev)
        cs = torch.zeros(N1, device=dev)
        linalg.gemm_tn_acc(a.to(dev).bfloat16(), b.to(dev).bfloat16(), out, colsum=cs)
        ref = (a.double().t() @ b.double()).float()
        assert torch.equal(out.cpu(), ref), (M, N1, N2, float((out.cpu() - ref).abs().max()))
        assert torch.equal(cs.cpu(), a.sum(0)), (M, N1, N2)


@pytest.mark.parametrize("M,N1,N2", [(65536, 96, 384), (131072 + 17, 384, 96), (70001, 192, 768), (65536 * 3, 768, 192), (262144, 384, 1536),
                                     (100000, 768, 3072), (65536, 128, 512), (65599, 200, 392), (80000, 1024, 256), (70003, 512, 128),
                                     (65537, 256, 1024), (66001, 2048, 512)])
def test_tn_wide(dev, M, N1, N2, monkeypatch):
    """Random data incl. ragged M (last stage partly beyond the matrix), widths that need clamped panels (128, 200, 392) and
    shapes the dispatcher must leave to the 128-wide kernel; both kernels must agree with fp64."""
    from mmgclip import linalg
    a, b = _rand((M, N1), dev, 1.0, 7), _rand((M, N2), dev, 1.0, 8)
    ref = 1.0 + (a.double().t() @ b.double()).float()
    refc = 2.0 + a.double().sum(0).float()
    for wide in ("1", "0"):
        monkeypatch.setenv("MMG_TN_WIDE8", wide)
        out = torch.ones(N1, N2, device=dev)
        cs = torch.full((N1,), 2.0, device=dev)
        linalg.gemm_tn_acc(a, b, out, colsum=cs)
        np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), rtol=2e-4, atol=2e-4 * M ** 0.5)
        np.testing.assert_allclose(cs.cpu().numpy(), refc.cpu().numpy(), rtol=2e-4, atol=2e-4 * M ** 0.5)
    # strided operands (a slice of a wider buffer), as the towers pass them
    big = _rand((M, N2 + 64), dev, 1.0, 9)
    out = torch.zeros(N1, N2, device=dev)
    monkeypatch.setenv("MMG_TN_WIDE8", "1")
    linalg.gemm_tn_acc(a, big[:, 32:32 + N2], out)
    np.testing.assert_allclose(out.cpu().numpy(), (a.double().t() @ big[:, 32:32 + N2].double()).float().cpu().numpy(), rtol=2e-4, atol=2e-4 * M ** 0.5)


# ---- fp8 backward (round 4): e5m2 gradients, 8-bit weight-gradient GEMM ---------------------------------------------------------------------
def _q8(x, dtype):
    return x.to(dtype).view(torch.uint8)


def test_quantize_e5m2_matches_torch(dev):
    """mmg_quantize_e5m2_bf16: per-tensor power-of-two scale from the device-side absmax, bytes equal to torch's float8_e5m2 cast of x * scale."""
    from mmgclip import kernels as K
    for amp in (1e-4, 3.0, 2.5e4):
        x = (_rand((4096, 96), dev, amp, 21)).contiguous()
        q, sc = K.quantize_e5m2(x)
        amax = float(x.float().abs().max())
        scale = float(sc[0])
        assert scale == 2.0 ** np.floor(np.log2(16384.0 / amax)) and abs(float(sc[1]) * scale - 1.0) < 1e-6
        ref = (x.float() * scale).to(torch.float8_e5m2).view(torch.uint8)
        assert torch.equal(q, ref)
    z = torch.zeros(64, 128, device=dev, dtype=torch.bfloat16)
    q, sc = K.quantize_e5m2(z)
    assert float(sc[0]) == 1.0 and int(q.max()) == 0
    # delayed scaling: the first call of a state is the two-pass form; the second takes its scale from the first call's absmax (target 4096) in one pass
    st = {}
    x1, x2 = _rand((4096, 96), dev, 0.01, 22).contiguous(), _rand((4096, 96), dev, 0.03, 23).contiguous()
    q1, s1 = K.quantize_e5m2(x1, st)
    assert torch.equal(q1, K.quantize_e5m2(x1)[0]) and float(st["amax"]) == float(x1.float().abs().max())
    q2, s2 = K.quantize_e5m2(x2, st)
    scale2 = 2.0 ** np.floor(np.log2(4096.0 / float(x1.float().abs().max())))
    assert float(s2[0]) == scale2 and float(st["amax"]) == float(x2.float().abs().max())
    assert torch.equal(q2, (x2.float() * scale2).to(torch.float8_e5m2).view(torch.uint8))


@pytest.mark.parametrize("M,C", [(4096, 512), (777, 256), (1, 1024), (5001, 2048), (300, 96)])
def test_quantize_e5m2_with_column_sums_in_one_pass(dev, M, C, monkeypatch):
    """mmg_quantize_e5m2_colsum_bf16 (delayed cast + bias-gradient column sums in one read of the gradient): the same bytes, scale and recorded
    absmax as the separate delayed cast, column sums ADDED to what the buffer held; widths the fused kernel does not take (96) fall back to the two
    kernels inside the same host call."""
    from mmgclip import kernels as K
    x0, x1 = _rand((M, C), dev, 0.02, 31).contiguous(), _rand((M, C), dev, 0.05, 32).contiguous()
    out = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("MMG_FP8_FUSED_CAST", fused)
        st = {}
        cs = torch.full((C,), 2.0, device=dev)
        K.quantize_e5m2(x0, st, colsum=cs)                       # first use of a state: two-pass cast + separate column sums
        q, sc = K.quantize_e5m2(x1, st, colsum=cs)               # delayed form
        out[fused] = (q, sc.clone(), float(st["amax"]), cs)
    assert torch.equal(out["1"][0], out["0"][0]) and torch.equal(out["1"][1], out["0"][1]) and out["1"][2] == out["0"][2]
    assert out["1"][2] == float(x1.float().abs().max())
    ref = 2.0 + x0.double().sum(0) + x1.double().sum(0)
    for fused in ("1", "0"):
        err = float((out[fused][3].double() - ref).abs().max())
        assert err < 1e-4 * (1.0 + float(ref.abs().max())), (fused, err)      # fp32 sums, different orders


@pytest.mark.parametrize("M,N,K", [(4096, 2048, 512), (8192, 512, 2048), (640, 384, 128), (5000, 1024, 1024)])
def test_nt_fp8_bwd_e5m2(dev, M, N, K):
    """mmg_gemm_nt_fp8_bwd: e5m2 x e4m3 products are exact in fp32 up to the accumulation; epilogues 0 / 5 / 7; bf16 / fp32 / e5m2 outputs."""
    from mmgclip import linalg as L
    g = torch.Generator().manual_seed(5)
    a8 = _q8(torch.randn(M, K, generator=g) * 3.0, torch.float8_e5m2).to(dev)
    b8 = _q8(torch.randn(N, K, generator=g) * 0.5, torch.float8_e4m3fn).to(dev)
    a = a8.view(torch.float8_e5m2).float().double()
    b = b8.view(torch.float8_e4m3fn).float().double()
    sa, sb = torch.tensor([0.25], device=dev), torch.tensor([0.5], device=dev)
    ref = (a @ b.t()) * (2.0 * 0.25 * 0.5)
    y = L.gemm_nt_fp8_bwd(a8, b8, out_kind=L.OUT_F32, alpha=2.0, alpha_dev=sa, alpha_dev2=sb)
    # (the K = 128 MFMA aligns the products of a k-step before adding them: 2e-5 of the output range with e4m3 operands, measured 2.5e-5 with the
    #  wider-ranged e5m2 ones - a rounding property of the instruction, exact on integers: test_tn_fp8_integer_exact_asymmetric)
    assert float((y.double() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    aux = _rand((M, N), dev, 1.0, 9)
    y7 = L.gemm_nt_fp8_bwd(a8, b8, aux_in=aux, epi=L.EPI_MUL_AUX, alpha=2.0, alpha_dev=sa, alpha_dev2=sb)
    assert y7.dtype == torch.bfloat16
    r7 = ref * aux.double()
    assert float((y7.double() - r7).abs().max()) <= 2 ** -8 * float(r7.abs().max()) + 1e-6
    y5 = L.gemm_nt_fp8_bwd(a8, b8, aux_in=aux, epi=L.EPI_DGELU_ONLY, out_kind=L.OUT_F32, alpha=2.0, alpha_dev=sa, alpha_dev2=sb)
    x = aux.double().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    assert float((y5.double() - ref * x.grad).abs().max()) <= 2e-4 * float(ref.abs().max())
    # e5m2 output (the gradient handed on in 8 bits): equal to torch's cast except results on a rounding boundary (fp32 accumulation order)
    y8 = L.gemm_nt_fp8_bwd(a8, b8, aux_in=aux, epi=L.EPI_MUL_AUX, out_kind=L.OUT_E5M2, alpha=2.0, alpha_dev=sa, alpha_dev2=sb)
    assert y8.dtype == torch.uint8
    want = r7.float().clamp(-57344, 57344).to(torch.float8_e5m2)
    got = y8.view(torch.float8_e5m2)
    mism = (got.view(torch.uint8) != want.view(torch.uint8))
    assert float(mism.float().mean()) < 2e-3, float(mism.float().mean())
    # one e5m2 step (25 %) at most - on top of the accumulator's own absolute error (1e-4 of the output range: the k-step alignment above), which is
    # what a result that nearly cancels sees (measured: 2 of 245 760 elements, |ref| = 5e-4 against a range of 45)
    rng = float(ref.abs().max())
    assert bool(((got.float() - want.float()).abs() <= 0.26 * torch.maximum(got.float().abs(), want.float().abs()) + 1e-4 * rng).all())
    # e4m3 gradients too (a_e5m2 = False)
    a4 = _q8(torch.randn(M, K, generator=g), torch.float8_e4m3fn).to(dev)
    y4 = L.gemm_nt_fp8_bwd(a4, b8, a_e5m2=False, out_kind=L.OUT_F32)
    r4 = a4.view(torch.float8_e4m3fn).float().double() @ b.t()
    assert float((y4.double() - r4).abs().max()) <= 1e-4 * float(r4.abs().max())


@pytest.mark.parametrize("wide", ["0", "1"])
def test_tn_fp8_integer_exact_asymmetric(dev, monkeypatch, wide):
    """Small integers are exact in e5m2 / e4m3 and in the fp32 accumulator: any wrong byte of a transposed 8-bit fragment read shows up exactly
    (both tilings of csrc/gemm_tn_fp8.hip: 128 x 128 and 256 x 256, the second forced onto a shape with partial tiles)."""
    from mmgclip import linalg as L
    monkeypatch.setenv("MMG_TN8_WIDE", wide)
    M, N1, N2 = 384, 128, 256
    a = (torch.arange(M * N1).reshape(M, N1) % 5 - 2).float()              # {-2..2}: exact in e5m2
    b = ((torch.arange(M * N2).reshape(M, N2) * 7) % 3 - 1).float()        # {-1, 0, 1}
    out = torch.zeros(N1, N2, device=dev)
    cs = torch.zeros(N1, device=dev)
    L.gemm_tn_fp8_acc(_q8(a, torch.float8_e5m2).to(dev), _q8(b, torch.float8_e4m3fn).to(dev), out, colsum=cs)
    assert torch.equal(out.cpu(), a.t() @ b)
    assert torch.equal(cs.cpu(), a.sum(0))


@pytest.mark.parametrize("kind,M,N1,N2", [("8bit", 262144, 512, 2048), ("8bit", 65536, 1024, 4096), ("bf16", 1048576, 384, 1536), ("bf16", 4194304, 96, 384)])
def test_weight_gradient_gemms_are_integer_exact_at_full_size_under_load(dev, kind, M, N1, N2):
    """BASELINE-size reductions on small-integer operands: every product and every fp32 partial sum is exact whatever the order of the atomics, so ONE
    wrong operand byte anywhere shows up as an inequality.  These kernels read transposed fragments (inline-assembly ds_read_tr) while their LDS-DMA
    ring is in flight - the property the tiled attention kernels got wrong for a build in round 4 (tools/tn_exact_check.py sweeps more shapes)."""
    from mmgclip import linalg as L
    g = torch.Generator().manual_seed(M % 1000 + N1)
    a = torch.randint(-2, 3, (M, N1), generator=g).float().to(dev)
    b = torch.randint(-1, 2, (M, N2), generator=g).float().to(dev)
    ref, refc = a.t() @ b, a.sum(0)                          # exact: |sums| <= 2 M < 2^24
    out, cs = torch.zeros(N1, N2, device=dev), torch.zeros(N1, device=dev)
    if kind == "8bit":
        L.gemm_tn_fp8_acc(a.to(torch.float8_e5m2).view(torch.uint8), b.to(torch.float8_e4m3fn).view(torch.uint8), out, colsum=cs)
    else:
        L.gemm_tn_acc(a.to(torch.bfloat16), b.to(torch.bfloat16), out, colsum=cs)
    assert torch.equal(out, ref) and torch.equal(cs, refc)


@pytest.mark.parametrize("wide", ["auto", "0", "1"])
@pytest.mark.parametrize("M,N1,N2,e5", [(4096, 512, 2048, True), (33000, 2048, 512, True), (1000, 144, 80, True), (8192, 1024, 1024, False), (130, 128, 128, True),
                                        (20000, 384, 1536, True)])
def test_tn_fp8(dev, monkeypatch, M, N1, N2, e5, wide):
    """8-bit weight-gradient GEMM against fp64 products of the same bytes; `wide` = which tiling (auto: 256 x 256 from 256-wide outputs and 8 192 rows)."""
    from mmgclip import linalg as L
    if wide != "auto":
        monkeypatch.setenv("MMG_TN8_WIDE", wide)
    g = torch.Generator().manual_seed(6)
    fa = torch.float8_e5m2 if e5 else torch.float8_e4m3fn
    a8 = _q8(torch.randn(M, N1, generator=g) * 2.0, fa).to(dev)
    b8 = _q8(torch.randn(M, N2, generator=g), torch.float8_e4m3fn).to(dev)
    a, b = a8.view(fa).float().double(), b8.view(torch.float8_e4m3fn).float().double()
    out = torch.ones(N1, N2, device=dev)              # accumulate semantics
    cs = torch.full((N1,), 2.0, device=dev)
    sd = torch.tensor([0.125], device=dev)
    L.gemm_tn_fp8_acc(a8, b8, out, a_e5m2=e5, alpha=2.0, alpha_dev=sd, colsum=cs)
    ref = 1.0 + 0.25 * (a.t() @ b)
    np.testing.assert_allclose(out.cpu().numpy(), ref.float().cpu().numpy(), rtol=5e-4, atol=5e-4 * M ** 0.5)
    np.testing.assert_allclose(cs.cpu().numpy(), (2.0 + 0.25 * a.sum(0)).float().cpu().numpy(), rtol=5e-4, atol=5e-4 * M ** 0.5)
