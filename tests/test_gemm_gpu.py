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
    # ReLU pair
    y = linalg.gemm_nt(a, b, bias=bias, epi=linalg.EPI_RELU, out_dtype=torch.float32)
    np.testing.assert_allclose(y.cpu().numpy(), torch.relu(pre).cpu().numpy(), rtol=1e-4, atol=1e-3)
    y = linalg.gemm_nt(a, b, epi=linalg.EPI_DRELU, aux_in=hh, out_dtype=torch.float32)
    np.testing.assert_allclose(y.cpu().numpy(), ((a.float() @ b.float().t()) * (hh.float() > 0)).cpu().numpy(), rtol=1e-4, atol=1e-3)


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
