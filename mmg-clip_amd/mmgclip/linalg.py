"""Thin host wrappers over the bf16 MFMA GEMM entry points (csrc/gemm_bf16.hip). No autograd here."""
import torch

from ._hip import call, ptr, stream

EPI_NONE, EPI_GELU, EPI_DGELU, EPI_RELU, EPI_DRELU, EPI_DGELU_ONLY, EPI_GELU_DAUX, EPI_MUL_AUX = 0, 1, 2, 3, 4, 5, 6, 7


def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1, "row-major 2-D tensor expected"
    return t.stride(0)


def gemm_nt(a, b, out=None, bias=None, colscale=None, residual=None, aux_in=None, aux_out=None, epi=EPI_NONE,
            out_dtype=torch.bfloat16, alpha=1.0):
    """out[M,N] = epilogue(alpha * a[M,K] @ b[N,K].T + bias) — see include/mmgclip_hip.h."""
    assert a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=out_dtype)
    call("mmg_gemm_nt_bf16", ptr(a), _ld(a), ptr(b), _ld(b), ptr(out), _ld(out), M, N, K, ptr(bias), ptr(colscale),
         ptr(residual), _ld(residual) if residual is not None else 0, ptr(aux_in),
         _ld(aux_in) if aux_in is not None else 0, ptr(aux_out), _ld(aux_out) if aux_out is not None else 0, epi,
         1 if out.dtype == torch.float32 else 0, float(alpha), stream())
    return out


OUT_BF16, OUT_F32, OUT_E4M3 = 0, 1, 2


def gemm_nt_fp8(a, b, out=None, bias=None, colscale=None, residual=None, aux_out=None, epi=EPI_NONE, out_kind=OUT_BF16,
                alpha=1.0, alpha_dev=None):
    """out[M,N] = epilogue(alpha * alpha_dev * a[M,K] @ b[N,K].T + bias), a / b e4m3 bytes (uint8) — include/mmgclip_hip.h."""
    assert a.dtype == torch.uint8 and b.dtype == torch.uint8
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=(torch.bfloat16, torch.float32, torch.uint8)[out_kind])
    call("mmg_gemm_nt_fp8", ptr(a), _ld(a), ptr(b), _ld(b), ptr(out), _ld(out), M, N, K, ptr(bias), ptr(colscale),
         ptr(residual), _ld(residual) if residual is not None else 0, ptr(aux_out),
         _ld(aux_out) if aux_out is not None else 0, epi, out_kind, float(alpha), ptr(alpha_dev), stream())
    return out


OUT_E5M2 = 3


def gemm_nt_fp8_bwd(a, b, a_e5m2=True, out=None, aux_in=None, epi=EPI_NONE, out_kind=OUT_BF16, alpha=1.0, alpha_dev=None, alpha_dev2=None):
    """out[M,N] = epilogue(alpha * alpha_dev * alpha_dev2 * a[M,K] @ b[N,K].T): a e5m2 / e4m3 bytes, b e4m3 bytes (uint8) - include/mmgclip_hip.h."""
    assert a.dtype == torch.uint8 and b.dtype == torch.uint8
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype={OUT_BF16: torch.bfloat16, OUT_F32: torch.float32, OUT_E5M2: torch.uint8}[out_kind])
    call("mmg_gemm_nt_fp8_bwd", ptr(a), _ld(a), 1 if a_e5m2 else 0, ptr(b), _ld(b), ptr(out), _ld(out), M, N, K, ptr(aux_in),
         _ld(aux_in) if aux_in is not None else 0, epi, out_kind, float(alpha), ptr(alpha_dev), ptr(alpha_dev2), stream())
    return out


def gemm_tn_fp8_acc(a, b, out, a_e5m2=True, alpha=1.0, alpha_dev=None, colsum=None):
    """out[N1,N2] (fp32) += alpha * alpha_dev * a[M,N1].T @ b[M,N2], a e5m2 / e4m3 bytes, b e4m3 bytes; colsum[N1] += scaled column sums of a."""
    assert a.dtype == torch.uint8 and b.dtype == torch.uint8 and out.dtype == torch.float32
    M, N1 = a.shape
    N2 = b.shape[1]
    assert b.shape[0] == M and out.shape == (N1, N2)
    call("mmg_gemm_tn_fp8", ptr(a), _ld(a), 1 if a_e5m2 else 0, ptr(b), _ld(b), ptr(out), _ld(out), M, N1, N2, float(alpha), ptr(alpha_dev),
         ptr(colsum), stream())
    return out


def gemm_tn_acc(a, b, out, alpha=1.0, colsum=None):
    """out[N1,N2] (fp32) += alpha * a[M,N1].T @ b[M,N2];  colsum[N1] (fp32, optional) += alpha * a.sum(0)."""
    assert a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and out.dtype == torch.float32
    M, N1 = a.shape
    N2 = b.shape[1]
    assert b.shape[0] == M and out.shape == (N1, N2)
    call("mmg_gemm_tn_bf16", ptr(a), _ld(a), ptr(b), _ld(b), ptr(out), _ld(out), M, N1, N2, float(alpha), ptr(colsum),
         stream())
    return out


def colsum_acc(a, out):
    """out[N] (fp32) += a[M,N].sum(0)."""
    M, N = a.shape
    call("mmg_colsum_bf16", ptr(a), _ld(a), M, N, ptr(out), stream())
    return out


from .profile import PROFILE  # noqa: E402  (bench.py reads linalg.PROFILE)

_gemm_nt_raw = gemm_nt
_gemm_tn_raw = gemm_tn_acc
_EPI_NAMES = {EPI_NONE: "none", EPI_GELU: "gelu", EPI_DGELU: "dgelu", EPI_RELU: "relu", EPI_DRELU: "drelu", EPI_DGELU_ONLY: "dgelu_only", EPI_GELU_DAUX: "gelu+dgelu_aux", EPI_MUL_AUX: "mul_aux"}


def _nt_label(M, N, K, kw):
    tags = [_EPI_NAMES.get(kw.get("epi", EPI_NONE), "?")]
    tags += [t for t, k in (("aux", "aux_out"), ("res", "residual"), ("bias", "bias")) if kw.get(k) is not None]
    return f"M={M} N={N} K={K} epi={'+'.join(tags)}"


def gemm_nt(a, b, out=None, **kw):       # noqa: F811  (profiling shim around the launch)
    if not PROFILE.on:
        return _gemm_nt_raw(a, b, out=out, **kw)
    M, K = a.shape
    N = b.shape[0]
    esz = 4 if (out is not None and out.dtype == torch.float32) or kw.get("out_dtype") == torch.float32 else 2
    nbytes = 2 * (M * K + N * K) + esz * M * N
    for extra in ("residual", "aux_in", "aux_out"):
        if kw.get(extra) is not None:
            nbytes += 2 * M * N
    return PROFILE.timed("gemm_nt_kernel", 2.0 * M * N * K, nbytes, lambda: _gemm_nt_raw(a, b, out=out, **kw), _nt_label(M, N, K, kw))


_gemm_nt_fp8_raw = gemm_nt_fp8


def gemm_nt_fp8(a, b, out=None, **kw):       # noqa: F811
    if not PROFILE.on:
        return _gemm_nt_fp8_raw(a, b, out=out, **kw)
    M, K = a.shape
    N = b.shape[0]
    esz = (2, 4, 1)[kw.get("out_kind", OUT_BF16)]
    nbytes = (M * K + N * K) + esz * M * N + sum(2 * M * N for extra in ("residual", "aux_out") if kw.get(extra) is not None)
    return PROFILE.timed("gemm_nt_fp8_kernel", 2.0 * M * N * K, nbytes, lambda: _gemm_nt_fp8_raw(a, b, out=out, **kw),
                         "e4m3 " + _nt_label(M, N, K, kw))


_gemm_nt_fp8_bwd_raw, _gemm_tn_fp8_raw = gemm_nt_fp8_bwd, gemm_tn_fp8_acc


def gemm_nt_fp8_bwd(a, b, out=None, **kw):       # noqa: F811
    if not PROFILE.on:
        return _gemm_nt_fp8_bwd_raw(a, b, out=out, **kw)
    M, K = a.shape
    N = b.shape[0]
    esz = {OUT_BF16: 2, OUT_F32: 4, OUT_E5M2: 1}[kw.get("out_kind", OUT_BF16)]
    nbytes = (M * K + N * K) + esz * M * N + (2 * M * N if kw.get("aux_in") is not None else 0)
    return PROFILE.timed("gemm_nt_fp8_kernel", 2.0 * M * N * K, nbytes, lambda: _gemm_nt_fp8_bwd_raw(a, b, out=out, **kw),
                         ("e5m2 " if kw.get("a_e5m2", True) else "e4m3 ") + _nt_label(M, N, K, kw))


def gemm_tn_fp8_acc(a, b, out, **kw):       # noqa: F811
    if not PROFILE.on:
        return _gemm_tn_fp8_raw(a, b, out, **kw)
    M, N1 = a.shape
    N2 = b.shape[1]
    return PROFILE.timed("gemm_tn8_kernel", 2.0 * M * N1 * N2, M * (N1 + N2) + 4 * N1 * N2,
                         lambda: _gemm_tn_fp8_raw(a, b, out, **kw), f"8-bit M={M} N1={N1} N2={N2}")


def gemm_tn_acc(a, b, out, alpha=1.0, colsum=None):       # noqa: F811
    if not PROFILE.on:
        return _gemm_tn_raw(a, b, out, alpha, colsum)
    M, N1 = a.shape
    N2 = b.shape[1]
    return PROFILE.timed("gemm_tn_kernel", 2.0 * M * N1 * N2, 2 * M * (N1 + N2) + 4 * N1 * N2,
                         lambda: _gemm_tn_raw(a, b, out, alpha, colsum), f"M={M} N1={N1} N2={N2}")
