"""Thin host wrappers over the bf16 MFMA GEMM entry points (csrc/gemm_bf16.hip). No autograd here."""
import torch

from ._hip import call, ptr, stream

EPI_NONE, EPI_GELU, EPI_DGELU, EPI_RELU, EPI_DRELU = 0, 1, 2, 3, 4


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


class _GemmProfile:
    """HIP-event timing of every gemm_nt launch (bench.py's roofline leg).  Events are recorded on the stream the kernel
    is launched on (torch's current stream), so the durations are the kernel's own; flops are the algorithmic 2*M*N*K."""

    def __init__(self):
        self.on = False
        self.records = []        # (start, end, flops, bytes)

    def enable(self):
        self.on, self.records = True, []

    def disable(self):
        self.on = False

    def summary(self, peak_tflops):
        if not self.records:
            return None
        torch.cuda.synchronize()
        t_ms = sum(s.elapsed_time(e) for s, e, _, _ in self.records)
        flops = sum(f for _, _, f, _ in self.records)
        byts = sum(b for _, _, _, b in self.records)
        n = len(self.records)
        ach = flops / (t_ms * 1e-3) / 1e12
        return {"kernel": "gemm_nt_kernel (bf16 MFMA, all shapes of the step)", "bound": "mfma", "achieved": round(ach, 1),
                "peak": peak_tflops, "unit": "TFLOP/s", "frac": round(ach / peak_tflops, 4), "traffic": None,
                "launches": n, "avg_launch_us": round(t_ms * 1000.0 / n, 1), "total_ms": round(t_ms, 2),
                "algorithmic_bytes_per_launch": round(byts / n), "algorithmic_gbytes_per_s": round(byts / (t_ms * 1e-3) / 1e9, 1)}


PROFILE = _GemmProfile()
_gemm_nt_raw = gemm_nt


def gemm_nt(a, b, out=None, **kw):       # noqa: F811  (profiling shim around the launch)
    if not PROFILE.on:
        return _gemm_nt_raw(a, b, out=out, **kw)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    r = _gemm_nt_raw(a, b, out=out, **kw)
    e.record()
    M, K = a.shape
    N = b.shape[0]
    byts = 2 * (M * K + N * K) + r.element_size() * M * N
    PROFILE.records.append((s, e, 2.0 * M * N * K, byts))
    return r
