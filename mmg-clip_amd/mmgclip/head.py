"""Host side of the contrastive head: autograd wrappers over the fp32 gfx950 kernels of csrc/clip_head.hip.

Reference arithmetic (path:line in the reference tree):
    mmgclip/networks/mmgclip_model.py:128-136  normalise, exp(logit_scale), two logit matmuls
    mmgclip/loss/losses.py:36-44               CLIPLoss
PyTorch is used for allocation / stream / autograd plumbing only; every FLOP below runs in the HIP library.
"""
import torch

from . import _hip
from ._hip import call, ptr, stream


def _f32c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class L2Normalize(torch.autograd.Function):
    """y = x / ||x||_2 per row (no epsilon) — mmgclip_model.py:128-129."""

    @staticmethod
    def forward(ctx, x):
        _hip.require_gpu(x)
        x = _f32c(x)
        y = torch.empty_like(x)
        norm = torch.empty(x.shape[0], device=x.device, dtype=torch.float32)
        call("mmg_l2norm_fwd", ptr(x), ptr(y), ptr(norm), x.shape[0], x.shape[1], stream())
        ctx.save_for_backward(y, norm)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, norm = ctx.saved_tensors
        dy = _f32c(dy)
        dx = torch.empty_like(y)
        call("mmg_l2norm_bwd", ptr(y), ptr(norm), ptr(dy), ptr(dx), y.shape[0], y.shape[1], stream())
        return dx


def rows_forward(x_loc, y_all, scale, diag_off=0, want_logits=False):
    """lse, pos (and logits) of s * x_loc @ y_all.T — one launch, no autograd."""
    n_loc, D = x_loc.shape
    N = y_all.shape[0]
    lse = torch.empty(n_loc, device=x_loc.device, dtype=torch.float32)
    pos = torch.empty(n_loc, device=x_loc.device, dtype=torch.float32)
    logits = torch.empty(n_loc, N, device=x_loc.device, dtype=torch.float32) if want_logits else None
    call("mmg_clip_rows_fwd", ptr(x_loc), ptr(y_all), ptr(scale), n_loc, N, D, diag_off, ptr(lse), ptr(pos),
         ptr(logits), N if want_logits else 0, stream())
    return lse, pos, logits


class ScaledLogits(torch.autograd.Function):
    """logits_per_image, logits_per_text = s * I @ T.t(), s * T @ I.t()  (mmgclip_model.py:135-136).

    Materialises both matrices ([n,k] and [k,n]; k = n in training) because the reference API returns them; the backward takes arbitrary
    upstream gradients of both (mmg_clip_rows_bwd_dense).
    """

    @staticmethod
    def forward(ctx, img, txt, scale):
        _hip.require_gpu(img, txt, scale)
        img, txt = _f32c(img), _f32c(txt)
        scale = _f32c(scale.reshape(1))
        _, _, li = rows_forward(img, txt, scale, 0, True)
        _, _, lt = rows_forward(txt, img, scale, 0, True)
        ctx.save_for_backward(img, txt, scale)
        return li, lt

    @staticmethod
    def backward(ctx, dli, dlt):
        img, txt, scale = ctx.saved_tensors
        n, D = img.shape
        k = txt.shape[0]                  # k != n for zero-shot prompt scoring (PromptClassifier): Li [n,k], Lt [k,n]
        dli = _f32c(dli) if dli is not None else None
        dlt = _f32c(dlt) if dlt is not None else None
        dimg = torch.empty_like(img)
        dtxt = torch.empty_like(txt)
        dscale = torch.zeros(1, device=img.device, dtype=torch.float32)
        # d img = s (dLi + dLt^T) T ; d txt = s (dLt + dLi^T) I ; ds counted once (first call)
        call("mmg_clip_rows_bwd_dense", ptr(img), ptr(txt), ptr(scale), ptr(dli), k, ptr(dlt), n, n, k, D,
             ptr(dimg), ptr(dscale), stream())
        call("mmg_clip_rows_bwd_dense", ptr(txt), ptr(img), ptr(scale), ptr(dlt), n, ptr(dli), k, k, n, D,
             ptr(dtxt), None, stream())
        return dimg, dtxt, dscale.reshape(())


class CrossEntropyRows(torch.autograd.Function):
    """mean_r ( logsumexp(z_r) - z_r[label_r] ) * weight — F.cross_entropy of losses.py:40-41 on device."""

    @staticmethod
    def forward(ctx, logits, labels, weight):
        _hip.require_gpu(logits)
        logits = _f32c(logits)
        rows, C = logits.shape
        if labels is not None:
            labels = labels.to(device=logits.device, dtype=torch.int64).contiguous()
        lse = torch.empty(rows, device=logits.device, dtype=torch.float32)
        loss = torch.zeros(1, device=logits.device, dtype=torch.float32)
        w = float(weight) / rows
        call("mmg_ce_rows_fwd", ptr(logits), C, ptr(labels), rows, C, w, ptr(lse), ptr(loss), stream())
        ctx.save_for_backward(logits, lse, labels if labels is not None else torch.empty(0))
        ctx.has_labels = labels is not None
        ctx.w = w
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        logits, lse, labels = ctx.saved_tensors
        rows, C = logits.shape
        gout = _f32c(gout.reshape(1))
        d = torch.empty_like(logits)
        call("mmg_ce_rows_bwd", ptr(logits), C, ptr(labels) if ctx.has_labels else None, ptr(lse), ptr(gout),
             ctx.w, rows, C, ptr(d), C, stream())
        return d, None, None


def cross_entropy(logits, labels=None, weight=1.0):
    return CrossEntropyRows.apply(logits, labels, weight)


class _HipHeadBackend:
    """The product arithmetic of the fused loss: three entry points of libmmgclip_hip.so.

    No product signature selects anything else; tests/test_distributed_cpu.py replaces this module attribute in its own worker
    processes to exercise the collective choreography below on CPU/gloo."""

    @staticmethod
    def rows_forward(x_loc, y_all, scale, diag_off):
        _hip.require_gpu(x_loc, y_all, scale)
        lse, pos, _ = rows_forward(x_loc, y_all, scale, diag_off)
        return lse, pos

    @staticmethod
    def loss_sum(lse_i, pos_i, lse_t, pos_t, coef):
        loss = torch.zeros(1, device=lse_i.device, dtype=torch.float32)
        call("mmg_clip_loss_reduce", ptr(lse_i), ptr(pos_i), ptr(lse_t), ptr(pos_t), lse_i.shape[0], coef, ptr(loss), stream())
        return loss

    @staticmethod
    def rows_backward(x_loc, y_all, scale, lse_row, lse_col, gout, coef, diag_off, want_dscale):
        n_loc, D = x_loc.shape
        dx = torch.empty_like(x_loc)
        dscale = torch.zeros(1, device=x_loc.device, dtype=torch.float32) if want_dscale else None
        call("mmg_clip_rows_bwd_fused", ptr(x_loc), ptr(y_all), ptr(scale), ptr(lse_row), ptr(lse_col), ptr(gout), coef,
             n_loc, y_all.shape[0], D, diag_off, ptr(dx), ptr(dscale), stream())
        return dx, dscale


class FusedClipLoss(torch.autograd.Function):
    """CLIPLoss over (already L2-normalised) embeddings without materialising logits.

    loss = 1/(2N) [ sum_i (lse_i(A) - A_ii) + sum_j (lse_j(A^T) - A_jj) ],  A = s I T^T   (losses.py:36-44)

    With a communicator the columns are the all-gathered embeddings of every rank: exchange 1
    gathers the normalised embeddings, exchange 2 the two log-sum-exp vectors; gradients w.r.t. the local
    embeddings are then complete locally (no reduce-scatter).  `comm=None` is the reference's local-batch loss.
    The returned loss is the GLOBAL mean (identical on every rank); gradients are d(global loss)/d(local rows),
    so parameter gradients must be SUMMED across ranks (mmgclip/distributed.py does that).  d loss / d scale is the
    LOCAL rows' share; summed over ranks it is the full derivative.
    """

    @staticmethod
    def forward(ctx, img, txt, scale, comm):
        be = _HipHeadBackend
        img, txt = _f32c(img), _f32c(txt)
        scale = _f32c(scale.reshape(1))
        n_loc, D = img.shape
        if comm is None or not comm.active:
            img_all, txt_all, off, N = img, txt, 0, n_loc
        else:
            img_all, txt_all = comm.all_gather_rows(img), comm.all_gather_rows(txt)
            off, N = comm.rank * n_loc, n_loc * comm.world_size
        lse_i, pos_i = be.rows_forward(img, txt_all, scale, off)
        lse_t, pos_t = be.rows_forward(txt, img_all, scale, off)
        loss = be.loss_sum(lse_i, pos_i, lse_t, pos_t, 1.0 / (2.0 * N))
        if comm is not None and comm.active:
            lse_i_all, lse_t_all = comm.all_gather_rows(lse_i), comm.all_gather_rows(lse_t)
            loss = comm.all_reduce_sum(loss)
        else:
            lse_i_all, lse_t_all = lse_i, lse_t
        ctx.save_for_backward(img, txt, scale, img_all, txt_all, lse_i, lse_t, lse_i_all, lse_t_all)
        ctx.off, ctx.N, ctx.be = off, N, be
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        img, txt, scale, img_all, txt_all, lse_i, lse_t, lse_i_all, lse_t_all = ctx.saved_tensors
        gout = _f32c(gout.reshape(1))
        coef = 1.0 / (2.0 * ctx.N)
        dimg, dscale = ctx.be.rows_backward(img, txt_all, scale, lse_i, lse_t_all, gout, coef, ctx.off, True)
        dtxt, _ = ctx.be.rows_backward(txt, img_all, scale, lse_t, lse_i_all, gout, coef, ctx.off, False)
        return dimg, dtxt, dscale.reshape(()), None


def fused_clip_loss(image_embeddings, text_embeddings, logit_scale, comm=None):
    """CLIPLoss on normalised embeddings; `logit_scale` is the already exponentiated scale (a tensor)."""
    return FusedClipLoss.apply(image_embeddings, text_embeddings, logit_scale, comm)


def greedy_threshold_labels(sim, threshold):
    """losses.py:148-162 on the device: (labels int64 [n], counts int32 [n], k) — one 4-byte read-back for k."""
    sim = _f32c(sim.detach())
    n = sim.shape[0]
    labels = torch.empty(n, device=sim.device, dtype=torch.int64)
    counts = torch.empty(n, device=sim.device, dtype=torch.int32)
    k = torch.empty(1, device=sim.device, dtype=torch.int32)
    call("mmg_greedy_threshold_labels", ptr(sim), sim.stride(0), n, float(threshold), ptr(labels), ptr(counts), ptr(k), stream())
    return labels, counts, int(k.item())


class ClusterMeanCols(torch.autograd.Function):
    """[n,N] logits -> [n,k] per-cluster column means (losses.py:164-186) and the matching gradient."""

    @staticmethod
    def forward(ctx, logits, labels, counts, k):
        _hip.require_gpu(logits, labels, counts)
        logits = _f32c(logits)
        n, N = logits.shape
        out = torch.empty(n, k, device=logits.device, dtype=torch.float32)
        call("mmg_cluster_mean_cols_fwd", ptr(logits), logits.stride(0), n, N, ptr(labels), ptr(counts), k, ptr(out), k, stream())
        ctx.save_for_backward(labels, counts)
        ctx.shape = (n, N)
        return out

    @staticmethod
    def backward(ctx, dout):
        labels, counts = ctx.saved_tensors
        n, N = ctx.shape
        dout = _f32c(dout)
        dl = torch.empty(n, N, device=dout.device, dtype=torch.float32)
        call("mmg_cluster_mean_cols_bwd", ptr(dout), dout.stride(0), n, N, ptr(labels), ptr(counts), ptr(dl), N, stream())
        return dl, None, None, None
