"""Parity of the HIP contrastive head (through the C ABI) with the golden vectors and the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import clip_oracle as O

pytestmark = pytest.mark.gpu

# fp32 on the f32 MFMA: logits/loss agree with the reference's fp32 ATen result to rounding
LOGIT_ATOL = 2e-5          # |logit| <= 14.3
LOSS_RTOL = 2e-6
GRAD_RTOL, GRAD_ATOL = 2e-4, 2e-7


def _g(golden_dir, n):
    return np.load(os.path.join(golden_dir, f"g2_head_n{n}.npz"))


def _t(a, dev, grad=False):
    t = torch.from_numpy(np.asarray(a)).to(dev)
    return t.requires_grad_(True) if grad else t


@pytest.mark.parametrize("n", [8, 32, 37, 256])
def test_dropin_tail_matches_reference_golden(golden_dir, dev, n):
    """L2Normalize + ScaledLogits + CrossEntropyRows == mmgclip_model.py:128-136 + losses.py:36-44 (reference run)."""
    from mmgclip import head
    g = _g(golden_dir, n)
    img, txt = _t(g["img"], dev, True), _t(g["txt"], dev, True)
    ls = _t(g["logit_scale_param"], dev, True)
    ie, te = head.L2Normalize.apply(img), head.L2Normalize.apply(txt)
    np.testing.assert_allclose(ie.detach().cpu().numpy(), g["image_embeddings"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(te.detach().cpu().numpy(), g["text_embeddings"], rtol=1e-6, atol=1e-7)
    li, lt = head.ScaledLogits.apply(ie, te, ls.exp())
    np.testing.assert_allclose(li.detach().cpu().numpy(), g["logits_per_image"], rtol=0, atol=LOGIT_ATOL)
    np.testing.assert_allclose(lt.detach().cpu().numpy(), g["logits_per_text"], rtol=0, atol=LOGIT_ATOL)
    loss = (head.cross_entropy(li) + head.cross_entropy(lt)) / 2
    assert abs(loss.item() - float(g["clip_loss"])) <= LOSS_RTOL * abs(float(g["clip_loss"])) + 1e-7
    loss.backward()
    np.testing.assert_allclose(img.grad.cpu().numpy(), g["clip_dimg"], rtol=GRAD_RTOL, atol=GRAD_ATOL)
    np.testing.assert_allclose(txt.grad.cpu().numpy(), g["clip_dtxt"], rtol=GRAD_RTOL, atol=GRAD_ATOL)
    np.testing.assert_allclose(ls.grad.cpu().numpy(), g["clip_dlogit_scale"], rtol=GRAD_RTOL, atol=1e-6)


@pytest.mark.parametrize("n,k", [(3, 5), (40, 8), (1, 2)])
def test_rectangular_logits_and_their_backward(dev, n, k):
    """n images against k != n prompts (zero-shot scoring): both matrices and arbitrary upstream gradients vs torch autograd."""
    from mmgclip import head
    g = torch.Generator().manual_seed(n * 100 + k)
    img = torch.nn.functional.normalize(torch.randn(n, 512, generator=g), dim=1)
    txt = torch.nn.functional.normalize(torch.randn(k, 512, generator=g), dim=1)
    wi, wt = torch.randn(n, k, generator=g), torch.randn(k, n, generator=g)
    a, b, s = img.clone().requires_grad_(True), txt.clone().requires_grad_(True), torch.tensor(14.2857, requires_grad=True)
    ((s * a @ b.t()) * wi).sum().add(((s * b @ a.t()) * wt).sum()).backward()
    ad, bd, sd = img.to(dev).requires_grad_(True), txt.to(dev).requires_grad_(True), torch.tensor(14.2857, device=dev, requires_grad=True)
    li, lt = head.ScaledLogits.apply(ad, bd, sd)
    assert li.shape == (n, k) and lt.shape == (k, n)
    np.testing.assert_allclose(li.detach().cpu().numpy(), (14.2857 * img @ txt.t()).numpy(), atol=LOGIT_ATOL)
    np.testing.assert_allclose(lt.detach().cpu().numpy(), (14.2857 * txt @ img.t()).numpy(), atol=LOGIT_ATOL)
    ((li * wi.to(dev)).sum() + (lt * wt.to(dev)).sum()).backward()
    np.testing.assert_allclose(ad.grad.cpu().numpy(), a.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(bd.grad.cpu().numpy(), b.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(sd.grad.cpu().numpy(), s.grad.numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("n", [8, 32, 37, 256])
def test_fused_loss_matches_reference_golden(golden_dir, dev, n):
    """The no-materialisation path gives the same loss and gradients as the reference's CLIPLoss."""
    from mmgclip import head
    g = _g(golden_dir, n)
    img, txt = _t(g["img"], dev, True), _t(g["txt"], dev, True)
    ls = _t(g["logit_scale_param"], dev, True)
    loss = head.fused_clip_loss(head.L2Normalize.apply(img), head.L2Normalize.apply(txt), ls.exp())
    assert abs(loss.item() - float(g["clip_loss"])) <= LOSS_RTOL * abs(float(g["clip_loss"])) + 1e-7
    loss.backward()
    np.testing.assert_allclose(img.grad.cpu().numpy(), g["clip_dimg"], rtol=GRAD_RTOL, atol=GRAD_ATOL)
    np.testing.assert_allclose(txt.grad.cpu().numpy(), g["clip_dtxt"], rtol=GRAD_RTOL, atol=GRAD_ATOL)
    np.testing.assert_allclose(ls.grad.cpu().numpy(), g["clip_dlogit_scale"], rtol=GRAD_RTOL, atol=1e-6)


@pytest.mark.parametrize("P", [2, 4, 8])
def test_rank_simulated_global_loss(golden_dir, dev, P):
    """Simulate P ranks on one GPU: row blocks + LSE exchange == unsharded reference loss/grads."""
    from mmgclip import head
    from mmgclip._hip import call, ptr, stream
    g = _g(golden_dir, 256)
    ie, te = _t(g["image_embeddings"], dev), _t(g["text_embeddings"], dev)
    scale = _t(g["scale"], dev).reshape(1)
    N, D = ie.shape
    nl = N // P
    lse_i = torch.cat([head.rows_forward(ie[r * nl:(r + 1) * nl].contiguous(), te, scale, r * nl)[0] for r in range(P)])
    lse_t = torch.cat([head.rows_forward(te[r * nl:(r + 1) * nl].contiguous(), ie, scale, r * nl)[0] for r in range(P)])
    pos = torch.cat([head.rows_forward(ie[r * nl:(r + 1) * nl].contiguous(), te, scale, r * nl)[1] for r in range(P)])
    loss = ((lse_i - pos).sum() + (lse_t - pos).sum()) / (2 * N)
    assert abs(loss.item() - float(g["clip_loss"])) <= 5e-6 * abs(float(g["clip_loss"]))
    # oracle gradients w.r.t. the normalised embeddings
    a, b = ie.cpu().clone().requires_grad_(True), te.cpu().clone().requires_grad_(True)
    s = scale.cpu().clone().reshape(()).requires_grad_(True)
    l, _ = O.clip_loss(s * a @ b.t(), s * b @ a.t())
    l.backward()
    ds = torch.zeros(1, device=dev)
    for r in range(P):
        sl = slice(r * nl, (r + 1) * nl)
        x, y = ie[sl].contiguous(), te[sl].contiguous()
        dx, dy = torch.empty_like(x), torch.empty_like(y)
        call("mmg_clip_rows_bwd_fused", ptr(x), ptr(te), ptr(scale), ptr(lse_i[sl].contiguous()), ptr(lse_t), None,
             1.0 / (2 * N), nl, N, D, r * nl, ptr(dx), ptr(ds), stream())
        call("mmg_clip_rows_bwd_fused", ptr(y), ptr(ie), ptr(scale), ptr(lse_t[sl].contiguous()), ptr(lse_i), None,
             1.0 / (2 * N), nl, N, D, r * nl, ptr(dy), None, stream())
        np.testing.assert_allclose(dx.cpu().numpy(), a.grad[sl].numpy(), rtol=5e-4, atol=5e-7)
        np.testing.assert_allclose(dy.cpu().numpy(), b.grad[sl].numpy(), rtol=5e-4, atol=5e-7)
    assert abs(ds.item() - s.grad.item()) <= 1e-3 * abs(s.grad.item()) + 1e-6


def test_config_c3_two_rank_head_matches_reference_golden(golden_dir, dev):
    """BASELINE config C3's head: 2 ranks x 256 local pairs against the N = 512 global batch, D = 512.  Each simulated rank runs
    `mmg_clip_rows_fwd` on its 256 rows against the gathered 512, the LSE vectors are exchanged, `mmg_clip_rows_bwd_fused` forms the
    gradients of its own rows; loss and gradients w.r.t. the un-normalised projection outputs and the logit-scale parameter must equal
    what the reference's CLIPLoss (losses.py:28-44) gives on the unsharded problem (tests/golden/g2_head_n512.npz)."""
    from mmgclip import head
    from mmgclip._hip import call, ptr, stream
    g = np.load(os.path.join(golden_dir, "g2_head_n512.npz"))
    img, txt = _t(g["img"], dev, True), _t(g["txt"], dev, True)
    ie, te = head.L2Normalize.apply(img), head.L2Normalize.apply(txt)
    scale = _t(g["scale"], dev).reshape(1)
    N, D, P = 512, 512, 2
    nl = N // P
    assert ie.shape == (N, D)
    ied, ted = ie.detach(), te.detach()
    fw_i = [head.rows_forward(ied[r * nl:(r + 1) * nl].contiguous(), ted, scale, r * nl) for r in range(P)]
    fw_t = [head.rows_forward(ted[r * nl:(r + 1) * nl].contiguous(), ied, scale, r * nl) for r in range(P)]
    lse_i, lse_t = torch.cat([f[0] for f in fw_i]), torch.cat([f[0] for f in fw_t])          # the "all-gather" of exchange 2
    loss = sum(((fi[0] - fi[1]).sum() + (ft[0] - ft[1]).sum()) for fi, ft in zip(fw_i, fw_t)) / (2 * N)   # the scalar all-reduce
    assert abs(loss.item() - float(g["clip_loss"])) <= LOSS_RTOL * abs(float(g["clip_loss"])) + 1e-7
    d_ie, d_te, ds = torch.empty_like(ied), torch.empty_like(ted), torch.zeros(1, device=dev)
    for r in range(P):
        sl = slice(r * nl, (r + 1) * nl)
        x, y = ied[sl].contiguous(), ted[sl].contiguous()
        dx, dy = torch.empty_like(x), torch.empty_like(y)
        call("mmg_clip_rows_bwd_fused", ptr(x), ptr(ted), ptr(scale), ptr(lse_i[sl].contiguous()), ptr(lse_t), None,
             1.0 / (2 * N), nl, N, D, r * nl, ptr(dx), ptr(ds), stream())
        call("mmg_clip_rows_bwd_fused", ptr(y), ptr(ied), ptr(scale), ptr(lse_t[sl].contiguous()), ptr(lse_i), None,
             1.0 / (2 * N), nl, N, D, r * nl, ptr(dy), None, stream())
        d_ie[sl], d_te[sl] = dx, dy
    torch.autograd.backward([ie, te], [d_ie, d_te])
    np.testing.assert_allclose(img.grad.cpu().numpy(), g["clip_dimg"], rtol=GRAD_RTOL, atol=GRAD_ATOL)
    np.testing.assert_allclose(txt.grad.cpu().numpy(), g["clip_dtxt"], rtol=GRAD_RTOL, atol=GRAD_ATOL)
    # d loss / d logit_scale_param = (summed over ranks) d loss / d s * s,  s = exp(param)
    np.testing.assert_allclose(ds.item() * scale.item(), float(g["clip_dlogit_scale"]), rtol=GRAD_RTOL, atol=1e-6)
    # and the product's own autograd function on the whole batch (P = 1) agrees with the two-rank arithmetic
    img1, txt1 = _t(g["img"], dev, True), _t(g["txt"], dev, True)
    l1 = head.fused_clip_loss(head.L2Normalize.apply(img1), head.L2Normalize.apply(txt1), scale.reshape(()))
    l1.backward()
    assert abs(l1.item() - loss.item()) <= 2e-6 * abs(loss.item())
    np.testing.assert_allclose(img1.grad.cpu().numpy(), img.grad.cpu().numpy(), rtol=1e-4, atol=1e-7)


def test_large_global_batch_properties(dev):
    """N = 8192 x D = 512 (config C5 shape): size-independent properties instead of a CPU oracle pass.

    (1) loss of identical towers with huge scale -> ~0; (2) sum_j softmax == 1 <=> d loss / d scale identity:
    sum_ij g_ij == 0 for each row block; (3) agreement with a torch fp32 reference of the same op on the GPU.
    """
    from mmgclip import head
    torch.manual_seed(0)
    N, D = 8192, 512
    x = torch.nn.functional.normalize(torch.randn(N, D, device=dev), dim=1)
    y = torch.nn.functional.normalize(x + 0.5 * torch.randn(N, D, device=dev), dim=1)
    s = torch.tensor(14.2857, device=dev)
    xr, yr = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    loss = head.fused_clip_loss(xr, yr, s)
    loss.backward()
    z = s * x @ y.t()
    lab = torch.arange(N, device=dev)
    ref = (torch.nn.functional.cross_entropy(z, lab) + torch.nn.functional.cross_entropy(z.t(), lab)) / 2
    assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item())
    # gradient rows are orthogonal to nothing in particular, but sum over all rows of dX equals
    # s * sum_j (sum_i g_ij) y_j and sum_i g_ij = (colsum of softmax_row + 1 - 2)/(2N): check against torch
    xa, ya = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    za = s * xa @ ya.t()
    ((torch.nn.functional.cross_entropy(za, lab) + torch.nn.functional.cross_entropy(za.t(), lab)) / 2).backward()
    np.testing.assert_allclose(xr.grad.cpu().numpy(), xa.grad.cpu().numpy(), rtol=2e-3, atol=2e-8)
    np.testing.assert_allclose(yr.grad.cpu().numpy(), ya.grad.cpu().numpy(), rtol=2e-3, atol=2e-8)
    # identical towers, very sharp temperature: loss -> 0
    big = torch.tensor(200.0, device=dev)
    assert head.fused_clip_loss(x, x.clone(), big).item() < 1e-3


def test_cross_entropy_with_labels_matches_torch(dev):
    from mmgclip import head
    torch.manual_seed(1)
    z = torch.randn(37, 5, device=dev, requires_grad=True)
    lab = torch.randint(0, 5, (37,), device=dev)
    loss = head.cross_entropy(z, lab)
    loss.backward()
    zc = z.detach().cpu().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(zc, lab.cpu())
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-6
    np.testing.assert_allclose(z.grad.cpu().numpy(), zc.grad.numpy(), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("n,groups,noise", [(8, 2, 0.0), (37, 5, 0.05), (256, 12, 0.2), (1500, 40, 0.3), (64, 64, 0.0)])
def test_device_threshold_clustering_and_cluster_means(dev, n, groups, noise):
    """`_assign_labels` / `_average_logits` of AveragedMedicalCLIPLoss (reference losses.py:148-186) on the device against the
    oracle's restatement of the reference loops: identical labels (order-dependent greedy first fit), means and gradients."""
    from mmgclip import head
    from oracle import clip_oracle as O
    g = torch.Generator().manual_seed(n)
    centers = torch.nn.functional.normalize(torch.randn(groups, 64, generator=g), dim=1)
    member = torch.randint(0, groups, (n,), generator=g)
    txt = torch.nn.functional.normalize(centers[member] + noise * torch.randn(n, 64, generator=g), dim=1)
    sim = txt @ txt.t()                       # noise makes the relation non-transitive: the visiting order matters
    want = O.assign_labels(sim, 0.65)
    labels, counts, k = head.greedy_threshold_labels(sim.to(dev), 0.65)
    assert labels.dtype == torch.int64 and labels.cpu().tolist() == want
    assert k == max(want) + 1 and counts[:k].cpu().tolist() == np.bincount(want).tolist()
    logits = torch.randn(19, n, generator=g)
    w = torch.randn(19, k, generator=g)
    lr = logits.clone().requires_grad_(True)
    (O.average_logits(lr, want) * w).sum().backward()
    ld = logits.to(dev).requires_grad_(True)
    out = head.ClusterMeanCols.apply(ld, labels, counts, k)
    np.testing.assert_allclose(out.detach().cpu().numpy(), O.average_logits(logits, want).numpy(), rtol=1e-5, atol=1e-6)
    (out * w.to(dev)).sum().backward()
    np.testing.assert_allclose(ld.grad.cpu().numpy(), lr.grad.numpy(), rtol=1e-5, atol=1e-7)
