"""Pin the CPU oracle against the golden vectors generated from the reference's own files (make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import clip_oracle as O


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_g1_projection_heads(golden_dir):
    g = np.load(os.path.join(golden_dir, "g1_projection.npz"))
    x = _t(g["x"])
    np.testing.assert_allclose(O.linear_projection(x, _t(g["linear.weight"])).numpy(), g["linear.y"], rtol=1e-6, atol=1e-6)
    y = O.multi_linear_head(x, [_t(g["multi.layers.0.weight"]), _t(g["multi.layers.1.weight"])],
                            [_t(g["multi.layers.0.bias"]), _t(g["multi.layers.1.bias"])])
    np.testing.assert_allclose(y.numpy(), g["multi.y"], rtol=1e-5, atol=1e-6)
    y = O.mlp_projection_head(x, _t(g["mlp.projection.weight"]), _t(g["mlp.projection.bias"]), _t(g["mlp.fc.weight"]),
                              _t(g["mlp.fc.bias"]), _t(g["mlp.layer_norm.weight"]), _t(g["mlp.layer_norm.bias"]))
    np.testing.assert_allclose(y.numpy(), g["mlp.y"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("n", [8, 32, 37, 256])
def test_g2_head(golden_dir, n):
    g = np.load(os.path.join(golden_dir, f"g2_head_n{n}.npz"))
    img = _t(g["img"]).requires_grad_(True)
    txt = _t(g["txt"]).requires_grad_(True)
    ls = _t(g["logit_scale_param"]).requires_grad_(True)
    out = O.forward_tail(img, txt, ls)
    np.testing.assert_allclose(out["image_embeddings"].detach().numpy(), g["image_embeddings"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(out["logits_per_image"].detach().numpy(), g["logits_per_image"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(out["logits_per_text"].detach().numpy(), g["logits_per_text"], rtol=1e-5, atol=1e-5)
    loss, labels = O.clip_loss(out["logits_per_image"], out["logits_per_text"])
    assert abs(float(loss) - float(g["clip_loss"])) <= 1e-6 * max(1.0, abs(float(g["clip_loss"])))
    assert (labels.numpy() == g["clip_labels"]).all()
    loss.backward()
    np.testing.assert_allclose(img.grad.numpy(), g["clip_dimg"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(txt.grad.numpy(), g["clip_dtxt"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(ls.grad.numpy(), g["clip_dlogit_scale"], rtol=1e-4, atol=1e-7)
    # MMGCLIPLoss
    txt2 = _t(g["txt2"])
    out = O.forward_tail(_t(g["img"]), _t(g["txt"]), _t(g["logit_scale_param"]))
    loss2, _ = O.mmgclip_loss(out["image_embeddings"], out["text_embeddings"], O.l2_normalize(txt2), out["logit_scale"])
    assert abs(float(loss2) - float(g["mmg_loss"])) <= 1e-6 * max(1.0, abs(float(g["mmg_loss"])))


def test_g2_sharded_equals_unsharded(golden_dir):
    """P-rank row blocks + LSE exchange reproduce the unsharded CLIPLoss and its gradients."""
    g = np.load(os.path.join(golden_dir, "g2_head_n32.npz"))
    ie, te, s = _t(g["image_embeddings"]), _t(g["text_embeddings"]), float(g["scale"])
    N = ie.shape[0]
    for P in (1, 2, 4, 8):
        nl = N // P
        lse_i = torch.cat([O.sharded_rows(ie[r * nl:(r + 1) * nl], te, s, r * nl)[0] for r in range(P)])
        lse_t = torch.cat([O.sharded_rows(te[r * nl:(r + 1) * nl], ie, s, r * nl)[0] for r in range(P)])
        pos = torch.cat([O.sharded_rows(ie[r * nl:(r + 1) * nl], te, s, r * nl)[1] for r in range(P)])
        loss = ((lse_i - pos).sum() + (lse_t - pos).sum()) / (2 * N)
        assert abs(float(loss) - float(g["clip_loss"])) < 2e-6 * abs(float(g["clip_loss"]))
        # gradients w.r.t. the NORMALISED embeddings, from autograd on the unsharded formula
        a = ie.clone().requires_grad_(True)
        b = te.clone().requires_grad_(True)
        sc = torch.tensor(s, requires_grad=True)
        l, _ = O.clip_loss(sc * a @ b.t(), sc * b @ a.t())
        l.backward()
        ds = 0.0
        for r in range(P):
            sl = slice(r * nl, (r + 1) * nl)
            da, dsr = O.sharded_rows_grad(ie[sl], te, s, lse_i[sl], lse_t, 1.0 / (2 * N), r * nl)
            db, _ = O.sharded_rows_grad(te[sl], ie, s, lse_t[sl], lse_i, 1.0 / (2 * N), r * nl)
            np.testing.assert_allclose(da.numpy(), a.grad[sl].numpy(), rtol=2e-4, atol=1e-7)
            np.testing.assert_allclose(db.numpy(), b.grad[sl].numpy(), rtol=2e-4, atol=1e-7)
            ds += float(dsr)
        assert abs(ds - float(sc.grad)) < 1e-4 * max(1e-3, abs(float(sc.grad)))


def test_g2_config_c3_two_ranks(golden_dir):
    """BASELINE config C3 (2 ranks x 256 pairs, N = 512, D = 512): the oracle's per-rank arithmetic (row blocks + LSE exchange, SURVEY
    §8e) against the reference's CLIPLoss on the unsharded batch (g2_head_n512.npz, written by the reference's losses.py)."""
    g = np.load(os.path.join(golden_dir, "g2_head_n512.npz"))
    img, txt = _t(g["img"]).requires_grad_(True), _t(g["txt"]).requires_grad_(True)
    ls = _t(g["logit_scale_param"]).requires_grad_(True)
    ie, te, s = O.l2_normalize(img), O.l2_normalize(txt), ls.exp()
    N, P = 512, 2
    nl = N // P
    parts = [(O.sharded_rows(ie[r * nl:(r + 1) * nl], te, s, r * nl), O.sharded_rows(te[r * nl:(r + 1) * nl], ie, s, r * nl)) for r in range(P)]
    loss = sum(((a[0] - a[1]).sum() + (b[0] - b[1]).sum()) for a, b in parts) / (2 * N)
    assert abs(float(loss) - float(g["clip_loss"])) <= 2e-6 * abs(float(g["clip_loss"]))
    loss.backward()
    np.testing.assert_allclose(img.grad.numpy(), g["clip_dimg"], rtol=2e-4, atol=2e-7)
    np.testing.assert_allclose(txt.grad.numpy(), g["clip_dtxt"], rtol=2e-4, atol=2e-7)
    np.testing.assert_allclose(ls.grad.numpy(), g["clip_dlogit_scale"], rtol=2e-4, atol=1e-6)


def test_g3_averaged_loss_and_notebook_kats(golden_dir):
    g = np.load(os.path.join(golden_dir, "g3_averaged.npz"))
    labels = O.assign_labels(_t(g["nb_cos"]), 0.8)
    assert labels == list(g["nb_labels"]) == [0, 1, 0, 0, 0, 1, 0, 2]          # notebooks/loss.ipynb cell 13
    avg = O.average_logits(_t(g["nb_logits"]), labels)
    np.testing.assert_allclose(avg.softmax(-1)[0].numpy(), [0.3354, 0.2250, 0.4396], atol=5e-5)   # cell 17
    ce = torch.nn.functional.cross_entropy(avg, torch.tensor(labels))
    assert abs(float(ce) - 1.2048) < 5e-5                                                          # cell 18
    assert O.assign_labels(_t(g["alt_cos"]), 0.65) == [0, 1, 0, 1, 0, 1, 0, 1]                      # losses.py:129-139
    img, txt, ls = _t(g["img"]).requires_grad_(True), _t(g["txt"]).requires_grad_(True), _t(g["logit_scale_param"])
    out = O.forward_tail(img, txt, ls)
    loss, lab = O.averaged_medical_clip_loss(**out)
    assert (lab.numpy() == g["labels"]).all()
    assert abs(float(loss) - float(g["loss"])) < 1e-6 * abs(float(g["loss"]))
    loss.backward()
    np.testing.assert_allclose(img.grad.numpy(), g["dimg"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(txt.grad.numpy(), g["dtxt"], rtol=1e-4, atol=1e-7)


def test_g4_lr_schedule(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "g4_lr_schedule.json")))
    for key, lrs in ref.items():
        total, warm = key.split("_")
        total = int(total)
        warm = float(warm) if "." in warm else int(warm)
        mine = [5e-5 * O.warmup_cosine_multiplier(s, total, warm) for s in range(total)]
        np.testing.assert_allclose(mine, lrs, rtol=1e-12, atol=0)
    assert ref["30_0.1"][0] == 0.0 and abs(ref["30_0.1"][3] - 5e-5) < 1e-18      # epoch 1 trains at lr 0


def test_g7_early_stopper(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "g7_early_stopper.json")))
    mine = O.early_stopper_trace(ref["trace"], patience=5)
    assert mine == ref["states"]
    assert ref["checkpoint_keys"] == sorted(["epoch", "model_state_dict", "optimizer_state_dict", "val_loss",
                                             "best_score", "counter"])
