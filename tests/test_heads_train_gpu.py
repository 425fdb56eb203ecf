"""Projection heads vs the reference golden (g1) and a short end-to-end run of train.py's experiment on synthetic data."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
# heads run their matrix products on the bf16 MFMA path (fp32 accumulate): 8-bit mantissa inputs
RTOL, ATOL = 3e-2, 3e-2


def _t(a, dev, grad=False):
    t = torch.from_numpy(np.asarray(a)).to(dev)
    return t.requires_grad_(True) if grad else t


def test_projection_heads_match_reference_golden(golden_dir, dev):
    from mmgclip.networks.projection_controller import get_projection_head
    g = np.load(os.path.join(golden_dir, "g1_projection.npz"))
    x, gy = _t(g["x"], dev, True), _t(g["gy"], dev)
    lin = get_projection_head("LinearProjectionLayer")(embedding_dim=96, projection_dim=64, dropout=0.5).to(dev)
    lin.load_state_dict({"layer.weight": _t(g["linear.weight"], dev)})
    y = lin(x)
    y.backward(gy)
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["linear.y"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["linear.dx"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(lin.layer.weight.grad.cpu().numpy(), g["linear.dweight"], rtol=RTOL, atol=2 * ATOL)

    x = _t(g["x"], dev, True)
    ml = get_projection_head("MultiLinearHead")(embedding_dim=96, projection_dim=[96, 64], dropout=0.0).to(dev).train()
    ml.load_state_dict({f"layers.{i}.{k}": _t(g[f"multi.layers.{i}.{k}"], dev) for i in (0, 1) for k in ("weight", "bias")})
    y = ml(x)
    y.backward(gy)
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["multi.y"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["multi.dx"], rtol=RTOL, atol=ATOL)
    for i in (0, 1):
        np.testing.assert_allclose(ml.layers[i].weight.grad.cpu().numpy(), g[f"multi.layers.{i}.dweight"], rtol=RTOL, atol=2 * ATOL)
        np.testing.assert_allclose(ml.layers[i].bias.grad.cpu().numpy(), g[f"multi.layers.{i}.dbias"], rtol=RTOL, atol=2 * ATOL)

    x = _t(g["x"], dev, True)
    mlp = get_projection_head("MLPProjectionHead")(embedding_dim=96, projection_dim=64, dropout=0.0).to(dev).train()
    mlp.load_state_dict({k[4:]: _t(g[k], dev) for k in g.files if k.startswith("mlp.") and not k.startswith("mlp.grad.") and k not in ("mlp.y", "mlp.dx")})
    y = mlp(x)
    y.backward(gy)
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["mlp.y"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["mlp.dx"], rtol=2 * RTOL, atol=2 * ATOL)
    for name, p in mlp.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), g["mlp.grad." + name], rtol=2 * RTOL, atol=4 * ATOL)


def test_dropout_head_statistics(dev):
    from mmgclip import ops
    torch.manual_seed(0)
    x = torch.ones(256, 512, device=dev, requires_grad=True)
    y = ops.dropout(x, 0.2, True)
    kept = (y > 0).float().mean().item()
    assert abs(kept - 0.8) < 0.01 and abs(y.mean().item() - 1.0) < 0.02
    y.sum().backward()
    assert torch.equal((x.grad > 0), (y > 0)) and abs(x.grad.max().item() - 1.25) < 1e-6
    assert ops.dropout(x, 0.2, False) is x


def test_classifier_experiment_runs_like_the_reference_loop(dev, tmp_path, monkeypatch):
    """ClassifierExperiment.run(): 3 short epochs on synthetic batches (reference-faithful mode: features + frozen BERT).
    Epoch 1 runs at lr 0, afterwards the loss must fall; a checkpoint with the reference's keys is written."""
    from mmgclip.config import compose
    from mmgclip.dataset.synthetic import SyntheticLoader
    from mmgclip.experiments.experiments_controller import create_experiment
    from mmgclip.networks import bert
    orig = bert.BertConfigLite.__init__

    def small(self, **kw):
        kw.setdefault("num_hidden_layers", 2)
        kw.setdefault("vocab_size", 3000)
        orig(self, **kw)
    monkeypatch.setattr(bert.BertConfigLite, "__init__", small)
    cfg_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mmg-clip_amd", "configs")
    cfg = compose(cfg_dir, "train_binary_class_clf", ["networks.text_encoder.random_init=true", "tokenizer=bert_clinical_seqlen=77",
                                                      "scheduler=warmup1_epo15", "optimizer.config.learning_rate=3e-3"])
    cfg.scheduler.config.epochs = 4
    cfg.checkpoints.checkpoints_export_dir = str(tmp_path / "ckpt")
    cfg.base.tensorboard_export_dir = str(tmp_path / "tb")
    torch.manual_seed(0)
    loader = SyntheticLoader(4, 32, seed=5, S=77, vocab_size=3000)
    exp = create_experiment("classification")(config=cfg, train_dataloader=loader, valid_dataloader=SyntheticLoader(1, 32, seed=5, S=77, vocab_size=3000),
                                              test_dataloader=None, tokenizer=None)
    w0 = exp.model.image_projection_layer.layer.weight.detach().clone()
    l1 = exp.train()
    assert torch.equal(w0, exp.model.image_projection_layer.layer.weight.detach())      # epoch 1: lr = 0
    exp.current_epoch = 1
    l2 = exp.train()
    exp.current_epoch = 2
    l3 = exp.train()
    assert l3 < l1 and not torch.equal(w0, exp.model.image_projection_layer.layer.weight.detach())
    val = exp.validate()
    assert len(val) == 5 and np.isfinite(val[0])                # (loss, auc_malig, auc_shapes, auc_birads, auc_mean)
    assert 0.0 <= val[1] <= 1.0 and val[2] == -1 and val[3] == -1 and val[4] == -1    # binary config: malignancy only
    exp.early_stopper(val[0], 2, exp.model, exp.optimizer, exp.ckp_path)
    ckpt = torch.load(exp.ckp_path, weights_only=False)
    assert sorted(ckpt) == sorted(["epoch", "model_state_dict", "optimizer_state_dict", "val_loss", "best_score", "counter"])
    assert "text_encoder.model.embeddings.word_embeddings.weight" in ckpt["model_state_dict"]
    assert "image_projection_layer.layer.weight" in ckpt["model_state_dict"] and "logit_scale" not in ckpt["model_state_dict"]


def test_validation_prompt_scores_match_oracle(dev, monkeypatch):
    """Zero-shot prompt scoring of validate(): cached prompt embeddings + the [n,D]x[D,k] logit kernel equal the reference
    arithmetic (logit_scale * image_embeddings @ prompt_embeddings.t(), ClassifierExperiment.py:192-229) on the oracle."""
    from mmgclip import head
    from mmgclip.config import compose
    from mmgclip.dataset.synthetic import synthetic_batch, synthetic_prompt_tokens, validation_prompts
    from mmgclip.networks import bert
    from mmgclip.networks.mmgclip_model import MMGCLIP
    from oracle import clip_oracle as O
    from oracle import encoders_oracle as E
    orig = bert.BertConfigLite.__init__

    def small(self, **kw):
        kw.setdefault("num_hidden_layers", 2)
        kw.setdefault("vocab_size", 3000)
        orig(self, **kw)
    monkeypatch.setattr(bert.BertConfigLite, "__init__", small)
    cfg_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mmg-clip_amd", "configs")
    cfg = compose(cfg_dir, "train_exam_reports_clf", ["networks.text_encoder.random_init=true", "tokenizer=bert_clinical_seqlen=77",
                                                      "networks/dropout=dropout0"])
    torch.manual_seed(0)
    model = MMGCLIP(cfg).eval()
    strings = validation_prompts(["MassShapeLabels"])["shapes"]
    assert strings == ["Mass shape is unknown.", "Mass shape is oval.", "Mass shape is round.", "Mass shape is irregular."]
    ptok = synthetic_prompt_tokens(strings, 77, 3000)
    batch = synthetic_batch(16, S=77, vocab_size=3000, seed=21)
    with torch.no_grad():
        out = model(batch, validation=True)
        te = head.L2Normalize.apply(model.text_projection_layer(model.encode_text({"text_tokens": ptok})))
        _, _, sims = head.rows_forward(out["image_embeddings"].contiguous(), te.contiguous(), out["logit_scale"].reshape(1), 0, True)
    sd = {k: v.detach().cpu().float() for k, v in model.state_dict().items()}
    bsd = {k[len("text_encoder.model."):]: v for k, v in sd.items() if k.startswith("text_encoder.model.")}
    hid = E.bert_forward(bsd, ptok["input_ids"].cpu(), ptok["attention_mask"].cpu(), ptok["token_type_ids"].cpu())
    tf = O.eos_pool(hid, ptok["attention_mask"].cpu())
    wts = [sd[f"text_projection_layer.layers.{i}.weight"] for i in (0, 1)]
    bs = [sd[f"text_projection_layer.layers.{i}.bias"] for i in (0, 1)]
    te_ref = O.l2_normalize(O.multi_linear_head(tf, wts, bs))
    iw = [sd[f"image_projection_layer.layers.{i}.weight"] for i in (0, 1)]
    ib = [sd[f"image_projection_layer.layers.{i}.bias"] for i in (0, 1)]
    ie_ref = O.l2_normalize(O.multi_linear_head(batch["image_features"].cpu().flatten(1), iw, ib))
    ref = (1 / 0.07) * ie_ref @ te_ref.t()
    np.testing.assert_allclose(sims.cpu().numpy(), ref.numpy(), atol=0.2)      # bf16 towers/heads; |logit| <= 14.3
    assert sims.shape == (16, 4)


@pytest.mark.parametrize("key,classes", [("BenignMalignantDatasetLabels", {"benign": 0, "malignant": 1}),
                                         ("MassShapeLabels", {"unknown": 0, "oval": 1, "round": 2, "irregular": 3})])
def test_evaluator_zeroshot_label_prompt_matches_oracle(dev, monkeypatch, key, classes):
    """Evaluator.zeroshot_label_prompt (evaluator.py:321-478): prompts encoded once, [n,512]x[512,k] scoring kernel, sklearn on the
    host - against the oracle's restatement fed with the oracle's own embeddings of the same model."""
    from mmgclip.config import compose
    from mmgclip.dataset.synthetic import synthetic_batch, synthetic_prompt_tokens
    from mmgclip.evaluator import Evaluator, label_prompts
    from mmgclip.networks import bert
    from mmgclip.networks.mmgclip_model import MMGCLIP
    from oracle import clip_oracle as O
    from oracle import encoders_oracle as E
    orig = bert.BertConfigLite.__init__

    def small(self, **kw):
        kw.setdefault("num_hidden_layers", 2)
        kw.setdefault("vocab_size", 3000)
        orig(self, **kw)
    monkeypatch.setattr(bert.BertConfigLite, "__init__", small)
    cfg_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mmg-clip_amd", "configs")
    cfg = compose(cfg_dir, "train_binary_class_clf", ["networks.text_encoder.random_init=true", "tokenizer=bert_clinical_seqlen=77"])
    torch.manual_seed(0)
    model = MMGCLIP(cfg).eval()
    ev = Evaluator(cfg, test_dataloader=None, tokenizer=None, model=model)
    batch = synthetic_batch(96, S=77, vocab_size=3000, seed=33)
    names = list(classes)
    g = torch.Generator().manual_seed(1)
    y = torch.randint(0, len(names), (96,), generator=g).numpy()
    label_names = [{key: names[i]} for i in y]
    img_emb = ev.encode_image(batch)
    assert img_emb.shape == (96, 512) and isinstance(img_emb, np.ndarray)
    np.random.seed(7)
    res = ev.zeroshot_label_prompt(img_emb, label_names, classes, key, n_iterations=40)
    # oracle: fp32 embeddings of the same weights, same prompts (hashed stand-in ids), same metric code path
    prompts = label_prompts(key, classes)
    ptok = synthetic_prompt_tokens(prompts, 77, 3000)
    sd = {k: v.detach().cpu().float() for k, v in model.state_dict().items()}
    bsd = {k[len("text_encoder.model."):]: v for k, v in sd.items() if k.startswith("text_encoder.model.")}
    tf = O.eos_pool(E.bert_forward(bsd, ptok["input_ids"], ptok["attention_mask"], ptok["token_type_ids"]), ptok["attention_mask"])
    te = O.l2_normalize(O.linear_projection(tf, sd["text_projection_layer.layer.weight"])).numpy()
    ie = O.l2_normalize(O.linear_projection(batch["image_features"].cpu().flatten(1), sd["image_projection_layer.layer.weight"])).numpy()
    np.testing.assert_allclose(img_emb, ie, atol=2e-3)
    te_dev = ev.encode_text(prompts)
    # (1) the device embeddings and the scoring kernel against the fp32 oracle of the same weights (bf16 text tower: |logit| <= 14.3)
    np.testing.assert_allclose(ev.prompt_similarities(img_emb, prompts), (1 / 0.07) * ie @ te.T, atol=0.2)
    np.testing.assert_allclose(ev.prompt_similarities(ie, prompts), (1 / 0.07) * ie @ te_dev.T, atol=2e-4)     # kernel alone, tightly
    # (2) everything downstream of the logits (softmax, per-prompt AUROC / accuracy, bootstrap CI, accuracy, F1): the oracle's
    # restatement on the SAME logits and the same numpy RNG state must give the same numbers.  An untrained model scores at
    # chance, so near-ties are common: a last-bit difference between the device scoring kernel and a numpy matmul (checked to
    # 2e-4 in (1)) flips one ordered pair and moves an AUROC by 1/(n_pos n_neg) ~ 6e-4.  The device logits therefore enter the
    # oracle through an identity "text embedding" at scale 1 (x*1 + 0 is exact), which leaves only the metric code under test.
    sims_dev = np.asarray(ev.prompt_similarities(img_emb, prompts))
    np.random.seed(7)
    per, ci, acc, f1 = O.zeroshot_label_prompt(sims_dev, np.eye(len(prompts), dtype=sims_dev.dtype), 1.0, y, n_iterations=40)
    assert set(res) == set(prompts) | {"accuracy", "f1score"} | ({"auc_ci_mean", "auc_ci_lower", "auc_ci_higher"} if len(prompts) == 2 else set())
    for i, pr in enumerate(prompts):
        assert abs(res[pr]["auc"] - per[i][0]) < 1e-9 and abs(res[pr]["accuracy"] - per[i][1]) < 1e-9, (pr, res[pr], per[i])
    assert abs(res["accuracy"] - acc) < 1e-9 and abs(res["f1score"] - f1) < 1e-9
    if ci is not None:
        assert np.allclose([res["auc_ci_mean"], res["auc_ci_lower"], res["auc_ci_higher"]], ci, atol=1e-9)
        assert res["auc_ci_lower"] <= res["auc_ci_mean"] <= res["auc_ci_higher"]
