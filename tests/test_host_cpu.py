"""Host-side mirror classes that need no GPU: scheduler, early stopper, registries, config composer (vs golden vectors
generated from the reference's own files)."""
import json
import os

import pytest
import torch


def test_scheduler_class_matches_reference_golden(golden_dir):
    from mmgclip.scheduler.warmup_cosine import LinearWarmupCosineAnnealingLR
    ref = json.load(open(os.path.join(golden_dir, "g4_lr_schedule.json")))
    for key, lrs in ref.items():
        total, warm = key.split("_")
        total, warm = int(total), (float(warm) if "." in warm else int(warm))
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([p], lr=5e-5, weight_decay=1e-4)
        sc = LinearWarmupCosineAnnealingLR(opt, total, warm)
        mine = []
        for _ in range(total):
            mine.append(opt.param_groups[0]["lr"])
            opt.step()
            sc.step()
        assert mine == pytest.approx(lrs, rel=1e-12, abs=0)
    with pytest.raises(AssertionError):
        LinearWarmupCosineAnnealingLR(opt, 5, 5)


def test_early_stopper_class_matches_reference_golden(golden_dir, tmp_path):
    from mmgclip.callbacks.early_stopping import EarlyStopper
    ref = json.load(open(os.path.join(golden_dir, "g7_early_stopper.json")))
    model = torch.nn.Linear(2, 2)
    opt = torch.optim.AdamW(model.parameters())
    es = EarlyStopper(patience=5, trace_func=lambda *_: None)
    path = str(tmp_path / "model.pth")
    for epoch, v in enumerate(ref["trace"]):
        es(v, epoch, model, opt, path)
        st = ref["states"][epoch]
        assert (es.counter, es.best_score, es.early_stop, es.val_loss_min) == (st["counter"], st["best_score"], st["early_stop"], st["val_loss_min"])
    ckpt = torch.load(path, weights_only=False)
    assert sorted(ckpt.keys()) == ref["checkpoint_keys"] and ckpt["epoch"] == ref["last_saved_epoch"]


def test_registries_raise_like_the_reference():
    from mmgclip.experiments.experiments_controller import create_experiment
    from mmgclip.loss.loss_controller import create_loss
    from mmgclip.networks.network_controller import getNetworkClass
    from mmgclip.networks.projection_controller import get_projection_head
    assert create_loss("CLIPLoss").__name__ == "CLIPLoss"
    assert get_projection_head("MultiLinearHead").__name__ == "MultiLinearHead"
    assert getNetworkClass("BertEncoder").__name__ == "BertEncoder"
    assert getNetworkClass("ResNet50Encoder").__name__ == "ResNet50Encoder"
    assert create_experiment("classification").__name__ == "ClassifierExperiment"
    for fn, bad in ((create_loss, "X"), (get_projection_head, "ZeroProjection"), (getNetworkClass, "ConvNextTiny"), (create_experiment, "x")):
        with pytest.raises(ValueError, match="Invalid network_name"):
            fn(bad)


def test_config_composer_reproduces_the_reference_surface():
    from mmgclip.config import compose
    cfg_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mmg-clip_amd", "configs")
    c = compose(cfg_dir, "train_binary_class_clf")
    assert c.optimizer.config.learning_rate == 5e-5 and c.optimizer.config.weight_decay == 1e-4
    assert c.scheduler.name == "cosine" and c.scheduler.config.epochs == 30 and c.scheduler.config.warmup_epochs == 0.1
    assert c.networks.image_encoder.name == "ConvNextTiny" and c.networks.logit_temperature == 0.07
    assert c.networks.dropout.config.dropout == 0.5 and c.dataset.percentage.config.percentage == 1
    assert c.projection.config.projection_name == "LinearProjectionLayer" and c.tokenizer.config.sequence_length == 256
    assert c.experiments.config.metrics == ["BenignMalignantDatasetLabels"]          # ${dataset.config.enums_class}
    assert c.checkpoints.checkpoints_export_dir.startswith("outputs/") and c.checkpoints.checkpoints_export_dir.endswith("/checkpoints")
    r = compose(cfg_dir, "train_exam_reports_clf", ["loss=mmgclip", "optimizer.config.learning_rate=1e-3"])
    assert r.projection.config.output_projection_dimension == [768, 512] and r.loss.config.loss_name == "MMGCLIPLoss"
    assert r.base.features_export_dir == "outputs/dataset/reports_studies/4_avg" and r.optimizer.config.learning_rate == 1e-3
    rn = compose(cfg_dir, "train_binary_class_clf", ["networks=clip_resnet50_bert"])        # reference configs/networks/clip_resnet50_bert.yaml
    assert rn.networks.image_encoder.name == "ResNet50Encoder" and rn.networks.image_encoder.image_features_dimension == 2048


def test_product_path_fails_loudly_without_gpu():
    from mmgclip import head
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        head.L2Normalize.apply(torch.randn(4, 64))


def test_committed_bench_line_follows_the_driver_contract():
    """profiles/rNN_bench_default.json (every round's) is a line `bench.py` printed on an MI355X: every key of the bench contract is
    present, the roofline object is self-consistent and names a kernel of the committed rocprof summary of the same round."""
    import glob
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = sorted(glob.glob(os.path.join(root, "profiles", "r*_bench_default.json")))
    assert lines
    for path in lines:
        _check_bench_line(json.load(open(path)), path.replace("_bench_default.json", "_bench_default_kernel_stats.csv"))


def _check_bench_line(b, stats_path):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in b, k
    assert b["n_gpus"] == 1 and b["higher_is_better"] is True and b["scaling"] == "weak" and b["vs_baseline"] is None
    assert b["data"] == "synthetic" and b["dtype"] == "bf16" and "workload" in b["config"] and "model" not in b["config"]
    assert abs(b["value"] - b["config"]["global_batch"] / (b["ms_per_step"] / 1e3)) < 0.01 * b["value"]
    r = b["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] >= 0.9 * r["algorithmic_bytes_per_launch"]
    stats = open(stats_path).read()
    assert r["kernel"] in stats
    if "arithmetic_intensity_flop_per_byte" in r:      # round 2 on: the bound follows the arithmetic intensity, not the larger fraction
        assert (r["bound"] == "mfma") == (r["arithmetic_intensity_flop_per_byte"] >= r["ridge_flop_per_byte"])
        assert r["traffic"] is None or str(r.get("traffic_source", "")).startswith("static")
    c = b["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["unit"] == b["unit"]


def test_label_prompt_sentences_and_oracle_scoring():
    """evaluator.py:329-345 prompt sentences (enum names through process_class_list's wording map, data_utils.py:921-960) and the
    oracle's restatement of the scoring on a separable toy problem."""
    import numpy as np
    from mmgclip.evaluator import label_prompts, process_class_list
    from oracle import clip_oracle as O
    margins = {"unknown": 0, "circumscribed": 1, "obscured": 2, "spiculated": 3, "illdefined": 4}      # prompts/enums.py:38-43
    assert label_prompts("MassMarginLabels", margins) == ["Mass margin is unknown.", "Mass margin is circumscribed.", "Mass margin is obscured.",
                                                          "Mass margin is spiculated.", "Mass margin is ill defined."]
    assert label_prompts("BenignMalignantDatasetLabels", {"benign": 0, "malignant": 1}) == ["Finding suggesting benign.", "Finding suggesting malignant."]
    assert label_prompts("HasMassLabels", {"nomass": 0, "mass": 1}) == ["No mass was observed.", "Findings revealed a mass."]
    assert process_class_list(["nomass", "oval", "hascalcification"]) == ["no mass", "oval", "has calcification"]
    with pytest.raises(ValueError):
        process_class_list("oval")
    rng = np.random.default_rng(0)
    te = np.eye(2, 16, dtype=np.float32)
    y = rng.integers(0, 2, 64)
    ie = te[y] + 0.05 * rng.standard_normal((64, 16)).astype(np.float32)
    ie /= np.linalg.norm(ie, axis=1, keepdims=True)
    np.random.seed(0)
    per, ci, acc, f1 = O.zeroshot_label_prompt(ie, te, 1 / 0.07, y, n_iterations=50)
    assert acc == 1.0 and f1 == 1.0 and all(a == 1.0 for a, _ in per) and ci == (1.0, 1.0, 1.0)


def test_train_main_cli_composes_builds_loaders_and_constructs_the_experiment(tmp_path, monkeypatch):
    """`train.main([...])` - the reference's entry point (train.py:9-90) - from the command line to the constructed experiment:
    config name + group / key overrides parsed, seeded, the three synthetic loaders sized from the split ratios, the model / loss /
    optimizer / scheduler built.  The epochs themselves need the GPU (tests/test_experiment_gpu.py runs one through the same entry)."""
    import train
    from mmgclip.experiments import ClassifierExperiment as CE
    from mmgclip.networks import bert
    orig = bert.BertConfigLite.__init__

    def small(self, **kw):
        kw.setdefault("num_hidden_layers", 1)
        kw.setdefault("vocab_size", 2000)
        orig(self, **kw)
    monkeypatch.setattr(bert.BertConfigLite, "__init__", small)
    ran = []
    monkeypatch.setattr(CE.ClassifierExperiment, "run", lambda self: ran.append(self))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    exp = train.main(["--config-name", "train_prompt_clf", "networks=clip_convnexttiny_bert_pixels", "tokenizer=bert_clinical_seqlen=77",
                      "networks.image_encoder.image_size=64", "dataset.config.synthetic_samples=1000", "optimizer.config.learning_rate=0.001",
                      f"checkpoints.checkpoints_export_dir={tmp_path}/ckpt", f"base.tensorboard_export_dir={tmp_path}/tb",
                      f"base.results_export_dir={tmp_path}/results"])
    assert ran == [exp] and type(exp).__name__ == "ClassifierExperiment"
    cfg = exp.config
    assert cfg.experiments.config.experiment_name == "classification" and cfg.loss.config.loss_name == "CLIPLoss"
    assert cfg.networks.image_encoder.name == "ConvNextTinyEncoder" and cfg.tokenizer.config.sequence_length == 77
    assert exp.optimizer.param_groups[0]["initial_lr"] == 0.001 and exp.optimizer.param_groups[0]["lr"] == 0.0     # epoch 1 runs at lr 0
    # 1000 samples, 70 % train, half of the rest validation, batches of 64 (dataloader_64): 10 / 2 / 2 batches
    assert (len(exp.train_dataloader), len(exp.valid_dataloader), len(exp.test_dataloader)) == (10, 2, 2)
    b = next(iter(exp.test_dataloader))
    assert b["image"].shape == (64, 1, 64, 64) and b["text_tokens"]["input_ids"].shape == (64, 77) and len(b["prompt_labels"]) == 64
    assert type(exp.model).__name__ == "MMGCLIP" and type(exp.criterion).__name__ == "CLIPLoss"
    assert torch.initial_seed() == cfg.base.seed == 42
    with pytest.raises(ValueError, match="Invalid"):
        train.main(["--config-name", "train_prompt_clf", "experiments.config.experiment_name=nope"])
