"""ClassifierExperiment on the real towers: the optimizer paths and the data-parallel loop.

  * `optimizer.config.fused=true` built BEFORE the first forward (the reference's construction order): the towers really train
    and follow torch.optim.AdamW step for step;
  * FusedAdamW state_dict: torch.optim.AdamW's layout, save -> load -> step round trip, interchangeable with torch's;
  * 2 ranks on the one GPU (gloo) through create_experiment(...).train() with MMGCLIPLoss (text tower runs twice per step):
    identical parameters on both ranks, equal to the 1-rank run on the whole batches.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG_DIR = os.path.join(ROOT, "mmg-clip_amd", "configs")
PIXELS = ["networks=clip_convnexttiny_bert_pixels", "tokenizer=bert_clinical_seqlen=77", "networks/dropout=dropout0",
          "networks.image_encoder.micro_batch=4", "networks.image_encoder.image_size=64", "scheduler=warmup1_epo15"]


def _small_bert():
    from mmgclip.networks import bert
    orig = bert.BertConfigLite.__init__

    def small(self, **kw):
        kw.setdefault("num_hidden_layers", 2)
        kw.setdefault("vocab_size", 3000)
        orig(self, **kw)
    bert.BertConfigLite.__init__ = small
    return lambda: setattr(bert.BertConfigLite, "__init__", orig)


def _experiment(tmp, overrides, loader, comm=None, seed=0):
    from mmgclip.config import compose
    from mmgclip.experiments.experiments_controller import create_experiment
    cfg = compose(CFG_DIR, "train_binary_class_clf", PIXELS + [f"checkpoints.checkpoints_export_dir={tmp}/ckpt",
                                                               f"base.tensorboard_export_dir={tmp}/tb"] + list(overrides))
    restore = _small_bert()
    try:
        torch.manual_seed(seed)
        exp = create_experiment("classification")(config=cfg, train_dataloader=loader, valid_dataloader=None, test_dataloader=None,
                                                  tokenizer=None, comm=comm)
    finally:
        restore()
    with torch.no_grad():
        for n, p in exp.model.named_parameters():
            if n.endswith("layer_scale"):
                p.fill_(0.5)                   # make the ConvNeXt blocks matter (1e-6 at init)
    return exp


def _loader(steps=2, n=8, seed=5, **kw):
    from mmgclip.dataset.synthetic import SyntheticLoader
    return SyntheticLoader(steps, n, seed=seed, S=77, image_size=64, vocab_size=3000, **kw)


def _weights(exp):
    return {k: v.detach().float().cpu().clone() for k, v in exp.model.state_dict().items() if v.is_floating_point()}


def _delta_agreement(wa, wb, w0):
    """cosine and norm ratio of the two runs' total parameter movement (after - before, every tensor concatenated).
    Key biases are left out: their gradient is analytically zero (softmax is invariant to a per-query constant), what arrives
    is rounding noise, and Adam turns noise of any size into +-lr steps."""
    keys = [k for k in w0 if not k.endswith(".key.bias") and not k.endswith("in_proj_bias")]
    da = torch.cat([(wa[k] - w0[k]).flatten() for k in keys]).double()
    db = torch.cat([(wb[k] - w0[k]).flatten() for k in keys]).double()
    return float(da @ db / (da.norm() * db.norm() + 1e-30)), float(da.norm() / (db.norm() + 1e-30))


def _probe(exp):
    """Tower outputs on a fixed batch (what the kernels compute from their bf16 working copies)."""
    from mmgclip.dataset.synthetic import synthetic_batch
    b = synthetic_batch(4, S=77, image_size=64, vocab_size=3000, seed=99)
    exp.model.eval()
    with torch.no_grad():
        out = exp.model(b)
    exp.model.train()
    return torch.cat([out["image_embeddings"], out["text_embeddings"]]).float().cpu()


def test_fused_adamw_built_before_first_forward_trains_the_towers(dev, tmp_path):
    """FusedAdamW is constructed in ClassifierExperiment.__init__, when no arena exists yet.  The towers'
    kernels must see every update (arena version bumps), and the path must follow torch.optim.AdamW."""
    runs = {}
    for name, over in (("fused", ["optimizer.config.fused=true"]), ("torch", [])):
        exp = _experiment(str(tmp_path / name), over + ["optimizer.config.learning_rate=5e-4"], _loader())
        assert type(exp.optimizer).__name__ == ("FusedAdamW" if name == "fused" else "AdamW")
        p0, w0 = _probe(exp), _weights(exp)
        exp.scheduler.step()                      # leave the reference's lr-0 first epoch
        exp.scheduler.step()
        losses = [exp.train() for _ in range(3)]
        p1 = _probe(exp)
        runs[name] = (p0, p1, losses, _weights(exp), w0)
        if name == "fused":
            ia, ta = exp.model.image_encoder.arena, exp.model.text_encoder.arena
            assert all("exp_avg" in exp.optimizer.state[p] for p in ia.params + ta.params)
            assert exp.optimizer.state[ia.params[0]]["exp_avg"].data_ptr() == exp.optimizer._flat[id(ia)]["m"].data_ptr()
            assert int(exp.optimizer.state[ia.params[3]]["step"]) == 6          # 3 epochs x 2 steps, one launch per tower per step
            assert len({v["step"].data_ptr() for v in exp.optimizer.state_dict()["state"].values()}) == len(exp.optimizer.state)
    for name, (p0, p1, losses, _, _) in runs.items():
        assert (p1 - p0).abs().max() > 3e-3, name             # the forward sees the trained weights (both towers' embeddings moved)
        assert min(losses[1:]) < losses[0], (name, losses)    # (6 steps on 2 alternating batches: the curve is not monotonic)
    assert torch.allclose(runs["fused"][0], runs["torch"][0])
    # same trajectory as torch.optim.AdamW (bf16 kernels under both: only the optimizer arithmetic differs, fp32 both sides)
    cos, ratio = _delta_agreement(runs["fused"][3], runs["torch"][3], runs["torch"][4])
    print(f"fused vs torch AdamW after 6 steps: movement cosine {cos:.4f}, norm ratio {ratio:.4f}, losses {runs['fused'][2]} / {runs['torch'][2]}")
    # (fp32 atomics reorder the weight-gradient sums from run to run, and Adam turns every gradient into a +-lr step: elements whose
    # gradient is at the noise floor move in either direction; the bulk of the movement and the loss curve must agree)
    assert np.allclose(runs["fused"][2], runs["torch"][2], rtol=1e-2), (runs["fused"][2], runs["torch"][2])
    assert cos > 0.9 and 0.95 < ratio < 1.05, (cos, ratio)


def test_fused_adamw_state_dict_round_trip_and_torch_interchange(dev, tmp_path):
    """the checkpoint's optimizer_state_dict holds the towers' moments in torch.optim.AdamW's layout."""
    from mmgclip.optim import FusedAdamW
    exp = _experiment(str(tmp_path / "a"), ["optimizer.config.fused=true", "optimizer.config.learning_rate=1e-3"], _loader(steps=2))
    exp.scheduler.step()
    exp.train()
    sd = exp.optimizer.state_dict()
    n_params = sum(1 for p in exp.model.parameters() if p.requires_grad)
    assert len(sd["state"]) == n_params and all(set(v) == {"step", "exp_avg", "exp_avg_sq"} for v in sd["state"].values())
    assert all(float(v["step"]) == 2.0 for v in sd["state"].values())
    # a torch.optim.AdamW accepts it (same parameter order) and continues identically to a re-loaded FusedAdamW
    path = str(tmp_path / "opt.pth")
    torch.save({"opt": sd, "model": exp.model.state_dict()}, path)
    ck = torch.load(path, weights_only=False)
    batch = next(iter(_loader(steps=1, seed=77)))
    w0 = {k: v.detach().float().cpu().clone() for k, v in ck["model"].items() if v.is_floating_point()}
    results = {}
    for kind in ("fused", "torch"):
        e2 = _experiment(str(tmp_path / kind), ["optimizer.config.fused=true" if kind == "fused" else "optimizer.config.fused=false",
                                                "optimizer.config.learning_rate=1e-3"], _loader(steps=1))
        e2.scheduler.step()                                 # (past the lr-0 first epoch, like the run that wrote the checkpoint)
        e2.model.load_state_dict(ck["model"])
        for t in (e2.model.image_encoder, e2.model.text_encoder):
            if t.arena is not None:
                t.arena.touch()
        e2.optimizer.load_state_dict(ck["opt"])
        assert isinstance(e2.optimizer, FusedAdamW) == (kind == "fused")
        b = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        loss, _ = e2.criterion(**e2.model(b))
        loss.backward()
        e2.optimizer.step()
        st = e2.optimizer.state[e2.model.image_encoder.arena.params[0]]
        assert float(st["step"]) == 3.0
        results[kind] = _weights(e2)
    cos, ratio = _delta_agreement(results["fused"], results["torch"], w0)
    assert cos > 0.99 and 0.98 < ratio < 1.02, (cos, ratio)     # the third step, with the loaded moments, is the same step
    # without the loaded moments the step would be a first Adam step (every element moves by ~lr): far from this one
    fresh = _experiment(str(tmp_path / "fresh"), ["optimizer.config.fused=true", "optimizer.config.learning_rate=1e-3"], _loader(steps=1))
    fresh.scheduler.step()
    fresh.model.load_state_dict(ck["model"])
    b = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
    loss, _ = fresh.criterion(**fresh.model(b))
    loss.backward()
    fresh.optimizer.step()
    cos_fresh, _ = _delta_agreement(_weights(fresh), results["torch"], w0)
    assert cos_fresh < 0.9, cos_fresh


# ---- two ranks on the one GPU ------------------------------------------------------------------------------------------------------
class _HalfLoader:
    """Rank r's half of every batch of a SyntheticLoader (so that 2 ranks together see exactly the 1-rank batches)."""

    def __init__(self, loader, rank, world):
        self.loader, self.rank, self.world = loader, rank, world

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        from mmgclip.dataset.synthetic import TokenBatch
        for b in self.loader:
            n = b["image"].shape[0] // self.world
            sl = slice(self.rank * n, (self.rank + 1) * n)
            out = {"image": b["image"][sl].clone()}
            for k in ("text_tokens", "image_impression_tokens"):
                if k in b:
                    out[k] = TokenBatch({kk: v[sl].clone() for kk, v in b[k].items()})
            yield out


def _train_worker(rank, world, port, tmp, loss_name, q):
    for p in (ROOT, os.path.join(ROOT, "mmg-clip_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    try:
        from mmgclip import distributed
        comm = distributed.init_from_env("gloo")
        res = _train_run(comm, rank, world, f"{tmp}/r{rank}", loss_name)
        q.put((rank, "ok", res))
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, "error", traceback.format_exc()))


def _train_run(comm, rank, world, tmp, loss_name):
    os.environ["MMG_BUCKET_MB"] = "8"          # several buckets per tower arena (ConvNeXt-T: 111 MB) in this small run
    full = _loader(steps=2, n=8, seed=31, with_impression=loss_name == "MMGCLIPLoss")
    exp = _experiment(tmp, ["optimizer.config.fused=true", "optimizer.config.learning_rate=1e-3", f"loss.config.loss_name={loss_name}"],
                      _HalfLoader(full, rank, world) if world > 1 else full, comm=comm)
    w0 = _weights(exp)
    exp.scheduler.step()
    losses = [exp.train(), exp.train()]
    torch.cuda.synchronize()
    sync = getattr(exp, "_sync", None)
    ncoll = len(sync.last_log) if (sync is not None and world > 1) else 0      # all-reduces of the last step's tower gradients
    return losses, {k: v.numpy() for k, v in _weights(exp).items()}, {k: v.numpy() for k, v in w0.items()}, ncoll


@pytest.mark.parametrize("loss_name", ["CLIPLoss", "MMGCLIPLoss"])
def test_two_rank_experiment_train_equals_one_rank(dev, tmp_path, loss_name):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, str(tmp_path), loss_name, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=420) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
    assert all(r[1] == "ok" for r in results), [r[2] for r in results if r[1] != "ok"]
    assert all(p.exitcode == 0 for p in procs)
    (l0, sd0, _, nc0), (l1, sd1, _, nc1) = results[0][2], results[1][2]
    assert nc0 == nc1 and nc0 > 2, (nc0, nc1)                     # bucketed: more collectives than tower arenas, same on both ranks
    for k in sd0:
        assert np.array_equal(sd0[k], sd1[k]), k                  # the replicas are bit-identical after 4 optimizer steps
    assert np.allclose(l0, l1, rtol=1e-6)                         # global loss: the same number on both ranks
    l_ref, ref, w0, _ = _train_run(None, 0, 1, str(tmp_path / "one"), loss_name)
    assert np.allclose(l0, l_ref, rtol=5e-3), (l0, l_ref)         # bf16 towers see different micro-batches
    t = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}    # noqa: E731
    cos, ratio = _delta_agreement(t(sd0), t(ref), t(w0))
    assert cos > 0.9 and 0.9 < ratio < 1.1, (cos, ratio)          # 4 Adam steps: same movement as the 1-rank run on whole batches


def test_fused_adamw_loads_a_reference_style_state_dict_with_frozen_bert(dev, tmp_path):
    """the reference builds torch.optim.AdamW(model.parameters()) with the frozen BERT listed
    (/root/reference/mmgclip/experiments/ClassifierExperiment.py:74).  A state dict written that way loads into the fused optimizer
    of the same model (and the other way round): same param-group sizes, state only for the parameters that trained."""
    frozen = ["networks.text_encoder.freeze=true", "optimizer.config.learning_rate=1e-3"]
    ref = _experiment(str(tmp_path / "ref"), frozen + ["optimizer.config.fused=false"], _loader(steps=2))
    assert type(ref.optimizer).__name__ == "AdamW"
    ref.scheduler.step()
    ref.train()
    sd = ref.optimizer.state_dict()
    n_all = sum(1 for _ in ref.model.parameters())
    n_train = sum(1 for p in ref.model.parameters() if p.requires_grad)
    assert n_train < n_all and len(sd["param_groups"][0]["params"]) == n_all and len(sd["state"]) == n_train
    fused = _experiment(str(tmp_path / "fused"), frozen + ["optimizer.config.fused=true"], _loader(steps=1))
    assert type(fused.optimizer).__name__ == "FusedAdamW"
    assert len(fused.optimizer.state_dict()["param_groups"][0]["params"]) == n_all
    fused.model.load_state_dict(ref.model.state_dict())
    fused.optimizer.load_state_dict(sd)                    # raised "size of parameter group" with the filtered list of round 2
    fused.scheduler.step()
    fused.train()
    sd2 = fused.optimizer.state_dict()
    assert len(sd2["state"]) == n_train and all(float(v["step"]) == 3.0 for v in sd2["state"].values())
    ref.optimizer.load_state_dict(sd2)                     # and back


def test_train_main_runs_one_tiny_epoch_and_the_end_of_run_test_pass(dev, tmp_path, monkeypatch):
    """The whole drop-in chain through `train.main([...])` (reference train.py:9-90 -> ClassifierExperiment.run -> test ->
    Evaluator.evaluate_experiment, ClassifierExperiment.py:291-301,337-339, evaluator.py:564-654) on synthetic loaders: two epochs (the
    first one at lr 0, as the reference's schedule has it),
    a checkpoint, and results.txt from the zero-shot label-prompt scorer over the test split."""
    import train
    restore = _small_bert()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    try:
        exp = train.main(["--config-name", "train_binary_class_clf"] + PIXELS[:-1] + [
            "scheduler=warmup1_epo15", "scheduler.config.epochs=2", "dataset.config.synthetic_samples=64", "dataloader=dataloader_32",
            "optimizer.config.fused=true", f"checkpoints.checkpoints_export_dir={tmp_path}/ckpt",
            f"base.tensorboard_export_dir={tmp_path}/tb", f"base.results_export_dir={tmp_path}/results"])
    finally:
        restore()
    assert exp.current_epoch == 1 and os.path.isfile(os.path.join(str(tmp_path), "ckpt", "model.pth"))
    assert list(exp.config.dataset.eval.enum_classes) == ["BenignMalignantDatasetLabels"]
    (res,) = exp.test_results                       # one enum class x the one scorer this build has ("ova" / "confustion_matrix" skipped)
    assert set(res) == {"Finding suggesting benign.", "Finding suggesting malignant.", "auc_ci_mean", "auc_ci_lower", "auc_ci_higher",
                        "accuracy", "f1score"}
    assert 0.0 <= res["Finding suggesting malignant."]["auc"] <= 1.0 and res["auc_ci_lower"] <= res["auc_ci_mean"] <= res["auc_ci_higher"]
    text = open(os.path.join(str(tmp_path), "results", "results.txt")).read()
    assert "Finding suggesting malignant." in text and "f1score" in text
