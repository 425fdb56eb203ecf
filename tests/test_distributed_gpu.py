"""Two ranks sharing the one MI355X of the test box (gloo transport, HIP kernels): the global-batch training step of
P = 2 ranks on halves of a batch must give the loss and the (summed) parameter gradients of one rank on the whole batch."""
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
OVERRIDES = ["networks=clip_convnexttiny_bert_pixels", "tokenizer=bert_clinical_seqlen=77", "networks/dropout=dropout0",
             "networks.image_encoder.micro_batch=4", "networks.image_encoder.image_size=64"]


def _build_and_step(comm, batch, cfg_name="train_binary_class_clf", overrides=OVERRIDES, overlap=None, want_report=False):
    from mmgclip import distributed
    from mmgclip.config import compose
    from mmgclip.loss.loss_controller import create_loss
    from mmgclip.networks import bert
    from mmgclip.networks.mmgclip_model import MMGCLIP
    orig = bert.BertConfigLite.__init__

    def small(self, **kw):
        kw.setdefault("num_hidden_layers", 2)
        kw.setdefault("vocab_size", 3000)
        orig(self, **kw)
    bert.BertConfigLite.__init__ = small
    try:
        torch.manual_seed(0)
        cfg = compose(CFG_DIR, cfg_name, overrides)
        model = MMGCLIP(cfg).train()
    finally:
        bert.BertConfigLite.__init__ = orig
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("layer_scale"):
                p.fill_(0.5)
    crit = create_loss(cfg.loss.config.loss_name)(comm=comm)
    out = model(batch, materialize_logits=False)
    loss, _ = crit(**out)
    extra = [p for n, p in model.named_parameters() if not (n.startswith("image_encoder.") or n.startswith("text_encoder."))]
    sync = distributed.GradSync(comm, [model.image_encoder.arena, model.text_encoder.arena], extra, overlap=overlap, timing=want_report)
    if comm is not None:
        model.image_encoder.post_backward_hook = sync.reduce_arena_async
        model.text_encoder.post_backward_hook = sync.reduce_arena_async
    loss.backward()
    sync.finish()
    torch.cuda.synchronize()
    g_img = model.image_encoder.arena.grad.clone()
    g_txt = model.text_encoder.arena.grad.clone()
    g_proj = model.image_projection_layer.layer.weight.grad.clone()
    if want_report:
        return loss.item(), g_img.cpu(), g_txt.cpu(), g_proj.cpu(), {"sync": sync.report(), "comm": comm.report() if comm else None,
                                                                     "cfg": {"loss": cfg.loss.config.loss_name, "dropout": cfg.networks.dropout.config.dropout,
                                                                             "batch_size": cfg.dataloader.train.batch_size, "dataset": cfg.dataset.name if hasattr(cfg.dataset, "name") else None}}
    return loss.item(), g_img.cpu(), g_txt.cpu(), g_proj.cpu()


def _slice_batch(batch, sl):
    from mmgclip.dataset.synthetic import TokenBatch
    return {"image": batch["image"][sl].clone(), "text_tokens": TokenBatch({k: v[sl].clone() for k, v in batch["text_tokens"].items()})}


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "mmg-clip_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from mmgclip import distributed
    from mmgclip.dataset.synthetic import synthetic_batch
    comm = distributed.init_from_env("gloo")
    batch = synthetic_batch(8, S=77, image_size=64, vocab_size=3000, seed=11)
    n = 8 // world
    loss, gi, gt, gp = _build_and_step(comm, _slice_batch(batch, slice(rank * n, (rank + 1) * n)))
    q.put((rank, loss, gi.numpy(), gt.numpy(), gp.numpy()))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_ranks_equal_one_rank_on_the_whole_batch(dev):
    from mmgclip.dataset.synthetic import synthetic_batch
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=300) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    batch = synthetic_batch(8, S=77, image_size=64, vocab_size=3000, seed=11)
    loss, gi, gt, gp = _build_and_step(None, _slice_batch(batch, slice(0, 8)))
    for rank, l2, gi2, gt2, gp2 in results:
        assert abs(l2 - loss) < 2e-3 * abs(loss), (rank, l2, loss)               # bf16 towers see different micro-batches
        for name, a, b in (("convnext", gi2, gi.numpy()), ("bert", gt2, gt.numpy()), ("proj", gp2, gp.numpy())):
            cos = float(np.dot(a.ravel(), b.ravel()) / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
            rel = float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))
            assert cos > 0.999 and rel < 5e-2, (rank, name, cos, rel)
    # both ranks hold identical reduced gradients
    assert np.array_equal(results[0][2], results[1][2]) and np.array_equal(results[0][3], results[1][3])


def _rccl_worker(port, q):
    for p in (ROOT, os.path.join(ROOT, "mmg-clip_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    from mmgclip import distributed
    from mmgclip.dataset.synthetic import synthetic_batch
    try:
        comm = distributed.init_from_env("nccl", single_rank=True)
        assert comm.active and comm.world_size == 1 and torch.distributed.get_backend() == "nccl"
        batch = synthetic_batch(8, S=77, image_size=64, vocab_size=3000, seed=11)
        with_comm = _build_and_step(comm, _slice_batch(batch, slice(0, 8)))
        plain = _build_and_step(None, _slice_batch(batch, slice(0, 8)))
        # the raw exchanges, on the dtypes and shapes the step uses
        rows = torch.randn(8, 512, device="cuda")
        assert torch.equal(comm.all_gather_rows(rows), rows)
        flat = torch.randn(1 << 22, device="cuda")
        ref = flat.clone()
        assert torch.equal(comm.all_reduce_sum(flat), ref)
        torch.cuda.synchronize()
        q.put(("ok", with_comm[0], plain[0], [float((a - b).abs().max() / (b.abs().max() + 1e-30)) for a, b in zip(with_comm[1:], plain[1:])]))
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    except Exception as e:           # report instead of hanging the parent on q.get
        import traceback
        q.put(("error", traceback.format_exc(), repr(e), None))


def test_rccl_one_rank_group_runs_every_exchange(dev):
    """backend "nccl" (= RCCL) with one rank on the one GPU: the embedding / LSE all-gathers, the loss all-reduce and the
    side-stream arena all-reduces all go through RCCL and must leave the step's loss and gradients unchanged."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    p = ctx.Process(target=_rccl_worker, args=(port, q))
    p.start()
    status, l_comm, l_plain, rels = q.get(timeout=300)
    p.join(timeout=120)
    assert status == "ok", l_comm
    assert p.exitcode == 0
    assert abs(l_comm - l_plain) < 1e-5 * abs(l_plain), (l_comm, l_plain)
    assert max(rels) < 1e-3, rels          # atomics in the weight-gradient reductions reorder fp32 sums


# ---- BASELINE config C3: train_prompt_clf, 2 ranks, global batch through the embedding all-gather ------------
C3_OVERRIDES = ["networks=clip_convnexttiny_bert_pixels", "tokenizer=bert_clinical_seqlen=77",
                "networks.image_encoder.micro_batch=4", "networks.image_encoder.image_size=64"]     # dropout 0.2 etc. as shipped


def _c3_worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "mmg-clip_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from mmgclip import distributed
    from mmgclip.dataset.synthetic import synthetic_batch
    try:
        comm = distributed.init_from_env("gloo")
        batch = synthetic_batch(8, S=77, image_size=64, vocab_size=3000, seed=11)
        n = 8 // world
        mine = lambda: _slice_batch(batch, slice(rank * n, (rank + 1) * n))      # noqa: E731
        over = _build_and_step(comm, mine(), "train_prompt_clf", C3_OVERRIDES, overlap=True, want_report=True)
        after = _build_and_step(comm, mine(), "train_prompt_clf", C3_OVERRIDES, overlap=False, want_report=True)
        q.put((rank, "ok", [over[0], over[1].numpy(), over[2].numpy(), over[3].numpy(), over[4]],
               [after[0], after[1].numpy(), after[2].numpy(), after[3].numpy(), after[4]]))
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, "error", traceback.format_exc(), None))


def test_config_c3_two_ranks_global_batch_and_both_gradient_sync_modes(dev):
    """`train_prompt_clf` composed as shipped (CLIPLoss, dropout 0.2 group, dataloader_64) with the pixel towers, 2 ranks on halves of
    a batch: the global-batch loss and the summed gradients equal the one-rank step on the whole batch; the bucketed / overlapped
    gradient all-reduce and the MMG_GRAD_OVERLAP=0 fallback (one all-reduce per arena after the backward) give the same gradients,
    and the `comm` report that bench.py prints at N > 1 is populated."""
    from mmgclip.dataset.synthetic import synthetic_batch
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = [ctx.Process(target=_c3_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=400) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in results:
        assert r[1] == "ok", r[2]
    batch = synthetic_batch(8, S=77, image_size=64, vocab_size=3000, seed=11)
    loss, gi, gt, gp = _build_and_step(None, _slice_batch(batch, slice(0, 8)), "train_prompt_clf", C3_OVERRIDES)
    for rank, _, over, after in results:
        rep = over[4]
        assert (rep["cfg"]["loss"], rep["cfg"]["dropout"], rep["cfg"]["batch_size"]) == ("CLIPLoss", 0.2, 64), rep["cfg"]
        for l2, gi2, gt2, gp2, _ in (over, after):
            assert abs(l2 - loss) < 2e-3 * abs(loss), (rank, l2, loss)
            for name, a, b in (("convnext", gi2, gi.numpy()), ("bert", gt2, gt.numpy()), ("proj", gp2, gp.numpy())):
                cos = float(np.dot(a.ravel(), b.ravel()) / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
                rel = float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))
                assert cos > 0.999 and rel < 5e-2, (rank, name, cos, rel)
        # the two synchronisation modes reduce the same numbers (the towers' weight-gradient atomics reorder fp32 sums between two runs)
        for a, b in zip(over[1:4], after[1:4]):
            assert float(np.abs(a - b).max()) <= 1e-3 * float(np.abs(b).max()) + 1e-12
        so, sa = over[4]["sync"], after[4]["sync"]
        n_grad = gi.numel() + gt.numel() + gp.numel() * 2
        assert so["mode"].startswith("bucketed") and sa["mode"].startswith("one all-reduce per arena")
        assert so["grad_allreduce_bytes_per_step"] == sa["grad_allreduce_bytes_per_step"] >= 4 * (gi.numel() + gt.numel())
        assert so["grad_allreduce_bytes_per_step"] <= 4 * n_grad + 64
        assert sa["grad_allreduce_calls_per_step"] == 3.0 and so["grad_allreduce_calls_per_step"] >= 3.0     # two arenas + the heads
        assert so["allreduce_busy_ms_per_step"] > 0 and so["exposed_wait_ms_per_step"] >= 0 and sa["exposed_wait_ms_per_step"] > 0
        c = over[4]["comm"]
        assert c["all_gather_calls"] == 4 and c["all_gather_bytes"] == 2 * (2 * 4 * 512 * 4) + 2 * (2 * 4 * 4)   # embeddings + LSE vectors
    assert np.array_equal(results[0][2][1], results[1][2][1]) and np.array_equal(results[0][3][2], results[1][3][2])


def test_bench_two_ranks_prints_one_compact_line_with_comm(dev, tmp_path):
    """bench.py exactly as the driver launches it at N = 2 (torch.distributed.run, one rank per process; gloo transport because the
    test box has one GPU, which both ranks share): rank 0 prints ONE JSON line below 4 KB that parses, carries the contract fields
    for N = 2 and a populated `comm` block; the full per-kernel record goes to --detail-out."""
    import json
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    detail = str(tmp_path / "bench_detail.json")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8",
           "--image-size", "224", "--micro-batch", "8", "--backend", "gloo", "--profile-steps", "1", "--detail-out", detail]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    assert len(lines[0]) < 4096
    d = json.loads(lines[0])
    assert (d["n_gpus"], d["steps"], d["warmup"], d["scaling"], d["higher_is_better"]) == (2, 2, 1, "weak", True)
    assert d["value"] > 0 and abs(d["value"] - 16 * 1000.0 / d["ms_per_step"]) < 0.02 * d["value"]
    assert d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp2" and d["config"]["loss_scope"].startswith("global")
    c = d["comm"]
    assert c["mode"].startswith("bucketed") and c["grad_allreduce_bytes_per_step"] > 4 * 100e6      # both towers' fp32 gradients
    assert c["grad_allreduce_calls_per_step"] >= 3 and c["allreduce_busy_ms_per_step"] > 0 and c["exposed_wait_ms_per_step"] >= 0
    assert c["all_gather_calls_per_step"] == 4 and c["all_gather_bytes_per_step"] > 0
    assert d["roofline"]["kernel"] and 0 <= d["roofline"]["frac"] < 1 and d["roofline"]["avg_launch_us"] > 0    # (8 pairs per rank: a tiny fraction)
    assert "cpu_baseline" not in d                                                                               # (the CPU leg is N = 1 only)
    full = json.load(open(detail))
    assert full["value"] == d["value"] and len(full.get("roofline_other_kernels", [])) >= 5
    assert "AccumulateGrad node's stream" not in r.stderr
