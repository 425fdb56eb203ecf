#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING the reference's own files.

Run in the build container only (it reads /root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

Reference modules are loaded BY FILE PATH (the `mmgclip` package itself cannot be imported here: its
__init__ chain needs torchvision, nltk, ... which are absent and not installable offline):
    mmgclip/networks/projection.py, mmgclip/loss/losses.py, mmgclip/scheduler/warmup_cosine.py,
    mmgclip/callbacks/early_stopping.py
`losses.py` imports `sentence_transformers.util.cos_sim` (absent): a 3-line stand-in with the published
definition (normalise + matmul) is placed in sys.modules for the import; `Tensor.cuda` is neutralised for the
call because losses.py:39,78 hard-code `.cuda()` and this container has no GPU.

Only arrays (inputs + expected outputs) are written; no reference source text is copied.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("MMGCLIP_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
if OUT not in sys.path:
    sys.path.insert(0, OUT)


def load_by_path(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def install_cos_sim_stub():
    st = types.ModuleType("sentence_transformers")
    util = types.ModuleType("sentence_transformers.util")

    def cos_sim(a, b):
        a = torch.nn.functional.normalize(a, p=2, dim=1)
        b = torch.nn.functional.normalize(b, p=2, dim=1)
        return a @ b.t()

    util.cos_sim = cos_sim
    st.util = util
    sys.modules["sentence_transformers"] = st
    sys.modules["sentence_transformers.util"] = util


def t2n(t):
    return t.detach().cpu().numpy()


def golden_projection(proj):
    """G1: projection heads, forward + gradients, seeded inputs."""
    out = {}
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(8, 96, generator=g)      # small dims keep the fixture small; the arithmetic is size-independent
    gy = torch.randn(8, 64, generator=g)
    out["x"] = t2n(x)
    out["gy"] = t2n(gy)

    torch.manual_seed(7)
    lin = proj.LinearProjectionLayer(embedding_dim=96, projection_dim=64, dropout=0.5)
    xx = x.clone().requires_grad_(True)
    y = lin(xx)
    y.backward(gy)
    out["linear.weight"] = t2n(lin.layer.weight)
    out["linear.y"] = t2n(y)
    out["linear.dx"] = t2n(xx.grad)
    out["linear.dweight"] = t2n(lin.layer.weight.grad)

    torch.manual_seed(8)
    ml = proj.MultiLinearHead(embedding_dim=96, projection_dim=[96, 64], dropout=0.0)
    xx = x.clone().requires_grad_(True)
    y = ml(xx)
    y.backward(gy)
    for i, layer in enumerate(ml.layers):
        out[f"multi.layers.{i}.weight"] = t2n(layer.weight)
        out[f"multi.layers.{i}.bias"] = t2n(layer.bias)
        out[f"multi.layers.{i}.dweight"] = t2n(layer.weight.grad)
        out[f"multi.layers.{i}.dbias"] = t2n(layer.bias.grad)
    out["multi.y"] = t2n(y)
    out["multi.dx"] = t2n(xx.grad)

    torch.manual_seed(9)
    mlp = proj.MLPProjectionHead(embedding_dim=96, projection_dim=64, dropout=0.0)
    xx = x.clone().requires_grad_(True)
    y = mlp(xx)
    y.backward(gy)
    for k, v in mlp.state_dict().items():
        out[f"mlp.{k}"] = t2n(v)
    for k, p in mlp.named_parameters():
        out[f"mlp.grad.{k}"] = t2n(p.grad)
    out["mlp.y"] = t2n(y)
    out["mlp.dx"] = t2n(xx.grad)
    np.savez_compressed(os.path.join(OUT, "g1_projection.npz"), **out)


def reference_forward_tail(img, txt, logit_scale_param):
    """The arithmetic of mmgclip/networks/mmgclip_model.py:128-136 applied to given projection outputs.

    MMGCLIP itself cannot be imported here (relative imports + absent dependencies), so these five lines are
    re-typed from the text of that file; the LOSS that consumes them is the reference's own class.
    """
    image_embeddings = img / img.norm(dim=1, keepdim=True)
    text_embeddings = txt / txt.norm(dim=1, keepdim=True)
    logit_scale = logit_scale_param.exp()
    logits_per_image = logit_scale * image_embeddings @ text_embeddings.t()
    logits_per_text = logit_scale * text_embeddings @ image_embeddings.t()
    return image_embeddings, text_embeddings, logit_scale, logits_per_image, logits_per_text


def golden_head(losses):
    """G2: contrastive head for n in {8, 32, 256} (+ a ragged n=37): embeddings, logits, CLIPLoss, MMGCLIPLoss, grads."""
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        for n in (8, 32, 37, 256):
            out = {}
            g = torch.Generator().manual_seed(100 + n)
            D = 128 if n == 256 else 512   # the big case uses D=128 to keep the fixture ~1 MB
            # projection outputs with realistic spread: a shared component + per-sample noise
            base = torch.randn(1, D, generator=g)
            img = (0.5 * base + torch.randn(n, D, generator=g)).requires_grad_(True)
            txt = (0.5 * base + 0.7 * img.detach() + torch.randn(n, D, generator=g)).requires_grad_(True)
            txt2 = (0.3 * base + 0.5 * txt.detach() + torch.randn(n, D, generator=g)).requires_grad_(True)
            ls = torch.tensor(float(np.log(1 / 0.07)), requires_grad=True)
            ie, te, s, li, lt = reference_forward_tail(img, txt, ls)
            loss, labels = losses.CLIPLoss()(logits_per_image=li, logits_per_text=lt, image_embeddings=ie,
                                             text_embeddings=te, logit_scale=s)
            loss.backward()
            out.update(img=t2n(img), txt=t2n(txt), txt2=t2n(txt2), logit_scale_param=t2n(ls),
                       image_embeddings=t2n(ie), text_embeddings=t2n(te), scale=t2n(s),
                       logits_per_image=t2n(li), logits_per_text=t2n(lt), clip_loss=t2n(loss),
                       clip_labels=t2n(labels), clip_dimg=t2n(img.grad), clip_dtxt=t2n(txt.grad),
                       clip_dlogit_scale=t2n(ls.grad))
            # MMGCLIPLoss (second text view)
            for t in (img, txt, txt2, ls):
                t.grad = None
            ie, te, s, li, lt = reference_forward_tail(img, txt, ls)
            te2 = txt2 / txt2.norm(dim=1, keepdim=True)
            loss2, _ = losses.MMGCLIPLoss()(image_embeddings=ie, text_embeddings=te, text_embeddings2=te2,
                                            logit_scale=s, logits_per_image=li, logits_per_text=lt)
            loss2.backward()
            out.update(text_embeddings2=t2n(te2), mmg_loss=t2n(loss2), mmg_dimg=t2n(img.grad),
                       mmg_dtxt=t2n(txt.grad), mmg_dtxt2=t2n(txt2.grad), mmg_dlogit_scale=t2n(ls.grad))
            np.savez_compressed(os.path.join(OUT, f"g2_head_n{n}.npz"), **out)
    finally:
        torch.Tensor.cuda = orig_cuda


def golden_head_c3(losses):
    """G2 (config C3: `train_prompt_clf`, 2 ranks x 256 pairs = global batch 512, D = 512): the reference's CLIPLoss
    (mmgclip/loss/losses.py:28-44) on the UNSHARDED 512 x 512 problem - what the two ranks' sharded arithmetic must reproduce.
    Lean fixture: inputs, loss and the three gradients only (the embeddings / logits follow from the inputs)."""
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        n, D = 512, 512
        g = torch.Generator().manual_seed(100 + n)
        base = torch.randn(1, D, generator=g)
        img = (0.5 * base + torch.randn(n, D, generator=g)).requires_grad_(True)
        txt = (0.5 * base + 0.7 * img.detach() + torch.randn(n, D, generator=g)).requires_grad_(True)
        ls = torch.tensor(float(np.log(1 / 0.07)), requires_grad=True)
        ie, te, s, li, lt = reference_forward_tail(img, txt, ls)
        loss, labels = losses.CLIPLoss()(logits_per_image=li, logits_per_text=lt, image_embeddings=ie, text_embeddings=te,
                                         logit_scale=s)
        loss.backward()
        assert labels.tolist() == list(range(n))
        np.savez_compressed(os.path.join(OUT, "g2_head_n512.npz"), img=t2n(img), txt=t2n(txt), logit_scale_param=t2n(ls),
                            scale=t2n(s), clip_loss=t2n(loss), clip_dimg=t2n(img.grad), clip_dtxt=t2n(txt.grad),
                            clip_dlogit_scale=t2n(ls.grad))
    finally:
        torch.Tensor.cuda = orig_cuda


# literal inputs printed in notebooks/loss.ipynb (cells 11, 13, 15, 17, 18)
NOTEBOOK_LOGITS = [
    [-0.3695, -0.8987, -0.3323, -0.3540, -0.3375, -0.5998, -0.3583, -0.0797],
    [-0.9398, -1.1682, -0.9602, -0.7505, -1.0275, -0.5558, -0.3456, -0.3068],
    [-0.8346, -1.1233, -0.7055, -0.4546, -0.6598, -0.6412, -0.6927, -0.1958],
    [-0.8875, -1.3657, -0.6414, -0.8099, -0.8178, -0.8100, -0.6184, -0.1464],
    [-0.7839, -1.2652, -0.6129, -0.4527, -0.5410, -0.4618, -0.4844, -0.3835],
    [-1.0263, -1.3110, -0.7902, -0.7323, -0.6832, -0.9224, -0.6688, -0.6417],
    [-0.5663, -0.5041, -0.5145, -0.0413, -0.2905, -0.2322, -0.3936, 0.0914],
    [-0.1942, -0.7119, -0.3226, -0.1033, -0.2929, -0.1779, -0.2586, -0.1330]]
NOTEBOOK_COS = [
    [1.0000, 0.7714, 0.8803, 0.8041, 0.9058, 0.7714, 0.9058, 0.7866],
    [0.7714, 1.0000, 0.8387, 0.9547, 0.8594, 1.0000, 0.8594, 0.6772],
    [0.8803, 0.8387, 1.0000, 0.9082, 0.9064, 0.8387, 0.9064, 0.8011],
    [0.8041, 0.9547, 0.9082, 1.0000, 0.8636, 0.9547, 0.8636, 0.7153],
    [0.9058, 0.8594, 0.9064, 0.8636, 1.0000, 0.8594, 1.0000, 0.7719],
    [0.7714, 1.0000, 0.8387, 0.9547, 0.8594, 1.0000, 0.8594, 0.6772],
    [0.9058, 0.8594, 0.9064, 0.8636, 1.0000, 0.8594, 1.0000, 0.7719],
    [0.7866, 0.6772, 0.8011, 0.7153, 0.7719, 0.6772, 0.7719, 1.0000]]
NOTEBOOK_LABELS = [0, 1, 0, 0, 0, 1, 0, 2]              # loss.ipynb cell 13 output (threshold 0.8)
NOTEBOOK_SOFTMAX_ROW0 = [0.3354, 0.2250, 0.4396]        # loss.ipynb cell 17 output, first row
NOTEBOOK_CE = 1.2048                                    # loss.ipynb cell 18 output


def golden_averaged(losses):
    """G3: the notebook's known answers, re-derived through the reference's AveragedMedicalCLIPLoss helpers."""
    crit = losses.AveragedMedicalCLIPLoss()
    cos = torch.tensor(NOTEBOOK_COS)
    logits = torch.tensor(NOTEBOOK_LOGITS)
    labels = crit._assign_labels(cos, threshold=0.8)
    assert labels == NOTEBOOK_LABELS, labels
    avg = crit._average_logits(logits=logits, list_labels=labels)
    sm = avg.softmax(-1)
    assert np.allclose(t2n(sm[0]), NOTEBOOK_SOFTMAX_ROW0, atol=5e-5), sm[0]
    ce = torch.nn.functional.cross_entropy(avg, torch.tensor(labels))
    assert abs(float(ce) - NOTEBOOK_CE) < 5e-5, float(ce)
    # docstring example of losses.py:129-139
    alt = torch.tensor([[1.0 if (i % 2) == (j % 2) else -0.0237 for j in range(8)] for i in range(8)])
    alt_labels = crit._assign_labels(alt, threshold=0.65)
    assert alt_labels == [0, 1, 0, 1, 0, 1, 0, 1]

    # full forward on seeded embeddings that cluster (4 prototypes), with gradients
    g = torch.Generator().manual_seed(77)
    protos = torch.randn(4, 512, generator=g)
    assign = torch.tensor([0, 1, 0, 2, 3, 1, 0, 2, 3, 3, 1, 0])
    txt = (protos[assign] + 0.05 * torch.randn(12, 512, generator=g)).requires_grad_(True)
    img = torch.randn(12, 512, generator=g).requires_grad_(True)
    ls = torch.tensor(float(np.log(1 / 0.07)), requires_grad=True)
    ie, te, s, li, lt = reference_forward_tail(img, txt, ls)
    # NOTE the reference pairs CE(logits_per_text [n,n], cluster labels) (losses.py:205-213): reproduced as is
    loss, lab = crit(image_embeddings=ie, text_embeddings=te, logit_scale=s, logits_per_image=li,
                     logits_per_text=lt)
    loss.backward()
    np.savez_compressed(
        os.path.join(OUT, "g3_averaged.npz"),
        nb_logits=np.array(NOTEBOOK_LOGITS, np.float32), nb_cos=np.array(NOTEBOOK_COS, np.float32),
        nb_labels=np.array(NOTEBOOK_LABELS), nb_avg=t2n(avg), nb_softmax=t2n(sm), nb_ce=t2n(ce),
        alt_cos=t2n(alt), alt_labels=np.array(alt_labels),
        img=t2n(img), txt=t2n(txt), logit_scale_param=t2n(ls), loss=t2n(loss), labels=t2n(lab),
        dimg=t2n(img.grad), dtxt=t2n(txt.grad), dlogit_scale=t2n(ls.grad))


def golden_schedule(sched):
    """G4: lr sequences for the shipped scheduler configs (configs/scheduler/*.yaml)."""
    res = {}
    for total, warm in ((30, 0.1), (15, 1), (300, 0.1)):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([p], lr=5e-5, weight_decay=1e-4)
        sc = sched.LinearWarmupCosineAnnealingLR(opt, total, warm)
        lrs = []
        for _ in range(total):
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sc.step()
        res[f"{total}_{warm}"] = lrs
    json.dump(res, open(os.path.join(OUT, "g4_lr_schedule.json"), "w"), indent=0)


def golden_early_stopper(es_mod):
    """G7: EarlyStopper state machine on a fixed validation-loss trace (checkpoint writes go to a temp dir)."""
    import tempfile
    trace = [1.0, 0.9, 0.95, 0.97, 0.85, 0.86, 0.87, 0.88, 0.89, 0.90, 0.91]
    model = torch.nn.Linear(2, 2)
    opt = torch.optim.AdamW(model.parameters())
    states = []
    with tempfile.TemporaryDirectory() as d:
        es = es_mod.EarlyStopper(patience=5, trace_func=lambda *_: None)
        path = os.path.join(d, "model.pth")
        for epoch, v in enumerate(trace):
            es(v, epoch, model, opt, path)
            states.append(dict(counter=es.counter, best_score=es.best_score, early_stop=es.early_stop,
                               val_loss_min=es.val_loss_min))
        ckpt = torch.load(path, weights_only=False)
        keys = sorted(ckpt.keys())
    json.dump(dict(trace=trace, states=states, checkpoint_keys=keys, last_saved_epoch=ckpt["epoch"]),
              open(os.path.join(OUT, "g7_early_stopper.json"), "w"), indent=0)


def main():
    torch.set_num_threads(1)   # bit-stable sums
    install_cos_sim_stub()
    proj = load_by_path("ref_projection", "mmgclip/networks/projection.py")
    losses = load_by_path("ref_losses", "mmgclip/loss/losses.py")
    sched = load_by_path("ref_warmup_cosine", "mmgclip/scheduler/warmup_cosine.py")
    es_mod = load_by_path("ref_early_stopping", "mmgclip/callbacks/early_stopping.py")
    if "--only-c3" in sys.argv:           # round 3: add the C3 fixture without re-writing the others
        golden_head_c3(losses)
        return
    golden_projection(proj)
    golden_head(losses)
    golden_head_c3(losses)
    golden_averaged(losses)
    golden_schedule(sched)
    golden_early_stopper(es_mod)
    print("golden vectors written to", OUT)
    # G5 / G9: encoder towers pinned by transformers' independent implementations (+ the reference's projection / loss for C1)
    import make_golden_encoders
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    make_golden_encoders.main(proj, losses)


if __name__ == "__main__":
    main()
