"""Diagnostic: per-parameter gradient agreement of the HIP C1 step with the golden (run on the GPU box)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mmg-clip_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import test_c1_gpu as T
S = int(sys.argv[1]) if len(sys.argv) > 1 else 77
g = np.load(os.path.join(ROOT, "tests", "golden", f"g9_c1_step_s{S}.npz"))
model, batch = T.build_c1_model(g)
e = T.c1_errors(model, batch, g)
print({k: v for k, v in e.items()})
img_p = dict(model.image_encoder.model.named_parameters()); txt_p = dict(model.text_encoder.model.named_parameters())
rows = []
for k in g.files:
    if not k.startswith("grad.") or k.endswith(".rows"): continue
    tower, name = k[5:].split(".", 1)
    p = (img_p if tower == "image" else txt_p)[name]
    rows.append((T.cosine(p.grad, g[k]), T.rel(p.grad, g[k]), float(np.linalg.norm(g[k])), float(p.grad.norm()), k))
for r in sorted(rows)[:60]:
    print("cos %.4f rel %.3e |gold| %.3e |ours| %.3e %s" % r)
