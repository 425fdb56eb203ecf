import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mmg-clip_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "bert_dropout: the test exercises the text tower's training-mode dropout (default: off in tests)")


@pytest.fixture(autouse=True)
def _deterministic_text_tower(request, monkeypatch):
    """The parity tests compare training-mode towers with deterministic (eval-semantics) oracles, so the text tower's
    training-mode dropout is switched off through its operational knob MMG_BERT_DROPOUT=0 (inherited by torchrun children).
    Tests marked `bert_dropout` run with the product default (dropout live under train())."""
    if request.node.get_closest_marker("bert_dropout") is None:
        monkeypatch.setenv("MMG_BERT_DROPOUT", "0")
    else:
        monkeypatch.delenv("MMG_BERT_DROPOUT", raising=False)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def measured(name, **values):
    """Append the values a tolerance test actually measured to gpurun_out/measured_tolerances.jsonl (the bars in the tests were
    set from such a run; profiles/r02_measured_tolerances.jsonl is the committed copy)."""
    import json
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "measured_tolerances.jsonl"), "a") as fh:
            fh.write(json.dumps({"test": name, **{k: (float(v) if not isinstance(v, (str, int)) else v) for k, v in values.items()}}) + "\n")
    except OSError:
        pass
