"""bench.py's stdout contract: ONE compact JSON line (< 4 KB) whatever the per-kernel record holds.

BENCH_r03.json: the driver keeps a tail of stdout; the round-3 line was 46 KB (every kernel instantiation x shape class), lost its
head and parsed to nothing.  The recorded detail of that very run (profiles/r03_bench_default.json) is the fixture here."""
import copy
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config")


@pytest.fixture(scope="module")
def recorded():
    return json.load(open(os.path.join(ROOT, "profiles", "r03_bench_default.json")))


def test_contract_line_from_a_recorded_profile_dump_is_compact(recorded):
    import bench
    assert len(json.dumps(recorded)) > 40000                       # the record that broke the driver's parse
    text = bench.contract_line(recorded, "gpurun_out/bench_detail.json")
    assert len(text) < 4096 and "\n" not in text
    line = json.loads(text)
    for k in CONTRACT:
        assert line[k] == recorded[k] or k == "config", k
    assert line["config"]["workload"].startswith("C2: train_binary_class_clf") and line["config"]["global_batch"] == 256
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert r[k] == recorded["roofline"][k]
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-3)
    assert 1 <= len(r["shape_classes"]) <= 3
    assert r["shape_classes"][0]["shape"] == recorded["roofline"]["shape_classes"][0]["shape"]
    c = line["cpu_baseline"]
    assert (c["value"], c["cores"], c["kind"]) == (recorded["cpu_baseline"]["value"], 16, "port")
    assert c["s256"] == recorded["cpu_baseline"]["s256"]["value"]
    assert c["faithful_s77"] == recorded["cpu_baseline"]["faithful"]["s77"]["value"]
    assert c["c2_sample"] == recorded["cpu_baseline"]["c2_sample"]["value"]
    assert "roofline_other_kernels" not in line and line["detail"] == "gpurun_out/bench_detail.json"
    assert line["roofline_method"]["profiled_steps"] == recorded["roofline_method"]["profiled_steps"]
    assert len(line["next_kernels"]) <= 8 and line["next_kernels"][0][0] == recorded["roofline_other_kernels"][0]["kernel"][:48]


def test_contract_line_stays_below_the_limit_for_a_bloated_record(recorded):
    """N > 1 adds `comm`; long kernel names, long samples and hundreds of shape classes must not grow the line."""
    import bench
    d = copy.deepcopy(recorded)
    d["n_gpus"] = 8
    d["comm"] = {"allgather_bytes_per_step": 16777216, "allreduce_bytes_per_step": 547356672, "allreduce_launches_per_step": 12,
                 "allreduce_device_ms_per_step": 3.21, "compute_wait_ms_per_step": 0.42, "overlap": True,
                 "bucket_env": {"MMG_GRAD_OVERLAP": None, "MMG_BUCKET_MB": "x" * 300, "MMG_RCCL_MAX_CHANNELS": None}}
    d["roofline"]["kernel"] = "k" * 400
    d["roofline"]["shape_classes"] = [dict(d["roofline"]["shape_classes"][0], shape="s" * 500) for _ in range(200)]
    d["roofline_other_kernels"] = [dict(o, kernel="o" * 300) for o in d["roofline_other_kernels"]] * 10
    d["cpu_baseline"]["sample"] = "z" * 5000
    d["config"]["workload"] = "w" * 3000
    text = bench.contract_line(d, "gpurun_out/bench_detail.json")
    assert len(text) < 4096
    line = json.loads(text)
    assert line["value"] == recorded["value"] and line["roofline"]["frac"] == recorded["roofline"]["frac"]
    assert line["cpu_baseline"]["value"] == recorded["cpu_baseline"]["value"] and line["comm"]["allreduce_launches_per_step"] == 12


def test_contract_line_without_optional_blocks():
    import bench
    d = {"metric": "image-text pairs/sec (global batch)", "value": 1.0, "unit": "image-text pairs/sec", "n_gpus": 1, "steps": 1, "warmup": 0,
         "ms_per_step": 1.0, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
         "config": {"workload": "x"}}
    assert json.loads(bench.contract_line(d)) == d


def test_detail_file_round_trips(tmp_path, recorded):
    import bench
    p = bench.write_detail(recorded, str(tmp_path / "sub" / "bench_detail.json"))
    assert p and json.load(open(p)) == recorded
