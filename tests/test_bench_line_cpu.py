"""bench.py's ONE stdout line stays small enough for the driver to parse (the 20 KB line of round 5 came back
`parsed: null`).  The assembler is fed the largest full record a round has produced (profiles/r05_bench.json, 20 KB) with every string
inflated and every side object present, and must still emit < 6000 bytes carrying headline + roofline + cpu_baseline + side_rates."""
import copy
import json
import os

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def full_record():
    return json.load(open(os.path.join(ROOT, "profiles", "r05_bench.json")))


def inflate(o):
    if isinstance(o, str):
        return o + " " + "x" * 400
    if isinstance(o, dict):
        return {k: inflate(v) for k, v in o.items()}
    if isinstance(o, list):
        return [inflate(v) for v in o]
    return o


def test_line_is_small_and_complete():
    rec = full_record()
    assert len(json.dumps(rec)) > 15000  # the record that broke the driver's parser
    text = bench.compact_line(rec)
    assert "\n" not in text and len(text) < bench.LINE_LIMIT <= 6000
    d = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in d, k
    assert d["metric"] == rec["metric"] and d["unit"] == "frames/s" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert abs(d["value"] - rec["value"]) / rec["value"] < 1e-4
    r = d["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launches", "avg_launch_ms", "flops_per_launch",
              "algorithmic_bytes_per_launch"):
        assert k in r, k
    assert len(r["kernel"]) <= 80 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["all_convs"]["frac"] > 0
    b = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "host_cpu", "frames_per_s_by_threads", "c_oracle_3_threads"):
        assert k in b, k
    side = d["config"]["side_rates"]
    assert set(side) == set(bench.SIDE_KEYS)
    assert side["f16hl_mode_1080p"][0] > 0 and side["configs4_r101_f16_4k"][1] > 0
    assert "workload" in d["config"] and "BASELINE configs[1]" in d["config"]["workload"]


def test_inflated_record_still_fits():
    text = bench.compact_line(inflate(full_record()))
    assert len(text) < bench.LINE_LIMIT
    json.loads(text)


def test_errors_in_side_objects_do_not_grow_the_line():
    rec = copy.deepcopy(full_record())
    for k in bench.SIDE_KEYS:
        rec[k] = {"error": "RuntimeError: " + "y" * 5000}
    d = json.loads(bench.compact_line(rec))
    assert all(len(v) <= 60 for v in d["config"]["side_rates"].values())


def test_oversized_line_is_refused(monkeypatch):
    monkeypatch.setattr(bench, "LINE_LIMIT", 500)
    with pytest.raises(AssertionError):
        bench.compact_line(full_record())


def test_emit_writes_detail_file_and_one_stdout_line(tmp_path, monkeypatch, capsys):
    monkeypatch.setattr(bench, "DETAIL_FILE", str(tmp_path / "bench_detail.json"))
    rec = full_record()
    bench.emit(rec)
    cap = capsys.readouterr()
    lines = [ln for ln in cap.out.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) < 6000
    assert json.loads(lines[0])["value"] > 0
    assert json.load(open(tmp_path / "bench_detail.json"))["roofline"]["other_kernels"]  # the tables live here now
    assert "other_kernels" in cap.err
