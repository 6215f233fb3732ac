"""bench.py's contract line stays a SHORT line (CPU; no GPU, no library call).

Round 5's line had grown to 32 KB and the driver, which keeps an 8 KB tail of stdout, could not parse it.  The line is now
built by `bench.contract_line` from the full report (which goes to bench_report.json / stderr): these tests feed it the
largest reports on record and check the size bound and the contract keys.
"""
import copy
import io
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _canned():
    """The 32 KB report of round 5 (profiles/r14_bench_bf16.json: eleven companions, a memory block in each)."""
    return json.loads(open(os.path.join(ROOT, "profiles", "r14_bench_bf16.json")).read().strip().splitlines()[-1])


def test_line_of_the_largest_report_on_record_is_short_and_complete():
    rep = _canned()
    assert len(json.dumps(rep)) > 30000
    # what the driver's `--steps 20` adds: eighteen more per-step entries in the companions
    rep["real_geometry"]["step_ms"] = [128.123456789] * 20
    rep["n1_same_arithmetic"] = {"value": 2600.123456, "unit": "frames/s", "ms_per_step": 787.6, "phase_ms": {"blocks": [700.0, 700.0]}}
    line = bench.contract_line(rep, "bench_report.json")
    text = json.dumps(line)
    assert len(text) < bench.LINE_LIMIT == 4096, len(text)
    for k in CONTRACT:
        assert k in line, k
    assert line["value"] == pytest.approx(rep["value"], rel=1e-6) and line["ms_per_step"] == pytest.approx(rep["ms_per_step"], rel=1e-6)
    r = line["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-4) and r["traffic"] > 0
    c = line["cpu_baseline"]
    assert set(c) == {"value", "unit", "cores", "kind", "sample"} and c["kind"] == "port" and c["cores"] == 128
    assert line["config"]["workload"] and all(len(v) <= 120 for v in line["config"].values() if isinstance(v, str))
    assert line["self_check"] == "ok" and line["report"] == "bench_report.json"
    assert len(line["memory"]) == 3
    assert all(isinstance(v, float) for v in line["roofline_hbm_kernels"].values())
    assert line["companions_frames_per_s"]["real_geometry"] == pytest.approx(rep["real_geometry"]["value"], rel=1e-4)
    assert line["n1_same_arithmetic"] == pytest.approx(2600.12, rel=1e-4)
    assert "step_ms" not in text and "kernels_timed_region" not in text and "split_bytes" not in text


def test_line_of_a_sharded_report_carries_phases_and_stays_short():
    """The N > 1 report (retake/sharded.py measure_sharded): `sharded_check` and the kernel table stay in the report."""
    rep = {k: v for k, v in _canned().items() if k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step",
                                                      "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "roofline",
                                                      "retained_kv_tokens_per_s", "cache_checksum")}
    names = ("assembly", "barrier_idle", "blocks", "dpselect", "finalize", "host_wall", "offsets", "rotate", "step")
    rep.update({"n_gpus": 8, "cpu_baseline": None, "sharded_equals_sequential": True, "rccl_world_size": 8,
                "config": {"workload": "BASELINE configs[3]: 2048-frame video sharded by frame chunk over 8 GPU(s), 64 chunks x 28 layers, "
                                       "L=6272", "workload_detail": "x" * 400, "frames": 2048, "chunks": 64, "layers": 28,
                           "chunk_tokens": 6272, "parallelism": "chunk-sharded x8", "transport": "rccl",
                           "assembled_cache_tokens": 100352},
                "sharded_check": {"equal": True, "cases": [{"dtype": "bf16", "chunks": 16, "blocks": [[i, i + 2] for i in range(8)]}] * 4,
                                  "checked": "y" * 300},
                "kernels_timed_region_rank0": {"score_pass1": {"launches": 16, "avg_us": 6500.0, "total_ms": 104.0}},
                "phase_ms": {n: [123.4, 120.1] for n in names}, "phase_bytes_rank0": {"assembly_rows_received": 5_000_000_000}})
    line = bench.contract_line(rep, None)
    assert len(json.dumps(line)) < 2048
    assert line["cpu_baseline"] is None and line["n_gpus"] == 8 and line["rccl_world_size"] == 8
    assert line["sharded_equals_sequential"] is True and set(line["phase_ms"]) == set(names)
    assert line["phase_ms"]["blocks"] == [123.4, 120.1]
    assert "sharded_check" not in line and "kernels_timed_region_rank0" not in line and "workload_detail" not in line["config"]


def test_optional_summaries_go_before_the_contract_keys_do():
    """A report with absurdly many extra kernels / companions still yields a line under the limit: the optional summaries
    are dropped, the contract keys never."""
    rep = copy.deepcopy(_canned())
    rep["roofline_hbm_kernels"] = {f"kernel_with_a_long_name_{i:04d}": {"frac": 0.123456} for i in range(400)}
    line = bench.contract_line(rep, None)
    assert len(json.dumps(line)) < bench.LINE_LIMIT
    assert "roofline_hbm_kernels" not in line
    for k in CONTRACT:
        assert k in line, k


def test_emit_prints_the_line_last_and_writes_the_report(tmp_path, capsys):
    rep = _canned()
    path = os.path.join(str(tmp_path), "r.json")
    line = bench.emit(rep, path)
    out, err = capsys.readouterr()
    assert json.loads(out.strip().splitlines()[-1]) == line and len(out.strip().splitlines()) == 1
    assert len(out) < 4096 + 1
    assert json.load(open(path)) == rep == json.loads(err.strip().splitlines()[-1])
    assert line["report"].endswith("r.json")


def test_a_failed_world1_companion_is_reported_in_the_line_not_raised():
    """`n1_same_arithmetic` runs in a child process; if it fails the headline must still be printed - the line then
    carries a short reason instead of a number."""
    rep = _canned()
    rep["n1_same_arithmetic"] = {"error": "the world-size-1 sharded run did not finish in 300 s " + "x" * 500}
    line = bench.contract_line(rep, None)
    assert isinstance(line["n1_same_arithmetic"], str) and len(line["n1_same_arithmetic"]) <= 80
    assert line["n1_same_arithmetic"].startswith("the world-size-1 sharded run did not finish")
    assert len(json.dumps(line)) < bench.LINE_LIMIT and line["value"] == pytest.approx(rep["value"], rel=1e-6)
