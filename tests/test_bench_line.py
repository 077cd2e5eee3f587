"""The driver parses the LAST stdout line of `bench.py`; round 4's line (24.6 KB) did not parse.  These tests pin the
formatter: a canned full record (round 4's own driver-like run, tests/golden/bench_record_r04.json) must come out as ONE
JSON object below 4 KB that round-trips and carries the contract keys, `roofline` and `cpu_baseline`."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline", "details")


def canned():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "bench_record_r04.json")))


def test_compact_line_is_small_and_round_trips():
    rec = canned()
    assert len(json.dumps(rec)) > 20000  # the record that broke the driver's parser
    line = bench.compact_line(rec, "gpurun_out/bench_details.json")
    assert "\n" not in line and len(line) < 4096
    got = json.loads(line)
    for k in CONTRACT:
        assert k in got, k
    assert got["value"] == rec["value"] and got["ms_per_step"] == rec["ms_per_step"]
    assert got["config"]["workload"].startswith("configs[2]")
    for k in ("kernel", "kernel_ms", "bound", "achieved", "peak", "frac", "traffic"):
        assert k in got["roofline"], k
    assert abs(got["roofline"]["frac"] - got["roofline"]["achieved"] / got["roofline"]["peak"]) < 1e-9
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in got["cpu_baseline"], k


def test_compact_line_survives_oversized_fields():
    rec = canned()
    rec["config"]["workload"] = "w" * 5000
    rec["cpu_baseline"]["sample"] = "s" * 5000
    rec["roofline"]["counter_source"] = "c" * 5000
    rec["n_gpus"] = 2
    rec["ranks"] = {"per_rank_kernel_ms": [0.1] * 512}
    line = bench.compact_line(rec, "gpurun_out/bench_details_n2.json")
    assert len(line) < 4096
    got = json.loads(line)
    assert got["n_gpus"] == 2 and "ranks" not in got and got["details"].endswith("_n2.json")


def test_emit_prints_the_compact_line_last_and_writes_details(tmp_path, monkeypatch):
    rec = canned()
    det = tmp_path / "d" / "bench_details.json"
    monkeypatch.setenv("HSM_BENCH_DETAILS", str(det))
    monkeypatch.delenv("HSM_BENCH_CHILD", raising=False)
    buf = io.StringIO()
    with redirect_stdout(buf):
        print("noise before the line")
        bench.emit(rec)
    last = buf.getvalue().strip().splitlines()[-1]
    assert len(last) < 4096 and json.loads(last)["value"] == rec["value"]
    full = json.load(open(det))
    assert full["configs"] == rec["configs"] and full["roofline"]["valu"] == rec["roofline"]["valu"]  # nothing is lost, only moved


def test_child_legs_keep_the_full_record(monkeypatch):
    rec = canned()
    monkeypatch.setenv("HSM_BENCH_CHILD", "1")
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.emit(rec)
    assert json.loads(buf.getvalue().strip().splitlines()[-1]) == rec


def test_deferred_emit_holds_the_line_back(monkeypatch, tmp_path):
    """N > 1: the line of an extra workload is held until the process group is destroyed and the other ranks have exited
    (RCCL's printf banner would otherwise land behind it); emit() then prints it"""
    rec = canned()
    monkeypatch.setenv("HSM_BENCH_DETAILS", str(tmp_path / "d.json"))
    monkeypatch.delenv("HSM_BENCH_CHILD", raising=False)
    from hsm_bench import common
    common.defer_emit(True)
    buf = io.StringIO()
    try:
        with redirect_stdout(buf):
            bench.emit(rec)
        assert buf.getvalue() == "" and common._PENDING and common._PENDING[-1] is rec
    finally:
        common.defer_emit(False)
    with redirect_stdout(buf):
        common.emit_pending()
    assert not common._PENDING
    assert json.loads(buf.getvalue().strip().splitlines()[-1])["value"] == rec["value"]


def test_driver_entry_is_small():
    """bench.py is the driver's entry point and nothing else: arguments, self-launch of --gpus N, dispatch into hsm_bench/"""
    n = len(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")).read().splitlines())
    assert n < 300, n
