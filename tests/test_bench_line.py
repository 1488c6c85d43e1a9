"""The line the driver reads has to survive its 8 000-character stdout tail (round 5's 24 KB line did not: BENCH_r05.parsed
was null).  bench.compact_line() is run on a canned FULL result -- round 5's own 24 KB line, committed under profiles/ -- and
on the C3 / C5 lines of the same round; the compact form must stay under 6 000 characters and keep the contract's keys."""
import json
import os

import pytest

from common import ROOT

import bench

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _canned(name):
    return json.load(open(os.path.join(ROOT, "profiles", name)))


def test_compact_line_of_the_drivers_command_fits_and_keeps_the_contract():
    full = _canned("r46_driver_cmd_bench_line.json")
    assert len(json.dumps(full)) > 20000          # the canned result IS the one that broke the record
    line = bench.compact_line(full)
    assert "\n" not in line and len(line) < bench.COMPACT_LIMIT <= 6000, len(line)
    d = json.loads(line)
    for k in CONTRACT:
        assert k in d, k
    assert d["value"] == pytest.approx(full["value"], rel=1e-3) and d["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-3)
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "algorithmic_bytes_per_launch"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-3)
    assert set(r["frame"]) >= {"algorithmic_frac", "frac"}
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert d["parity_check"]["ok"] is True and d["growth_phase"]["parity_ok"] is True
    assert d["host_frames"]["value"] > 0 and d["roofline_valu"]["frac"] > 0
    oc = d["other_configs"]
    assert set(oc) == {"C3", "C5"}
    for name in oc:
        assert oc[name]["value"] > 0 and oc[name]["parity_ok"] is True and oc[name]["roofline"]["frac"] > 0
        assert oc[name]["cpu_baseline"]["value"] > 0
    assert "frame" in oc["C3"]["roofline"]


@pytest.mark.parametrize("name", ["r46_c3_bench_line.json", "r46_c5_bench_line.json", "r46_c2_bench_line.json"])
def test_compact_line_of_the_other_configs(name):
    full = _canned(name)
    line = bench.compact_line(full)
    assert len(line) < 4000, len(line)
    d = json.loads(line)
    for k in CONTRACT:
        assert k in d, k
    assert d["roofline"]["frac"] > 0 and d["parity_check"]["ok"] is True


def test_a_failed_parity_block_is_not_reported_ok():
    full = _canned("r46_driver_cmd_bench_line.json")
    full["cpu_baseline"]["parity_check"]["rows_not_bit_equal"] = [3]
    assert json.loads(bench.compact_line(full))["parity_check"]["ok"] is False
    full["cpu_baseline"]["parity_check"] = {"frames": 2, "surfels": 5, "counts_equal": False, "rows_not_bit_equal": []}
    assert json.loads(bench.compact_line(full))["parity_check"]["ok"] is False


def test_emit_writes_the_detail_file_and_prints_one_short_line(tmp_path, capsys):
    class A:
        config = "C2"
        detail_out = str(tmp_path / "detail.json")
        full_line = False
    full = _canned("r46_driver_cmd_bench_line.json")
    bench.emit(full, A)
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 1 and len(out[0]) < 6000
    assert json.load(open(A.detail_out))["roofline"]["kernels"]      # everything measured is in the detail file
    assert json.loads(out[0])["detail"]
