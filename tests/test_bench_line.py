"""bench.py's stdout contract (CPU): ONE line, under 6 KB, with the keys the driver parses -- whatever the run held.

Round 5's driver record had `parsed: null`: the single line had grown to 28.5 KB.  The line is now assembled by
bench.compact_line() from the full record (which goes to stderr and gpurun_out/); these tests feed it the full records of
earlier rounds (profiles/) and a deliberately oversized one."""
import copy
import json
import os

import pytest

import bench
from tests.bench_io import COMPACT_LIMIT

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline")


def _full_records():
    out = []
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles"))):
        if name.startswith(("r04_bench", "r05_bench")) and name.endswith(".json"):
            with open(os.path.join(ROOT, "profiles", name)) as f:
                try:
                    rec = json.loads(f.read().strip().splitlines()[-1])
                except ValueError:
                    continue
            if isinstance(rec, dict) and "metric" in rec and "roofline" in rec:
                out.append((name, rec))
    return out


def test_earlier_rounds_full_lines_compact_below_the_limit():
    recs = _full_records()
    assert len(recs) >= 5
    biggest = 0
    for name, rec in recs:
        biggest = max(biggest, len(json.dumps(rec)))
        text = bench.compact_line(rec)
        assert "\n" not in text and len(text) < COMPACT_LIMIT, (name, len(text))
        c = json.loads(text)
        for k in REQUIRED:
            assert k in c, (name, k)
        assert c["roofline"]["frac"] == pytest.approx(rec["roofline"]["frac"], rel=1e-5)
        assert c["roofline"]["bound"] == rec["roofline"]["bound"] and c["roofline"]["kernel"] == rec["roofline"]["kernel"]
        assert c["value"] == pytest.approx(rec["value"], rel=1e-5) and c["config"]["workload"] == rec["config"]["workload"]
        if rec.get("cpu_baseline"):
            assert c["cpu_baseline"]["value"] == pytest.approx(rec["cpu_baseline"]["value"], rel=1e-5)
            assert set(c["cpu_baseline"]) <= {"value", "unit", "cores", "kind", "wall_s", "sample"}
        if "summary" in rec:                           # (round 5 on: one short row per workload of the run)
            assert len(c["summary"]["rows"]) == len(rec["summary"]["rows"])
        # nothing bulky rides along
        assert "secondary" not in c and "plan" not in c["config"] and not any(k.endswith("_sorted") for k in c["roofline"])
    assert biggest > 20000          # (the records fed in really are the long ones)


def test_an_oversized_record_sheds_detail_but_never_the_contract():
    name, rec = max(_full_records(), key=lambda nr: len(json.dumps(nr[1])))
    big = copy.deepcopy(rec)
    big["n_gpus"] = 8
    big.update(rccl_ranks=8, backend="nccl", gather_ms=1.234567, gather_bytes=8 * 365 * 24378 * 4, gather_ok=True,
               per_rank_ms=[127.0 + 0.01 * i for i in range(8)])
    big["config"]["workload"] = big["config"]["workload"] + " -- " + "x" * 3000
    big["cpu_baseline"]["sample"] = big["cpu_baseline"]["sample"] + " " + "y" * 3000
    row = dict(big["summary"]["rows"][0], cpu_port_s=0.36, cpu_reference_shaped_s={"numpy": 12.5, "pandas": 9.25, "rows_timed": [64, 96]},
               gather_ms=1.5, gather_ok=True)
    big["summary"]["rows"] = [dict(row, wl="w%02d" % i) for i in range(40)]
    text = bench.compact_line(big)
    assert len(text) < COMPACT_LIMIT
    c = json.loads(text)
    for k in REQUIRED:
        assert k in c
    assert c["n_gpus"] == 8 and c["rccl_ranks"] == 8 and len(c["per_rank_ms"]) == 8 and c["gather_ms"] == pytest.approx(1.234567, rel=1e-5)
    assert c["roofline"]["frac"] == pytest.approx(rec["roofline"]["frac"], rel=1e-5) and c["cpu_baseline"]["value"] > 0
    assert len(c["config"]["workload"]) <= 120 and len(c["cpu_baseline"]["sample"]) <= 120


def test_non_finite_numbers_do_not_reach_the_line():
    name, rec = _full_records()[0]
    bad = copy.deepcopy(rec)
    bad["roofline"]["traffic"] = float("nan")
    bad["max_ms"] = float("inf")
    text = bench.compact_line(bad)
    assert "NaN" not in text and "Infinity" not in text
    c = json.loads(text)
    assert c["roofline"]["traffic"] is None and c["max_ms"] is None


def test_reference_shaped_cpu_leg_times_the_literal_restatements():
    """bench.cpu_reference_shaped (VERDICT r5 Next #2): the NumPy scatter-add and the pandas groupby restatements of
    aggregations.py:24-27,73,78-80 on the workload's own operands -- full size when it fits the budget, else a window of
    leading rows with the per-call fixed cost (label lookup, factorisation) kept out of the scaling."""
    import numpy as np
    from climate_toolbox_amd import synth
    lat, lon, tas, df = synth.c1_workload(T=40)
    X = np.ascontiguousarray(tas.reshape(40, -1))
    full = bench.cpu_reference_shaped(X, lat, lon, df, "areawt", "hierid", budget_s=30.0)
    for leg in ("numpy_scatter", "pandas_groupby"):
        r = full[leg]
        assert r["rows_timed"] == r["rows"] == 40 and r["scaled"] is False and r["wall_s"] == r["timed_wall_s"] > 0
        assert 0 <= r["fixed_s"] <= r["wall_s"] * 1.5 and r["value"] > 0
    assert full["threads"] == 1
    tight = bench.cpu_reference_shaped(X, lat, lon, df, "areawt", "hierid", budget_s=1e-4)      # nothing fits: the 8-row probe window
    for leg in ("numpy_scatter", "pandas_groupby"):
        r = tight[leg]
        assert r["rows_timed"] == 8 and r["scaled"] is True and r["wall_s"] >= r["fixed_s"]
        assert r["wall_s"] == pytest.approx(r["fixed_s"] + (r["timed_wall_s"] - r["fixed_s"]) * 40 / 8, rel=1e-2, abs=1e-3)          # (the fields are rounded to 1e-4 s)
