"""The driver keeps a bounded tail of bench.py's stdout (round 4: a 25.7 KB line came back unparsed, so the headline counted as
unmeasured).  The printed record must stay small, carry the contract's fields and round-trip; everything else lives in
bench_detail.json.  Input = the FULL records earlier rounds committed under profiles/ (the largest real ones there are)."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline")
FULL = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[2-9]_bench_*.json")))


@pytest.mark.parametrize("path", FULL, ids=[os.path.basename(p) for p in FULL])
def test_compact_line_is_small_complete_and_round_trips(path, tmp_path, capsys, monkeypatch):
    with open(path) as fh:
        full = json.load(fh)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))               # the detail file goes to a scratch directory
    line = bench.emit(full)
    text = capsys.readouterr().out.strip()
    assert "\n" not in text and len(text) < bench.LINE_LIMIT == 4096
    back = json.loads(text)
    assert back == json.loads(json.dumps(line))
    for k in CONTRACT:
        assert k in back, k
    for k in ("metric", "value", "unit", "ms_per_step"):
        assert back[k] == full[k]
    assert "workload" in back["config"] and "model" not in back["config"]
    if full.get("roofline"):
        r = back["roofline"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in r, k
        assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] <= 1.0
        assert "per_kernel" not in r and "conv_family" not in r and "whole_step" not in r      # tables and re-based figures: detail file only
    if full.get("cpu_baseline"):
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in back["cpu_baseline"], k
    with open(tmp_path / bench.DETAIL_FILE) as fh:                  # the full record is what the detail file holds
        assert json.load(fh) == full
    assert back["detail"] == bench.DETAIL_FILE


def test_line_survives_a_pathologically_long_record(tmp_path, capsys, monkeypatch):
    """Strings are bounded and optional parts are dropped before the limit is reached; an impossible record fails loudly HERE."""
    with open(FULL[-1]) as fh:
        full = json.load(fh)
    full["dtype"] = "x" * 5000
    full["config"]["workload"] = "NYU-v2 228x304 batch=16 " + "y" * 5000
    full["config"]["extra_configs"] = (full["config"].get("extra_configs") or [{"metric": "m", "value": 1.0}]) * 8
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.emit(full)
    text = capsys.readouterr().out.strip()
    assert len(text) < bench.LINE_LIMIT
    assert json.loads(text)["config"]["workload"].startswith("NYU-v2 228x304 batch=16")
