"""bench.py contract on a reduced workload (fast): one JSON line with the required keys,
roofline + cpu_baseline objects, and GPU decisions identical to the oracle's."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract_small():
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1",
                        "--queries", "16", "--db-songs", "400", "--real-songs", "8", "--cpu-queries", "4"],
                       capture_output=True, text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["value"] > 0
    assert out["vs_baseline"] is None and out["dtype"] == "f32" and "workload" in out["config"]
    rf = out["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and 0 < rf["frac"] <= 1.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1
    par = out["oracle_decision_parity"]
    assert par["identical_song_and_offset"] == par["queries"]
    assert out["top1_hit_rate"] > 0.5
