"""tools/bench_cli.py (the product's wall clock with its phases) runs and reports what its consumers read: the sample cases only here
(the ERA5-size files are written and timed by hand: profiles/r04_cli_end_to_end.json)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_cli_times_the_samples_and_splits_the_wall_clock(tmp_path):
    out = tmp_path / "cli.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_cli.py"), "--skip-big", "--out", str(out)], capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.load(open(out))
    assert len(d["cases"]) == 8
    for c in d["cases"]:
        assert c["returncode"] == 0, c
        ph = c["phases_s"]
        for k in ("imports", "library_and_hip_init", "ingest_compute_gather", "csv_writes", "exit"):
            assert k in ph and ph[k] >= 0, (c["case"], k)
        assert ("open_and_plan" in ph) == ("device_ingest" in c["case"]) and ("open_decode_and_prepare" in ph) == ("resident" in c["case"])
        assert abs(sum(ph.values()) - c["wall_s"]) < 0.05 and c["files_written"] >= 23          # 22 CSVs (+ trackfile) + the log
        assert any("framework ran in" in ln for ln in c["log"])
